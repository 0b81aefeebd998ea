#!/bin/bash
# after the last kernel changes (non-temporal stores of the bf16 GEMM tiles): GEMM / attention / block tests, smoke, the metric's bench line
TAG=${1:-r05fin}
(timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gemm or attention or block or dit" 2>&1 | tail -4) > gpurun_out/${TAG}_pytest_subset.log
python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1
timeout 900 python bench.py --workload 1080p --steps 3 --warmup 1 > gpurun_out/${TAG}_bench1080p.json.log 2>&1
tail -2 gpurun_out/${TAG}_pytest_subset.log; tail -1 gpurun_out/${TAG}_smoke.log
tail -1 gpurun_out/${TAG}_bench1080p.json.log | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline())
t=l.get('telemetry') or {}
print({k:round(l[k],4) if isinstance(l[k],float) else l[k] for k in ('value','ms_per_step','sec_per_video','model_tflops_per_gpu','mfma_frac_whole_step','box_attn_tflops','box_gemm_tflops')}, 'attn', round(l['roofline']['achieved'],1), round(l['roofline']['frac'],4), round(l['roofline']['ms_per_launch'],2), 'vae', round(l['vae_decode']['seconds'],3), 'tel', {k:t.get(k) for k in ('sclk_mhz_mean','power_w_mean','temp_c_max')}, (t.get('residency') or {}).get('ppt'))
"
