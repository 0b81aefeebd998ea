#!/bin/bash
# round-5 closing pass: the whole GPU suite, smoke, the metric's bench line (telemetry, calibration, live PMC traffic), 720p and 1056p lines
TAG=${1:-r05z}
(timeout 1700 python -m pytest tests -q -m gpu 2>&1 | tail -25) > gpurun_out/${TAG}_pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1
timeout 900 python bench.py --workload 1080p --steps 3 --warmup 1 > gpurun_out/${TAG}_bench1080p.json.log 2>&1
timeout 600 python bench.py --workload 720p --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench720p.json.log 2>&1
timeout 900 python bench.py --workload 1056p --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench1056p.json.log 2>&1
tail -4 gpurun_out/${TAG}_pytest_gpu.log; tail -1 gpurun_out/${TAG}_smoke.log
for w in 1080p 720p 1056p; do tail -1 gpurun_out/${TAG}_bench${w}.json.log | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline())
t=l.get('telemetry') or {}
print('$w', {k:round(l[k],4) if isinstance(l[k],float) else l[k] for k in ('value','ms_per_step','sec_per_video','model_tflops_per_gpu','mfma_frac_whole_step','box_attn_tflops','box_gemm_tflops')}, 'attn', round(l['roofline']['achieved'],1), round(l['roofline']['frac'],4), round(l['roofline']['ms_per_launch'],2), 'traffic', l['roofline']['traffic'], 'vae', round(l['vae_decode']['seconds'],3), 'tel', {k:t.get(k) for k in ('sclk_mhz_mean','power_w_mean','temp_c_max')}, (t.get('residency') or {}).get('ppt'))
"; done
