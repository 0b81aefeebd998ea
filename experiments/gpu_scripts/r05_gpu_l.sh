#!/bin/bash
# round-5 pass L: the tests touched by the library split and the new full-size tests, smoke, one bench line with telemetry
TAG=${1:-r05l}
(timeout 1200 python -m pytest tests -q -m gpu -x -k "gemm or rmsnorm_rope or rowwise or block_composition or attention_rescale or attention_vs_oracle or attention_prescaled or fullsize_attention_properties or cabi or ln_modulate or gate_residual or dit_forward_vs_reference or operator_seam" 2>&1 | tail -15) > gpurun_out/${TAG}_pytest_subset.log
python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1
timeout 900 python bench.py --workload 1080p --steps 2 --warmup 1 > gpurun_out/${TAG}_bench1080p.json.log 2>&1
tail -6 gpurun_out/${TAG}_pytest_subset.log; tail -1 gpurun_out/${TAG}_smoke.log; tail -1 gpurun_out/${TAG}_bench1080p.json.log | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline())
print({k:l[k] for k in ('value','ms_per_step','sec_per_video','model_tflops_per_gpu','mfma_frac_whole_step','box_attn_tflops','box_gemm_tflops')})
print('roofline',{k:l['roofline'][k] for k in ('achieved','frac','ms_per_launch','traffic')})
print('telemetry',l['telemetry'])
print('vae',l['vae_decode']['seconds'])
"
