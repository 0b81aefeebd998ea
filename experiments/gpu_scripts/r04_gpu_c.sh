#!/bin/bash
# round-4 GPU pass C: the tests that failed in pass B, GEMM residual warm-up A/B, one bench step with the live PMC passes
TAG=${1:-r04c}
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu -k "seam or vae_attention or bench_multirank or vae_fast or gemm or test_vae_pipelined" 2>&1 | tail -120) > gpurun_out/${TAG}_pytest_sel.log
timeout 900 moviigen1.1_amd/lib/mg_selftest gemmab 131040 2 80 81 82 83 > gpurun_out/${TAG}_gemm_warm.log 2>&1
timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench1080p.json.log 2>&1
grep -n "^FAILED\|passed\|failed" gpurun_out/${TAG}_pytest_sel.log | tail -8; cat gpurun_out/${TAG}_gemm_warm.log | tail -40; python3 - <<PY
import json
for ln in open('gpurun_out/${TAG}_bench1080p.json.log'):
    if ln.startswith('{'):
        d = json.loads(ln); print(d['ms_per_step'], d['roofline'], d.get('vae_decode'))
PY
tail -3 gpurun_out/${TAG}_bench1080p.json.log | cut -c1-400
