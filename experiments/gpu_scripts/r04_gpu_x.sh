#!/bin/bash
# step-level A/B on ONE box: the metric's workload with GEMM variant 8 forced (the round-3 default) and with the library default (11)
TAG=${1:-r04x}
for v in ${AB_ORDER:-8 0 8 0}; do
timeout 400 python bench.py --workload 1080p --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-video-tail --gemm-variant $v 2>&1 | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('gemm variant', d['config'].get('gemm_variant', 'default (11)'), ': %.3f s/step, attention %.2f ms/launch' % (d['ms_per_step'] / 1e3, d['roofline']['ms_per_launch']))"
done > gpurun_out/${TAG}_step_ab.log 2>&1
cat gpurun_out/${TAG}_step_ab.log
