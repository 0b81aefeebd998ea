#!/bin/bash
# round-4 final GPU pass: tools/round_end_gpu.sh (whole parity suite, smoke, the metric's bench line with live PMC traffic,
# rocprofv3 kernel stats, FETCH/WRITE summaries), SQ counters of the attention kernel, the 720p / 1056p lines
TAG=${1:-r04z}
BENCH_STEPS=2 bash tools/round_end_gpu.sh $TAG 1080p > gpurun_out/${TAG}_round_end.log 2>&1
bash tools/pmc_kernel.sh attn1 ${TAG}_pmc_attn > gpurun_out/${TAG}_pmc_attn_m16.txt 2>&1
rm -rf gpurun_out/${TAG}_pmc_attn
timeout 600 python bench.py --workload 720p --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench720p.json.log 2>&1
timeout 900 python bench.py --workload 1056p --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench1056p.json.log 2>&1
tail -12 gpurun_out/${TAG}_round_end.log | cut -c1-600; tail -12 gpurun_out/${TAG}_pmc_attn_m16.txt; tail -1 gpurun_out/${TAG}_bench720p.json.log | cut -c1-300; tail -1 gpurun_out/${TAG}_bench1056p.json.log | cut -c1-300
