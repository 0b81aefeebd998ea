#!/bin/bash
# round-4 closing pass with GEMM variant 11 as the default: whole parity suite, smoke, the metric's bench line (live PMC traffic),
# rocprofv3 kernel stats of one step
TAG=${1:-r04y}
R=$PWD; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25) > gpurun_out/${TAG}_pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1
timeout 900 python bench.py --workload 1080p --steps 2 --warmup 1 > gpurun_out/${TAG}_bench1080p.json.log 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --workload 1080p --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > $R/gpurun_out/${TAG}_bench1080p_prof.log 2>&1
cd $R
python3 tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*/*_results.db gpurun_out/${TAG}_prof/*_results.db 2>/dev/null | head -1) gpurun_out/${TAG}_bench1080p_kernel_stats.txt > /dev/null 2>&1
rm -rf gpurun_out/${TAG}_prof
tail -4 gpurun_out/${TAG}_pytest_gpu.log; tail -2 gpurun_out/${TAG}_smoke.log; tail -1 gpurun_out/${TAG}_bench1080p.json.log | cut -c1-1500; head -12 gpurun_out/${TAG}_bench1080p_kernel_stats.txt | cut -c1-160
