#!/bin/bash
# round-5 pass Q: rocprofv3 kernel stats of one 1080p step with the current tree (GEMM rows, attention, the rest)
TAG=${1:-r05q}
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --workload 1080p --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-calibration > $R/gpurun_out/${TAG}_bench1080p_prof.log 2>&1
cd $R
python3 tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*/*_results.db gpurun_out/${TAG}_prof/*_results.db 2>/dev/null | head -1) gpurun_out/${TAG}_bench1080p_kernel_stats.txt > /dev/null 2>&1
rm -rf gpurun_out/${TAG}_prof
head -16 gpurun_out/${TAG}_bench1080p_kernel_stats.txt | cut -c1-170; tail -1 gpurun_out/${TAG}_bench1080p_prof.log | cut -c1-300
