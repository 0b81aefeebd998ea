#!/bin/bash
# round 6: rocprofv3 --kernel-trace --stats of one warm-up + one timed step of the final tree WITHOUT the shared_prefix leg (the closing pass's trace, r06zz, holds that leg's
# launches too: per-kernel totals there are over ~4.5 steps) — the per-step composition of the step
TAG=${1:-r06zz2}
R=$PWD; mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --workload 1080p --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-calibration --no-shared-prefix-leg --no-video-tail > $R/gpurun_out/${TAG}_bench1080p_prof.log 2>&1
cd $R
python3 tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*/*_results.db gpurun_out/${TAG}_prof/*_results.db 2>/dev/null | head -1) gpurun_out/${TAG}_bench1080p_kernel_stats.txt > /dev/null 2>&1
find gpurun_out/${TAG}_prof -name '*.db' -size +20M -delete 2>/dev/null
head -24 gpurun_out/${TAG}_bench1080p_kernel_stats.txt; tail -3 gpurun_out/${TAG}_bench1080p_kernel_stats.txt; tail -1 gpurun_out/${TAG}_bench1080p_prof.log | cut -c1-300
