#!/bin/bash
# per-kernel durations INSIDE the step on one box: rocprofv3 --kernel-trace --stats of two steps with GEMM variant 8 forced and with the default (11)
TAG=${1:-r04v}
R=$PWD
cd /tmp; export TMPDIR=/tmp
for v in 8 0; do
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$v -o bench -- python $R/bench.py --workload 1080p --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-video-tail --gemm-variant $v > $R/gpurun_out/${TAG}_bench_v$v.log 2>&1
python3 $R/tools/rocprof_summary.py $(ls $R/gpurun_out/${TAG}_prof_$v/*/*_results.db $R/gpurun_out/${TAG}_prof_$v/*_results.db 2>/dev/null | head -1) $R/gpurun_out/${TAG}_kernel_stats_v$v.txt > /dev/null 2>&1
rm -rf $R/gpurun_out/${TAG}_prof_$v
done
cd $R
for v in 8 0; do echo "== gemm variant $v"; tail -1 gpurun_out/${TAG}_bench_v$v.log | cut -c1-200; grep -E "gemm_bf16|attn_hd128_m16|ln_modulate" gpurun_out/${TAG}_kernel_stats_v$v.txt | cut -c1-130; done
