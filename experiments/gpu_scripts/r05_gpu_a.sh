#!/bin/bash
# round-5 pass A: (1) what the firmware says holds the clock under each hot kernel (tools/gpu_telemetry.py: amdsmi violation accumulators
# while one kernel runs back to back), (2) the vendor library against GEMM variant 11 on this box, (3) the PMC passes of variant 11
TAG=${1:-r05a}
S=moviigen1.1_amd/lib/mg_selftest
OUT=gpurun_out/${TAG}_telemetry.log
: > $OUT
python tools/gpu_telemetry.py --raw --label idle -- sleep 2 >> $OUT 2>&1
python tools/gpu_telemetry.py --label attn -- timeout 100 $S powerloop attn 0 9 >> $OUT 2>&1
python tools/gpu_telemetry.py --label gemm11 -- timeout 100 $S powerloop gemm 11 9 >> $OUT 2>&1
python tools/gpu_telemetry.py --label gemm8 -- timeout 100 $S powerloop gemm 8 6 >> $OUT 2>&1
python tools/gpu_telemetry.py --label attn_b -- timeout 100 $S powerloop attn 0 9 >> $OUT 2>&1
python tools/gpu_telemetry.py --label lib_vs_v11 -- timeout 300 python tools/bench_lib_gemm.py 131040 > gpurun_out/${TAG}_lib_gemm.log 2>&1
tail -1 gpurun_out/${TAG}_lib_gemm.log >> $OUT
timeout 600 bash tools/pmc_lib_gemm.sh ${TAG} > /dev/null 2>&1
grep -v raw_metrics $OUT | cut -c1-1200; cat gpurun_out/${TAG}_lib_gemm.log | cut -c1-400; cat gpurun_out/${TAG}_pmc_lib_gemm.txt
