#!/bin/bash
# round-5 pass G: load placement in variant 12's body (232 + 32 s = body s): the k-loop alone (K = 40960), q|k|v, cross-q
S=moviigen1.1_amd/lib/mg_selftest
OUT=gpurun_out/${1:-r05g}_gemm_v12_sched.log
: > $OUT
V="${V12_VARIANTS:-232 264 296 328 360 392}"
for v in $V; do
  echo "== gemmdiff $v 16384 5120 1024 2" >> $OUT
  timeout 100 $S gemmdiff $v 16384 5120 1024 2 2>&1 | grep differ >> $OUT || echo "FAIL rc=$?" >> $OUT
done
timeout 300 $S gemmab1 131040 5120 40960 0 1 11 $V 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 15360 5120 0 2 $V 2>&1 | grep -v "^device" >> $OUT
timeout 200 $S gemmab1 131040 5120 5120 0 2 $V 2>&1 | grep -v "^device" >> $OUT
grep -E "differ|FAIL|gemm_ab|TFLOP" $OUT | sed 's/  \[SAME\] variant/ v/; s/elements differ/diff/' | cut -c1-90
