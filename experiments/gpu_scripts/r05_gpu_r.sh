#!/bin/bash
# round-5 pass R: the fp32 epilogue with a rolling window of residual loads (232) against the round-4 form (233 = flag 1) and variant 11
S=moviigen1.1_amd/lib/mg_selftest
OUT=gpurun_out/${1:-r05r}_gemm_rows2.log
: > $OUT
for args in "16384 5120 1024 2" "16384 5120 1024 3" "4000 2304 128 2" "33000 2500 192 2" "33000 2500 192 3" "20000 5120 13824 2" "131040 5120 5120 2"; do
  echo "== gemmdiff 232 $args" >> $OUT
  timeout 200 $S gemmdiff 232 $args 2>&1 | grep -E "differ|\(m " >> $OUT || echo "FAIL rc=$?" >> $OUT
done
timeout 300 $S gemmab1 131040 5120 5120 2 3 11 233 232 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 5120 13824 2 2 11 233 232 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 5120 5120 3 2 233 232 2>&1 | grep -v "^device" >> $OUT
grep -E "differ|FAIL|gemm_ab|TFLOP|\(m " $OUT | sed 's/  \[SAME\] variant/ v/; s/  \[DIFF\] variant/ DIFF v/; s/elements differ/diff/' | cut -c1-100
