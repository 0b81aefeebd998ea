#!/bin/bash
# round-5 pass K: variant 12 with the single-block k-loop (two passes through one body copy, state advanced inside the body): bits, then timing
S=moviigen1.1_amd/lib/mg_selftest
OUT=gpurun_out/${1:-r05k}_gemm_v12.log
V="${V12_VARIANTS:-232}"
: > $OUT
for v in $V; do
for args in "16384 5120 1024 0" "16384 5120 1024 1" "16384 5120 1024 2" "16384 5120 1024 3" "4000 2304 128 0" "4000 2304 128 2" "33000 2560 192 1" "33000 2500 192 2" "20000 13824 5120 1" "700 512 128 0" "131040 5120 5120 2"; do
  echo "== gemmdiff $v $args" >> $OUT
  timeout 200 $S gemmdiff $v $args 2>&1 | grep -E "differ|\(m " >> $OUT || echo "FAIL rc=$?" >> $OUT
done
done
timeout 200 $S gemmab1 131040 5120 5120 0 2 11 $V 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 15360 5120 0 2 11 $V 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 13824 5120 1 2 11 $V 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 5120 13824 2 2 11 $V 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 5120 5120 2 2 11 $V 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 5120 40960 0 1 11 $V 2>&1 | grep -v "^device" >> $OUT
grep -E "differ|FAIL|gemm_ab|TFLOP|\(m " $OUT | sed 's/  \[SAME\] variant/ v/; s/  \[DIFF\] variant/ DIFF v/; s/elements differ/diff/' | cut -c1-100
