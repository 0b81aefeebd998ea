#!/bin/bash
# round-5 pass I: variant 12 with the k-tile stream running through the tile boundary (bf16 outputs) — bits, then timing: 232 = split, 240 = continuous (flag 8), 11
S=moviigen1.1_amd/lib/mg_selftest
OUT=gpurun_out/${1:-r05i}_gemm_v12_cont.log
: > $OUT
for args in "16384 5120 1024 0" "16384 5120 1024 1" "4000 2304 128 0" "33000 2560 192 1" "33000 2560 192 0" "16384 5120 1024 2" "20000 13824 5120 1" "700 512 128 0" "131040 5120 5120 0"; do
  echo "== gemmdiff 232 $args" >> $OUT
  timeout 200 $S gemmdiff 232 $args 2>&1 | grep -E "differ|\(m " >> $OUT || echo "FAIL rc=$?" >> $OUT
done
timeout 200 $S gemmab1 131040 5120 5120 0 2 11 232 240 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 15360 5120 0 2 11 232 240 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 13824 5120 1 2 11 232 240 2>&1 | grep -v "^device" >> $OUT
grep -E "differ|FAIL|gemm_ab|TFLOP|\(m " $OUT | sed 's/  \[SAME\] variant/ v/; s/  \[DIFF\] variant/ DIFF v/; s/elements differ/diff/' | cut -c1-100
