#!/bin/bash
# round-5 pass B: GEMM variant 12 — every element against variant 8 (several shapes / epilogues / k-tile counts), then timing against 11, then s_memtime
S=moviigen1.1_amd/lib/mg_selftest
OUT=gpurun_out/${1:-r05b}_gemm_v12.log
: > $OUT
for args in "16384 5120 1024 0" "16384 5120 1024 1" "16384 5120 1024 2" "16384 5120 1024 3" "4000 2304 128 2" "33000 2560 192 0" "33000 2500 192 2" "33000 2500 192 3" "20000 13824 5120 1"; do
  echo "== gemmdiff 12 $args" >> $OUT
  timeout 120 $S gemmdiff 12 $args 2>&1 | tail -8 >> $OUT || echo "TIMEOUT/FAIL rc=$?" >> $OUT
done
if grep -q "TIMEOUT" $OUT; then tail -30 $OUT; exit 1; fi
echo "== gemmab 131040 2 11 12" >> $OUT
timeout 200 $S gemmab 131040 2 11 12 2>&1 | grep -v "^device" >> $OUT
echo "== gemmab1 q|k|v, ffn.0" >> $OUT
timeout 100 $S gemmab1 131040 15360 5120 0 2 11 12 2>&1 | grep -v "^device" >> $OUT
timeout 100 $S gemmab1 131040 13824 5120 1 2 11 12 2>&1 | grep -v "^device" >> $OUT
echo "== gemmprof 12 / 11" >> $OUT
timeout 100 $S gemmprof 12 131040 5120 5120 2>&1 | grep -E "wave|per tile" >> $OUT
timeout 100 $S gemmprof 11 131040 5120 5120 2>&1 | grep -E "wave 0|per tile" | head -2 >> $OUT
grep -E "==|differ|TFLOP|gemm_ab|wave 0|per tile|FAIL" $OUT | cut -c1-200 | head -70
