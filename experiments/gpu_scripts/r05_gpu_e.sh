#!/bin/bash
# round-5 pass E: what a tile costs outside its k-loop — variant 12 without the stores (flag 4), and the same shape with 8x the k-tiles per tile
S=moviigen1.1_amd/lib/mg_selftest
OUT=gpurun_out/${1:-r05e}_gemm_v12_tilecost.log
: > $OUT
timeout 200 $S gemmab1 131040 5120 5120 0 2 11 232 236 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 5120 40960 0 2 11 232 236 2>&1 | grep -v "^device" >> $OUT
timeout 300 $S gemmab1 131040 5120 40960 2 2 11 232 2>&1 | grep -v "^device" >> $OUT
grep -E "gemm_ab|TFLOP" $OUT | sed 's/  \[SAME\] variant/ v/; s/  \[DIFF\] variant/ DIFF v/' | cut -c1-90
