#!/bin/bash
# round-5 pass N: more filler placements (10 = the shipped one = 1; 17 = the round-4 placement 0), then clock / power under the new kernel and its cycles
S=moviigen1.1_amd/lib/mg_selftest
OUT=gpurun_out/${1:-r05n}_attn_order.log
timeout 600 $S attnab 131040 8 0 2 17 10 14 15 13 > $OUT 2>&1
python tools/gpu_telemetry.py --label attn_ord1 -- timeout 100 $S powerloop attn 0 8 >> $OUT 2>&1
timeout 200 $S w64prof 75584 8 0 1 >> $OUT 2>&1
grep -E "attn_ab|TFLOP|FAIL|powerloop|wave 0|label" $OUT | cut -c1-420 | head -40
