#!/bin/bash
# round-5 pass M: where the softmax fillers stand in the m16 attention step (10 + k = pre-scaled entry on placement k), same operands, alternating
S=moviigen1.1_amd/lib/mg_selftest
OUT=gpurun_out/${1:-r05m}_attn_order.log
V="${ATTN_VARIANTS:-10 11 12 13}"
timeout 600 $S attnab 131040 8 0 3 $V > $OUT 2>&1
timeout 600 $S attnab 131040 8 1 2 $V >> $OUT 2>&1
grep -E "attn_ab|TFLOP|PASS|FAIL|SELFTEST" $OUT | cut -c1-160 | head -60
