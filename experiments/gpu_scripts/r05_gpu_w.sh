#!/bin/bash
# m16 attention: items by ticket (variant 10) against the static per-XCD partition (variant 18 = debug bit 4)   bash tools/r05_gpu_w.sh <tag>
tag=${1:-r05w}
mkdir -p gpurun_out
out=gpurun_out/${tag}_attn_tickets.log
: > $out
st=moviigen1.1_amd/lib/mg_selftest
echo "== selftest attn (small + ragged shapes, both entries)" >> $out
timeout 300 $st attn 2>&1 | grep -E "PASS|FAIL|SELFTEST" | grep -v "variant 3" | head -30 >> $out
echo "== attnab 33000 x 2 heads, data 2 (ragged tail, sink data)" >> $out
timeout 200 $st attnab 33000 2 2 1 10 18 2>&1 | grep -E "PASS|FAIL|DIFF" >> $out
echo "== attnab 131040 x 40 heads (the metric's launch): 10 = tickets, 18 = static partition" >> $out
timeout 500 $st attnab 131040 40 0 3 10 18 2>&1 | grep -E "PASS|FAIL|DIFF|SELFTEST" >> $out
echo "== attnab 131040 x 8 heads" >> $out
timeout 300 $st attnab 131040 8 0 3 10 18 2>&1 | grep -E "PASS|FAIL|DIFF|SELFTEST" >> $out
echo "== w64prof (tickets): per-workgroup end times" >> $out
timeout 400 $st w64prof 131040 40 0 1 0 131040 2>&1 | grep -v "^wave [123]" >> $out
tail -60 $out
