#!/bin/bash
# per-workgroup start / end of the m16 attention (static item partition per XCD): is there a tail?   bash tools/r05_gpu_v.sh <tag>
tag=${1:-r05v}
mkdir -p gpurun_out
out=gpurun_out/${tag}_attn_balance.log
: > $out
st=moviigen1.1_amd/lib/mg_selftest
# w64prof Lk heads variant prescaled dbg Lq
echo "== w64prof: L = 131040, 40 heads (the metric's launch), pre-scaled entry" >> $out
timeout 400 $st w64prof 131040 40 0 1 0 131040 2>&1 | grep -v "^wave [123]" >> $out
tail -40 $out
