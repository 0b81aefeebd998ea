#!/bin/bash
# round 6, call bb: the attention kernel with its output rows through the LDS (16-byte stores, 4 rows x 256 bytes per instruction) against HEAD before (ab_old/): item phases,
# same-box A/B at the self-attention and cross-attention launch shapes, then the attention parity tests
TAG=${1:-r06bb}
mkdir -p gpurun_out
{
for B in ab_old moviigen1.1_amd/lib; do
  for LK in 512 131040; do H=40; [ $LK = 131040 ] && H=8; echo "== $B  w64prof Lk=$LK heads=$H"; timeout 300 $B/mg_selftest w64prof $LK $H 0 1 0 131040 2>&1 | grep -v "^  XCD" | tail -4; done
done
for r in 1 2 3; do
  for B in ab_old moviigen1.1_amd/lib; do
    echo "== $B  attnab L=131040 heads=8 (self-attention shape, 1/5 launch)"; timeout 300 $B/mg_selftest attnab 131040 8 0 2 10 | tail -3
  done
done
} > gpurun_out/${TAG}_attn_wide_stores.log 2>&1
(python -m pytest tests -q -m gpu -x -k "attention or attn or block_composition or dit_forward or ring" 2>&1 | tail -6) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_attn_wide_stores.log; tail -4 gpurun_out/${TAG}_pytest.log
