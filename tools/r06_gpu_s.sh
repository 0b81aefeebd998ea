#!/bin/bash
# round 6, call s: MFMA ORDER inside a group of eight under placement 1's gaps — alternating S P S P (shipped, variant 10) against pairs (26) and quads (42) that
# share their A operand (K / V fragment) in consecutive instructions: does operand hold reduce switching energy under the package-power limit?
TAG=${1:-r06s}
mkdir -p gpurun_out
{
for D in 0 1; do
  echo "== attnab L=131040 heads=8 data=$D variants 10 (S P S P) 26 (pairs) 42 (quads)"
  timeout 600 moviigen1.1_amd/lib/mg_selftest attnab 131040 8 $D 4 10 26 42 | tail -14
done
} > gpurun_out/${TAG}_attn_mfma_order.log 2>&1
cat gpurun_out/${TAG}_attn_mfma_order.log
