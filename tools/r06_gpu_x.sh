#!/bin/bash
# round 6, call x: GEMM variant 12 without its per-tile scratch round trips (piece offsets and epilogue rows from a fresh lane index: 216 / 196 / 212 / 236 -> 8 / 0 / 0 / 36
# bytes of scratch) against the build before (ab_old/): the five GEMMs of a block at M = 131 040, alternating three times; then the GEMM parity tests
TAG=${1:-r06x}
mkdir -p gpurun_out
{
for r in 1 2 3; do
  for B in ab_old moviigen1.1_amd/lib; do
    echo "== $B  gemmshapes 12 131040 (round $r)"; timeout 600 $B/mg_selftest gemmshapes 12 131040 2>&1 | grep -i "tflop\|fail\|pass" | tail -8
  done
done
} > gpurun_out/${TAG}_gemm_no_scratch.log 2>&1
(python -m pytest tests -q -m gpu -x -k "gemm or dit_forward or block_composition or t5" 2>&1 | tail -6) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_gemm_no_scratch.log; tail -4 gpurun_out/${TAG}_pytest.log
