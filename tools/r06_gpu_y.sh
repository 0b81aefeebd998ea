#!/bin/bash
# round 6, call y: the GEMM parity tests on the build without the per-tile scratch round trips (the fp32-store instantiation keeps its old code), and the per-tile
# account of both builds (gemmprof 200 = variant 12, PROF): last k-tile / wait / epilogue / tile prologue
TAG=${1:-r06y}
mkdir -p gpurun_out
(python -m pytest tests -q -m gpu -k "gemm or dit_forward or block_composition or t5 or small_fp32" 2>&1 | tail -6) > gpurun_out/${TAG}_pytest.log
{
for B in ab_old moviigen1.1_amd/lib; do
  echo "== $B gemmprof variant 12  M=131040 N=15360 K=5120 epi=0"; timeout 300 $B/mg_selftest gemmprof 200 131040 15360 5120 0 | grep -v "^  XCD" | head -8
  echo "== $B gemmprof variant 12  M=131040 N=5120 K=5120 epi=2"; timeout 300 $B/mg_selftest gemmprof 200 131040 5120 5120 2 | grep -v "^  XCD" | head -8
done
} > gpurun_out/${TAG}_gemmprof.log 2>&1
cat gpurun_out/${TAG}_gemmprof.log; tail -4 gpurun_out/${TAG}_pytest.log
