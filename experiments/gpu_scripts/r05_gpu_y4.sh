#!/bin/bash
# GEMM variant 12, bf16 epilogues: r05y4 ran 200 = plain stores against 1224 = non-temporal stores (since then the default; 1224 now switches the hint OFF)   bash tools/r05_gpu_y4.sh <tag>
tag=${1:-r05y4}
mkdir -p gpurun_out
out=gpurun_out/${tag}_gemm_pair_nt.log
: > $out
st=moviigen1.1_amd/lib/mg_selftest
for shape in "131040 5120 5120 0" "131040 15360 5120 0" "131040 13824 5120 1"; do
  echo "== gemmab1 $shape: 200 1224" >> $out
  timeout 300 $st gemmab1 $shape 3 200 1224 2>&1 | grep -E "SAME|DIFF|FAIL" >> $out
done

tail -30 $out
