#!/bin/bash
# GEMM variant 12, s_memtime build: per-workgroup start / end (is there a tail?) and what a tile costs outside its k-loop   bash tools/r05_gpu_u.sh <tag>
tag=${1:-r05u}
mkdir -p gpurun_out
out=gpurun_out/${tag}_gemm_balance.log
: > $out
st=moviigen1.1_amd/lib/mg_selftest
for shape in "131040 5120 5120 0" "131040 5120 5120 2" "131040 15360 5120 0" "131040 5120 13824 2"; do
  echo "== gemmprof 12 $shape" >> $out
  timeout 120 $st gemmprof 12 $shape 2>&1 | grep -v "^wave [123]" >> $out
done
tail -70 $out
