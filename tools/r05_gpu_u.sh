#!/bin/bash
# per-workgroup start / end of GEMM variant 12 (static tile assignment): is there a tail?   bash tools/r05_gpu_u.sh <tag>
tag=${1:-r05u}
mkdir -p gpurun_out
out=gpurun_out/${tag}_gemm_balance.log
: > $out
export LD_LIBRARY_PATH=$PWD/moviigen1.1_amd/lib:$LD_LIBRARY_PATH
st=moviigen1.1_amd/lib/mg_selftest
for shape in "131040 5120 5120" "131040 15360 5120" "131040 5120 13824"; do
  echo "== gemmprof 12 $shape" >> $out
  timeout 120 $st gemmprof 12 $shape 2>&1 | grep -v "^wave\|^        per tile" >> $out
  timeout 120 $st gemmprof 12 $shape 2>&1 | grep "^wave 0\|^        per tile" | head -2 >> $out
done
tail -45 $out
