#!/bin/bash
# GEMM variant 12: staggered start of the XCDs (2248 = 4 us per XCD, 4296 = 16 us) against the plain launch (200)   bash tools/r05_gpu_y5.sh <tag>
tag=${1:-r05y5}
mkdir -p gpurun_out
out=gpurun_out/${tag}_gemm_stagger.log
: > $out
st=moviigen1.1_amd/lib/mg_selftest
for shape in "131040 5120 5120 0" "131040 5120 5120 2" "131040 15360 5120 0"; do
  echo "== gemmab1 $shape: 200 2248 4296" >> $out
  timeout 300 $st gemmab1 $shape 2 200 2248 4296 2>&1 | grep -E "SAME|DIFF|FAIL" >> $out
done
cat $out
