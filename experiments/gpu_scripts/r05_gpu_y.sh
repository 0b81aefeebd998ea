#!/bin/bash
# GEMM variant 12, gated-residual epilogue: what it waits for.  200 = as shipped, 204 = no stores, 328 = no residual loads, 332 = neither, 208 = touch loads,
# (r05y3: 456 = nt stores, 712 = nt loads, 968 = both; since then nt is the default and 456 / 712 / 968 switch it OFF)  (204 / 328 / 332 are timing-only: DIFF is expected)   bash tools/r05_gpu_y.sh <tag> [variants...]
tag=${1:-r05y}; shift
vars=${@:-200 456 712 968}
mkdir -p gpurun_out
out=gpurun_out/${tag}_gemm_epilogue_parts.log
: > $out
st=moviigen1.1_amd/lib/mg_selftest
for shape in "131040 5120 5120 2" "131040 5120 13824 2"; do
  echo "== gemmab1 $shape: $vars" >> $out
  timeout 300 $st gemmab1 $shape 2 $vars 2>&1 | grep -E "SAME|DIFF|FAIL" >> $out
done
tail -40 $out
