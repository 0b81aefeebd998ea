#!/bin/bash
# round 6, call c: do the gated-residual / store epilogues of GEMM variant 12 get shorter when the 32 workgroups of an XCD do NOT reach them in the
# same microseconds?  (A/B library flags 2048 / 4096: slot s of the XCD starts s x 4 us / (s & 15) x 4 us late.)
TAG=${1:-r06c}
S=moviigen1.1_amd/lib/mg_selftest
mkdir -p gpurun_out
{
for V in 200 2248 4296; do echo "== gemmprof variant $V  M=131040 N=5120 K=5120 epi=2"; timeout 300 $S gemmprof $V 131040 5120 5120 2 | grep -v "^  XCD"; done
for V in 200 2248; do echo "== gemmprof variant $V  M=131040 N=15360 K=5120 epi=0"; timeout 300 $S gemmprof $V 131040 15360 5120 0 | grep -v "^  XCD"; done
echo "== gemmab1 o-proj (5120, 5120, epi 2)";  timeout 600 $S gemmab1 131040 5120 5120 2 3 200 2248 4296
echo "== gemmab1 ffn.2 (5120, 13824, epi 2)";  timeout 600 $S gemmab1 131040 5120 13824 2 3 200 2248 4296
echo "== gemmab1 q|k|v (15360, 5120, epi 0)";  timeout 600 $S gemmab1 131040 15360 5120 0 3 200 2248 4296
echo "== gemmab1 ffn.0 (13824, 5120, epi 1)";  timeout 600 $S gemmab1 131040 13824 5120 1 3 200 2248 4296
} > gpurun_out/${TAG}_gemm_stagger.log 2>&1
cat gpurun_out/${TAG}_gemm_stagger.log
