#!/bin/bash
# round 6, call i: the GELU epilogue with one v_exp_f32 + one v_rcp_f32 per value (new build) against the round-5 form (IEEE division; ab_old/ = the
# A/B library and selftest built from the tree before the change), same box, alternating; then the parity tests that touch it
TAG=${1:-r06i}
mkdir -p gpurun_out
{
for r in 1 2 3; do
  echo "== old build, ffn.0 (13824, 5120, epi 1)"; timeout 300 ab_old/mg_selftest gemmab1 131040 13824 5120 1 2 200 | grep variant
  echo "== new build, ffn.0 (13824, 5120, epi 1)"; timeout 300 moviigen1.1_amd/lib/mg_selftest gemmab1 131040 13824 5120 1 2 200 | grep variant
done
echo "== new build gemmprof ffn.0"; timeout 300 moviigen1.1_amd/lib/mg_selftest gemmprof 200 131040 13824 5120 1 | grep -v "^  XCD" | head -8
echo "== old build gemmprof ffn.0"; timeout 300 ab_old/mg_selftest gemmprof 200 131040 13824 5120 1 | grep -v "^  XCD" | head -8
} > gpurun_out/${TAG}_gelu_ab.log 2>&1
(python -m pytest tests -q -m gpu -x -k "gemm or dit_forward or block_composition or dit_real_width or t5_encoder or text" 2>&1 | tail -8) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_gelu_ab.log; tail -5 gpurun_out/${TAG}_pytest.log
