#!/bin/bash
# round 6, call t: the quads order as the shipped default (variant 10) against round 5's order (variant 11 = placement 1, S P S P), then the attention parity tests
TAG=${1:-r06t}
mkdir -p gpurun_out
{
echo "== attnab L=131040 heads=8 data=0 variants 10 (shipped: quads) 11 (S P S P)"
timeout 600 moviigen1.1_amd/lib/mg_selftest attnab 131040 8 0 3 10 11 | tail -8
echo "== w64prof (PROF build of the shipped order) Lk=131040 heads=8"
timeout 300 moviigen1.1_amd/lib/mg_selftest w64prof 131040 8 0 1 0 131040 2>&1 | grep -v "^  XCD" | tail -4
} > gpurun_out/${TAG}_attn_quads_default.log 2>&1
(python -m pytest tests -q -m gpu -x -k "attention or attn or block_composition or dit_forward or ring" 2>&1 | tail -6) > gpurun_out/${TAG}_pytest.log
cat gpurun_out/${TAG}_attn_quads_default.log; tail -4 gpurun_out/${TAG}_pytest.log
