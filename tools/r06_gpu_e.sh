#!/bin/bash
# round 6, call e: the W-band VAE decode — one rank of 2 / 4 / 8 emulated at 1920x832x81f (tools/bench_vae.py --bands), the bench.py N > 1 code path with
# the new layouts / preflight / --vae-parallel (gloo ranks on one GPU), the bf16x3 error at a BASELINE size
TAG=${1:-r06e}
mkdir -p gpurun_out
(timeout 900 python tools/bench_vae.py --chunk 4 --bands 2 4 8) > gpurun_out/${TAG}_vae_bands.log 2>&1
(python -m pytest tests -q -m gpu -x -s -k "bench_multirank or fullsize_vae_decode_config5 or launcher_end_to_end" 2>&1 | tail -30) > gpurun_out/${TAG}_pytest.log
grep -v "^$" gpurun_out/${TAG}_vae_bands.log | tail -8; tail -12 gpurun_out/${TAG}_pytest.log
