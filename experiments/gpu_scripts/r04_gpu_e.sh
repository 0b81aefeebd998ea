#!/bin/bash
# round-4 GPU pass E: the 256-voxel tile of the wide VAE convolutions — parity, then decode time 128 / 256 / auto alternating
TAG=${1:-r04e}
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -q -m gpu -k "vae or reserved_cus" 2>&1 | tail -15) > gpurun_out/${TAG}_pytest_sel.log
: > gpurun_out/${TAG}_vae_tiles.log
for t in 128 256 auto 128 256 auto; do
  timeout 300 python tools/bench_vae.py --chunk 4 --tile $t 2>/dev/null | grep vae_decode_sec >> gpurun_out/${TAG}_vae_tiles.log
done
timeout 300 python tools/bench_vae.py --chunk 4 --tile auto --stages > gpurun_out/${TAG}_vae_stages.txt 2>&1
tail -4 gpurun_out/${TAG}_pytest_sel.log; cut -c1-200 gpurun_out/${TAG}_vae_tiles.log; grep -A24 "^stage " gpurun_out/${TAG}_vae_stages.txt | head -26
