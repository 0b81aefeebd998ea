#!/bin/bash
# round-4 GPU pass A: full parity suite, attention robustness table (attnab data modes), VAE stage times, one bench line
TAG=${1:-r04a}
R=$PWD; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -40) > gpurun_out/${TAG}_pytest_gpu.log
S=moviigen1.1_amd/lib/mg_selftest
for d in 0 1 2 3 4 5; do
  timeout 300 $S attnab 131040 8 $d 2 10 0 > gpurun_out/${TAG}_attnab_data$d.log 2>&1
done
timeout 600 python tools/bench_vae.py --chunk 4 --stages > gpurun_out/${TAG}_vae_stages.txt 2>&1
timeout 900 python bench.py --steps 1 --warmup 1 > gpurun_out/${TAG}_bench1080p.json.log 2>&1
tail -5 gpurun_out/${TAG}_pytest_gpu.log; grep -h "variant\|attn_ab" gpurun_out/${TAG}_attnab_data*.log; tail -12 gpurun_out/${TAG}_vae_stages.txt; tail -c 1500 gpurun_out/${TAG}_bench1080p.json.log
