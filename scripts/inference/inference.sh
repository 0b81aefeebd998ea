#!/bin/bash
# 4-GPU launch, the reference's scripts/inference/inference.sh with this engine's rendezvous address
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29500 \
  "$(dirname "$0")/generate.py" --task t2v-14B --size 1280*720 --sample_steps 50 --ckpt_dir ./MoviiGen1.1 \
  --dit_fsdp --t5_fsdp --ulysses_size 4 --ring_size 1 --sample_shift 5 --sample_guide_scale 5.0 \
  --prompt "A cat walks on the grass, realistic style." --base_seed 42 --frame_num 81
