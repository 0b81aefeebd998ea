#!/bin/bash
# round 6, closing pass on the FINAL tree (after the attention / GEMM scratch work and WanModel.forward_pair): tools/round_end_gpu.sh (GPU suite, smoke, the 1080p bench line with
# live PMC traffic, rocprofv3 kernel stats of one step, FETCH / WRITE passes) + the 720p and 1056p bench lines + one whole 720p video end to end (BASELINE configs[1])
TAG=${1:-r06zz}
bash tools/round_end_gpu.sh $TAG 1080p
python bench.py --workload 720p --steps 2 --warmup 1 --no-pmc > gpurun_out/${TAG}_bench720p.json.log 2>&1
python bench.py --workload 1056p --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > gpurun_out/${TAG}_bench1056p.json.log 2>&1
python tools/e2e_video.py --workload 720p --steps 50 > gpurun_out/${TAG}_e2e_video_720p.json.log 2> gpurun_out/${TAG}_e2e_video_720p.err
tail -1 gpurun_out/${TAG}_bench720p.json.log | cut -c1-400; tail -1 gpurun_out/${TAG}_bench1056p.json.log | cut -c1-400; tail -1 gpurun_out/${TAG}_e2e_video_720p.json.log
