#!/bin/bash
# round 6, call v: one whole video end to end through WanT2V.generate at the metric's configuration (50 steps), after a 2-step run of the same tool
TAG=${1:-r06v}
mkdir -p gpurun_out
python tools/e2e_video.py --steps 2 > gpurun_out/${TAG}_e2e_2steps.json.log 2> gpurun_out/${TAG}_e2e_2steps.err || { tail -20 gpurun_out/${TAG}_e2e_2steps.err; exit 1; }
tail -1 gpurun_out/${TAG}_e2e_2steps.json.log
python tools/e2e_video.py --steps 50 > gpurun_out/${TAG}_e2e_video_1080p.json.log 2> gpurun_out/${TAG}_e2e_video_1080p.err
tail -1 gpurun_out/${TAG}_e2e_video_1080p.json.log; tail -3 gpurun_out/${TAG}_e2e_video_1080p.err
