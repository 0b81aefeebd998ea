#!/bin/bash
# round-5 final pass on the final tree: tools/r05_gpu_z.sh (suite, smoke, three bench lines) + the rocprofv3 kernel stats of one 1080p step
TAG=${1:-r05zz}
bash tools/r05_gpu_z.sh $TAG
bash tools/r05_gpu_q.sh $TAG
