#!/bin/bash
# round-4 GPU pass F: the whole parity suite + smoke on the final tree
TAG=${1:-r04f}
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12) > gpurun_out/${TAG}_pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1
tail -6 gpurun_out/${TAG}_pytest_gpu.log; tail -1 gpurun_out/${TAG}_smoke.log
