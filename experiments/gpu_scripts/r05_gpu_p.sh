#!/bin/bash
# round-5 pass P: the multi-process one-GPU tests touched by the transport change
TAG=${1:-r05p}
(timeout 1500 python -m pytest tests -q -m gpu -x -k "bench_multirank" 2>&1 | tail -80) > gpurun_out/${TAG}_pytest_dist.log
grep -E "Error|error|Traceback|File \"/root/repo|passed|failed" gpurun_out/${TAG}_pytest_dist.log | head -30 | cut -c1-300
