#!/bin/bash
# round 6, call g: the wave-per-row RMS-norm + RoPE kernel and the direct K pack — tests, then the row-wise kernels' rates at the metric's size
TAG=${1:-r06g}
mkdir -p gpurun_out
(python -m pytest tests -q -m gpu -x -k "rmsnorm or rowwise or dit_forward or smoke or block_composition or pipeline_cfg1 or dit_context or sequence_parallel_two" 2>&1 | tail -15) > gpurun_out/${TAG}_pytest.log
(python tools/bench_rowwise.py) > gpurun_out/${TAG}_rowwise.log 2>&1
tail -6 gpurun_out/${TAG}_pytest.log; cat gpurun_out/${TAG}_rowwise.log
