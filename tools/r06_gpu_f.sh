#!/bin/bash
# round 6, call f: joules per launch of the attention kernel, its A/B partners and the GEMM variants (tools/energy_table.py)
TAG=${1:-r06f}
mkdir -p gpurun_out
(timeout 1500 python tools/energy_table.py --secs 6) > gpurun_out/${TAG}_energy_table.log 2>&1
tail -30 gpurun_out/${TAG}_energy_table.log
