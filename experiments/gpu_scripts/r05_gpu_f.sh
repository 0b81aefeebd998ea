#!/bin/bash
# round-5 pass F: the k-loop alone (K = 40960: 640 k-tiles per output tile) — variant 12 against the vendor kernel, timing and SQ counters
TAG=${1:-r05f}
export MG_GEMM_VARIANT=${MG_GEMM_VARIANT:-232} MG_LIB_GEMM_SHAPES="5120,40960"
python tools/bench_lib_gemm.py 131040 2>&1 | grep shape | cut -c1-300 > gpurun_out/${TAG}_kloop.log
PMC_SETS="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" timeout 600 bash tools/pmc_lib_gemm.sh ${TAG} > /dev/null 2>&1
cat gpurun_out/${TAG}_kloop.log; cut -c1-4,60-150 gpurun_out/${TAG}_pmc_lib_gemm.txt
