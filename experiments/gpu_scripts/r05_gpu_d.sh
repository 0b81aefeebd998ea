#!/bin/bash
# round-5 pass D: where variant 12's cycles go against the vendor kernel's — SQ wait / issue / LDS counters of both on the same operands
TAG=${1:-r05d}
export MG_GEMM_VARIANT=${MG_GEMM_VARIANT:-232}
PMC_SETS="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" timeout 600 bash tools/pmc_lib_gemm.sh ${TAG} > /dev/null 2>&1
cut -c1-150 gpurun_out/${TAG}_pmc_lib_gemm.txt
