#!/bin/bash
# round 6: counter passes of the FINAL attention kernel (the metric's launch) and of GEMM variant 12 on the five block shapes — what r06_pmc_* measured for the first half's kernels
TAG=${1:-r06zz}
mkdir -p gpurun_out
(timeout 900 bash tools/pmc_kernel.sh "attnpmc 131040 40" ${TAG}_pmc_attn) > gpurun_out/${TAG}_pmc_attn_m16.txt 2>&1
(timeout 900 bash tools/pmc_kernel.sh "gemmshapes 12 131040" ${TAG}_pmc_gemm) > gpurun_out/${TAG}_pmc_gemm_v12.txt 2>&1
rm -rf gpurun_out/${TAG}_pmc_attn gpurun_out/${TAG}_pmc_gemm
cat gpurun_out/${TAG}_pmc_attn_m16.txt gpurun_out/${TAG}_pmc_gemm_v12.txt
