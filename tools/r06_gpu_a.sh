#!/bin/bash
# round 6, call a: the new N=50 parity tests, then the evidence VERDICT r05 item 2 asks for on ONE box:
#   - SQ / LDS / TCC counter passes of the SHIPPED attention kernel at L = 131 040, 40 heads (mg_selftest attnpmc)
#   - the same passes of GEMM variant 12 on the five block shapes at M = 131 040 (mg_selftest gemmshapes 12)
#   - vendor GEMM vs variant 12, same process (tools/bench_lib_gemm.py)
TAG=${1:-r06a}
mkdir -p gpurun_out
(python -m pytest tests -q -m gpu -x -s -k "scheduler_gpu or pipeline_50 or pipeline_cfg1" 2>&1 | tail -40) > gpurun_out/${TAG}_pytest_new.log
(timeout 900 bash tools/pmc_kernel.sh "attnpmc 131040 40" ${TAG}_pmc_attn) > gpurun_out/${TAG}_pmc_attn_m16.txt 2>&1
(timeout 900 bash tools/pmc_kernel.sh "gemmshapes 12 131040" ${TAG}_pmc_gemm) > gpurun_out/${TAG}_pmc_gemm_v12.txt 2>&1
(timeout 600 python tools/bench_lib_gemm.py 131040) > gpurun_out/${TAG}_lib_gemm.log 2>&1
rm -rf gpurun_out/${TAG}_pmc_attn/p*/ gpurun_out/${TAG}_pmc_gemm/p*/
cat gpurun_out/${TAG}_pytest_new.log | tail -30; cat gpurun_out/${TAG}_pmc_attn_m16.txt gpurun_out/${TAG}_pmc_gemm_v12.txt; tail -12 gpurun_out/${TAG}_lib_gemm.log
