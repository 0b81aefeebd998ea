#!/bin/bash
# variant 11: paired 16-byte epilogue (bf16 outputs) + row-wise epilogue through LDS (fp32 outputs; 126 = the direct one)
TAG=${1:-r04s}
S=moviigen1.1_amd/lib/mg_selftest
{
timeout 300 $S gemmab 131040 2 8 126 11
for e in 2 3; do timeout 120 $S gemmv 11 4200 4100 640 $e | tail -1; timeout 120 $S gemmv 11 1030 1284 640 $e | tail -1; timeout 120 $S gemmv 11 9000 2304 64 $e | tail -1; done
timeout 120 $S gemmv 11 131040 5120 13824 2 | tail -1
} > gpurun_out/${TAG}_gemm_v11.log 2>&1
cat gpurun_out/${TAG}_gemm_v11.log
