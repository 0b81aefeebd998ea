#!/bin/bash
# round-4 GPU pass D: SQ / TCC counters of the attention kernel at the metric's sequence length, the live-traffic test
TAG=${1:-r04d}
mkdir -p gpurun_out
bash tools/pmc_kernel.sh "attn1 0 131040" ${TAG}_pmc_attn > gpurun_out/${TAG}_pmc_attn_m16_L131040.txt 2>&1
rm -rf gpurun_out/${TAG}_pmc_attn
(timeout 900 python -m pytest tests -q -m gpu -k "live_traffic or seam_flash" 2>&1 | tail -15) > gpurun_out/${TAG}_pytest_sel.log
grep "attn_hd128_m16" gpurun_out/${TAG}_pmc_attn_m16_L131040.txt | awk '{print $4,$5,$6,$7}'; tail -3 gpurun_out/${TAG}_pytest_sel.log
