#!/bin/bash
# round-4 GPU pass B: whole parity suite, attention robustness table, VAE stage times, SP / FSDP overlap traces
TAG=${1:-r04b}
R=$PWD; mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -60) > gpurun_out/${TAG}_pytest_gpu.log
S=moviigen1.1_amd/lib/mg_selftest
for d in 0 1 2 3 4 5; do
  timeout 300 $S attnab 131040 8 $d 2 10 0 > gpurun_out/${TAG}_attnab_data$d.log 2>&1
done
MG_ATTN_RESERVE_CUS=8 timeout 300 $S attnab 131040 8 0 2 10 > gpurun_out/${TAG}_attnab_reserve8.log 2>&1
timeout 600 python tools/bench_vae.py --chunk 4 --stages > gpurun_out/${TAG}_vae_stages.txt 2>&1
export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_sp_overlap.txt; : > $OUT
for tr in "" rccl_direct; do
 for rs in 0 8; do
  D=/tmp/spov_${tr:-torch}_$rs; rm -rf $D
  (cd /tmp && SP_TRACE_SHAPE=cfg2 MOVIIGEN_SP_TRANSPORT=$tr MOVIIGEN_SP_RESERVE_CUS=$rs GPU_MAX_HW_QUEUES=8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -o t -- python $R/tools/sp_overlap_trace.py run 2>&1 | grep "SP_OVERLAP_RUN_OK\|Error\|error" >> $OUT)
  SP_TRACE_SHAPE=cfg2 MOVIIGEN_SP_TRANSPORT=$tr MOVIIGEN_SP_RESERVE_CUS=$rs SP_TRACE_LABEL="${tr:-torch}, reserve $rs" python tools/sp_overlap_trace.py analyse $D $OUT > /dev/null
  rm -rf $D
 done
done
(cd /tmp && SP_TRACE_SHAPE=fsdp GPU_MAX_HW_QUEUES=8 timeout 600 python $R/tools/sp_overlap_trace.py run 2>&1 | grep "FSDP_OVERLAP_RUN_OK\|Error" >> $OUT)
tail -15 gpurun_out/${TAG}_pytest_gpu.log; grep -h "variant\|attn_ab" gpurun_out/${TAG}_attnab_*.log; tail -9 gpurun_out/${TAG}_vae_stages.txt; grep -v "^  " $OUT
