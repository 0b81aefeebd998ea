R=$PWD; mkdir -p gpurun_out
(python -m pytest tests -q -m gpu -k "vae or gemm or rccl or pipeline or smoke or launcher or bench_multirank" 2>&1 | tail -30) > gpurun_out/r02d_pytest_gpu.log
python tools/bench_vae.py > gpurun_out/r02d_vae_1080p.json.log 2>&1
python tools/bench_vae.py --size 1280x720 > gpurun_out/r02d_vae_720p.json.log 2>&1
cd /tmp; export TMPDIR=/tmp
rm -f $R/gpurun_out/r02d_sp_overlap.txt
for cfg in default hwq8 direct direct_hwq8; do
  unset GPU_MAX_HW_QUEUES MOVIIGEN_SP_TRANSPORT
  case $cfg in hwq8) export GPU_MAX_HW_QUEUES=8;; direct) export MOVIIGEN_SP_TRANSPORT=rccl_direct;; direct_hwq8) export GPU_MAX_HW_QUEUES=8 MOVIIGEN_SP_TRANSPORT=rccl_direct;; esac
  rm -rf $R/gpurun_out/r02d_sp_trace
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02d_sp_trace -o sp -- python $R/tools/sp_overlap_trace.py run > $R/gpurun_out/r02d_sp_trace_$cfg.log 2>&1
  SP_TRACE_LABEL=$cfg python $R/tools/sp_overlap_trace.py analyse $R/gpurun_out/r02d_sp_trace $R/gpurun_out/r02d_sp_overlap.txt > /dev/null
done
unset GPU_MAX_HW_QUEUES MOVIIGEN_SP_TRANSPORT
rm -rf $R/gpurun_out/r02d_sp_trace
cd $R
tail -8 gpurun_out/r02d_pytest_gpu.log; cat gpurun_out/r02d_vae_1080p.json.log gpurun_out/r02d_vae_720p.json.log | tail -2; grep -v "^  " gpurun_out/r02d_sp_overlap.txt; grep -h "SP_OVERLAP_RUN_OK\|Error\|error" gpurun_out/r02d_sp_trace_*.log | head
