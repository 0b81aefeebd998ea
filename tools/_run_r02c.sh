R=$PWD; mkdir -p gpurun_out
(python -m pytest tests -q -m gpu 2>&1 | tail -60) > gpurun_out/r02c_pytest_gpu.log
for v in 5 6 5 6; do moviigen1.1_amd/lib/mg_selftest gemmshapes $v 131040 >> gpurun_out/r02c_gemmshapes_v$v.log 2>&1; done
moviigen1.1_amd/lib/mg_selftest gemmshapes 6 75600 > gpurun_out/r02c_gemmshapes_v6_75600.log 2>&1
bash tools/pmc_vae.sh r02c_pmc_vae 9 > gpurun_out/r02c_pmc_vae.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r02c_sp_trace -o sp -- python $R/tools/sp_overlap_trace.py run > $R/gpurun_out/r02c_sp_trace.log 2>&1
cd $R
python tools/sp_overlap_trace.py analyse gpurun_out/r02c_sp_trace gpurun_out/r02c_sp_overlap.txt
rm -rf gpurun_out/r02c_sp_trace
tail -12 gpurun_out/r02c_pytest_gpu.log; grep -h "TFLOP" -B1 gpurun_out/r02c_gemmshapes_v5.log | grep -v "^--" | paste - - | awk '{print "v5",$0}' | cut -c1-150; grep -h "TFLOP" -B1 gpurun_out/r02c_gemmshapes_v6.log | grep -v "^--" | paste - - | awk '{print "v6",$0}' | cut -c1-150; tail -3 gpurun_out/r02c_sp_trace.log; tail -40 gpurun_out/r02c_pmc_vae/summary.txt
