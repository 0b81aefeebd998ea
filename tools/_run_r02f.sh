R=$PWD; mkdir -p gpurun_out
(python -m pytest tests -q -m gpu -k "vae or pipeline or smoke or launcher" 2>&1 | tail -30) > gpurun_out/r02f_pytest_gpu.log
python tools/bench_vae.py > gpurun_out/r02f_vae_1080p.json.log 2>&1
python tools/bench_vae.py --size 1280x720 > gpurun_out/r02f_vae_720p.json.log 2>&1
bash tools/pmc_vae.sh r02f_pmc_vae 9 > gpurun_out/r02f_pmc_vae.log 2>&1
tail -6 gpurun_out/r02f_pytest_gpu.log; tail -qn1 gpurun_out/r02f_vae_*.json.log; grep -A24 "== vae_conv_kernel<3>" gpurun_out/r02f_pmc_vae/summary.txt | grep "==\|INSTS\|derived"
