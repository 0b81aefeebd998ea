R=$PWD; mkdir -p gpurun_out
(python -m pytest tests -q -m gpu -k "vae or pipeline or smoke or launcher" 2>&1 | tail -30) > gpurun_out/r02e_pytest_gpu.log
python tools/bench_vae.py --conv-variant 1 > gpurun_out/r02e_vae_1080p_v1.json.log 2>&1
python tools/bench_vae.py --conv-variant 2 > gpurun_out/r02e_vae_1080p_v2.json.log 2>&1
python tools/bench_vae.py --conv-variant 2 --size 1280x720 > gpurun_out/r02e_vae_720p_v2.json.log 2>&1
bash tools/pmc_vae.sh r02e_pmc_vae 9 > gpurun_out/r02e_pmc_vae.log 2>&1
tail -6 gpurun_out/r02e_pytest_gpu.log; tail -qn1 gpurun_out/r02e_vae_*.json.log; grep -A24 "== vae_conv2_kernel<3>" gpurun_out/r02e_pmc_vae/summary.txt
