R=$PWD; mkdir -p gpurun_out
(python -m pytest tests -q -m gpu -k "vae or pipeline or smoke or launcher" 2>&1 | tail -30) > gpurun_out/r02g_pytest_gpu.log
python tools/bench_vae.py > gpurun_out/r02g_vae_1080p.json.log 2>&1
python tools/bench_vae.py --chunk 4 > gpurun_out/r02g_vae_1080p_chunk4.json.log 2>&1
python tools/bench_vae.py --size 1280x720 --chunk 4 > gpurun_out/r02g_vae_720p_chunk4.json.log 2>&1
bash tools/pmc_vae.sh r02g_pmc_vae 9 > gpurun_out/r02g_pmc_vae.log 2>&1
tail -6 gpurun_out/r02g_pytest_gpu.log; tail -qn1 gpurun_out/r02g_vae_*.json.log; grep -A24 "== vae_conv_kernel<3>" gpurun_out/r02g_pmc_vae/summary.txt | grep "==\|INSTS\|derived"
