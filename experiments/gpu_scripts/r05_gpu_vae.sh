#!/bin/bash
# WanVAE decode with / without the non-temporal hint on the activation stores / residual + normalisation reads: the tree's library against
# lib_alt/libmoviigen_hip_base.so (the same tree without the hint), alternating   bash tools/r05_gpu_vae.sh <tag>
tag=${1:-r05vae}
mkdir -p gpurun_out
out=gpurun_out/${tag}_vae_nt.log
: > $out
run() {   # $1 = library path ('' = the tree's)
python - "$1" <<'PY' 2>&1 | grep -E '"metric"' | python3 -c "import sys,json; [print({k:round(v,4) if isinstance(v,float) else v for k,v in json.loads(l).items() if k in ('value','second_decode_sec','max_abs')}) for l in sys.stdin]"
import os, runpy, sys
root = os.getcwd()
sys.path.insert(0, os.path.join(root, 'moviigen1.1_amd'))
from wan.backend import lib
if sys.argv[1]:
    lib.LIB_PATH = os.path.join(root, sys.argv[1])
sys.argv = ['bench_vae.py']
runpy.run_path(os.path.join(root, 'tools', 'bench_vae.py'), run_name='__main__')
PY
}
for i in 1 2; do
  echo "== base (no hint)" >> $out; run moviigen1.1_amd/lib_alt/libmoviigen_hip_base.so >> $out
  echo "== tree (non-temporal)" >> $out; run "" >> $out
done
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "vae" 2>&1 | tail -2) >> $out
cat $out
