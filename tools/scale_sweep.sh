#!/bin/bash
# Scaling sweep of the metric on ONE node with N visible GPUs: bench.py at 1 / 2 / 4 / 8 ranks and the three transports of the exchange.  Since round 6 ONE
# bench.py run measures both layouts of the multi-GPU forward (primary = Ulysses over all ranks, the reference's layout = BASELINE configs[2]; `other_layout` =
# cond / uncond halves x Ulysses N/2) and starts with the preflight (ranks, peer access, a 64 MiB all-to-all; with --transport peer_copy also the IPC windows and
# the same exchange as peer copies -> link_gbps_measured, transport_recommended).  One JSON line per run is appended to gpurun_out/scale_sweep_<tag>.jsonl, and a
# table is printed at the end.  Nothing here could be run by the builder (1-GPU boxes); it needs nothing but visible GPUs.
# usage: tools/scale_sweep.sh [tag] [steps] [workload] [extra bench flags, e.g. "--vae-parallel"]
TAG=${1:-r06}; STEPS=${2:-3}; WL=${3:-1080p}; EXTRA=${4:-}
OUT=gpurun_out/scale_sweep_${TAG}.jsonl
mkdir -p gpurun_out; : > $OUT
NG=$(python -c 'import torch; print(torch.cuda.device_count())')
for N in 1 2 4 8; do
  [ "$N" -gt "$NG" ] && break
  for TR in torch rccl_direct peer_copy; do
    [ "$N" = 1 ] && [ "$TR" != torch ] && continue
    echo "== N=$N transport=$TR" >&2
    python bench.py --gpus $N --steps $STEPS --warmup 1 --workload $WL --no-cpu-baseline --no-video-tail --transport $TR $EXTRA 2> gpurun_out/scale_sweep_${TAG}_N${N}_${TR}.err \
      | grep '^{' | python -c "import sys, json; d = json.loads(sys.stdin.readline()); d['sweep'] = {'n': $N, 'transport_env': '$TR'}; print(json.dumps(d))" >> $OUT \
      || echo "{\"sweep\": {\"n\": $N, \"transport_env\": \"$TR\"}, \"failed\": true}" >> $OUT
  done
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = next((r['value'] for r in rows if r.get('n_gpus') == 1 and not r.get('failed')), None)
print(f'{"N":>2s} {"transport":12s} {"layout":28s} {"steps/s":>9s} {"s/step":>8s} {"speedup":>8s} {"eff":>6s} {"hidden":>7s} {"a2a GB/s":>9s} {"copy GB/s":>9s} {"recommended":>11s}')
for r in rows:
    s = r['sweep']
    if r.get('failed'):
        print(f'{s["n"]:2d} {s["transport_env"]:12s} FAILED (see gpurun_out/*.err)')
        continue
    pf = r.get('preflight') or {}
    g = pf.get('link_gbps_measured') or {}
    for lay in (r, r.get('other_layout')):
        if not lay:
            continue
        name = lay.get('parallelism') or r['config']['parallelism']
        sp = lay['value'] / base if base else float('nan')
        ov = (lay.get('overlap') or {}).get('hidden_frac')
        print(f'{s["n"]:2d} {s["transport_env"]:12s} {name:28s} {lay["value"]:9.4f} {lay["ms_per_step"] / 1e3:8.2f} {sp:8.2f} {sp / s["n"]:6.2f} '
              f'{("%.2f" % ov) if ov is not None else "-":>7s} {("%.1f" % g["all_to_all"]) if "all_to_all" in g else "-":>9s} '
              f'{("%.1f" % g["peer_copy"]) if "peer_copy" in g else "-":>9s} {pf.get("transport_recommended", "-"):>11s}')
PY
