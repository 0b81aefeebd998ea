#!/bin/bash
# Scaling sweep of the metric on ONE node with N visible GPUs: bench.py at 1 / 2 / 4 / 8 ranks, both layouts of the
# multi-GPU forward (cond / uncond halves x Ulysses, and Ulysses over all ranks = the reference's layout) and the three
# transports of the exchange.  One JSON line per run is appended to gpurun_out/scale_sweep_<tag>.jsonl, and a table is
# printed at the end.  Nothing here could be run by the builder (1-GPU boxes); it needs nothing but visible GPUs.
# usage: tools/scale_sweep.sh [tag] [steps] [workload]
TAG=${1:-r03}; STEPS=${2:-3}; WL=${3:-1080p}
OUT=gpurun_out/scale_sweep_${TAG}.jsonl
mkdir -p gpurun_out; : > $OUT
NG=$(python -c 'import torch; print(torch.cuda.device_count())')
for N in 1 2 4 8; do
  [ "$N" -gt "$NG" ] && break
  for LAYOUT in cfg ulysses; do
    [ "$N" = 1 ] && [ "$LAYOUT" = ulysses ] && continue
    for TR in torch rccl_direct peer_copy; do
      [ "$N" = 1 ] && [ "$TR" != torch ] && continue
      [ "$N" = 2 ] && [ "$LAYOUT" = cfg ] && [ "$TR" != torch ] && continue      # cfg2 on 2 ranks has no exchange
      EXTRA=""; [ "$LAYOUT" = ulysses ] && EXTRA="--no-cfg-parallel"
      if [ "$TR" = torch ]; then unset MOVIIGEN_SP_TRANSPORT; else export MOVIIGEN_SP_TRANSPORT=$TR; fi
      echo "== N=$N layout=$LAYOUT transport=$TR" >&2
      python bench.py --gpus $N --steps $STEPS --warmup 1 --workload $WL --no-cpu-baseline --no-video-tail $EXTRA 2> gpurun_out/scale_sweep_${TAG}_N${N}_${LAYOUT}_${TR}.err \
        | grep '^{' | python -c "import sys, json; d = json.loads(sys.stdin.readline()); d['sweep'] = {'n': $N, 'layout': '$LAYOUT', 'transport_env': '$TR'}; print(json.dumps(d))" >> $OUT \
        || echo "{\"sweep\": {\"n\": $N, \"layout\": \"$LAYOUT\", \"transport_env\": \"$TR\"}, \"failed\": true}" >> $OUT
    done
  done
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
base = next((r['value'] for r in rows if r.get('n_gpus') == 1 and not r.get('failed')), None)
print(f'{"N":>2s} {"layout":8s} {"transport":12s} {"steps/s":>9s} {"s/step":>8s} {"speedup":>8s} {"eff":>6s} {"attn TF/s":>10s} {"hidden":>7s}')
for r in rows:
    s = r['sweep']
    if r.get('failed'):
        print(f'{s["n"]:2d} {s["layout"]:8s} {s["transport_env"]:12s}    FAILED (see gpurun_out/*.err)')
        continue
    sp = r['value'] / base if base else float('nan')
    ov = (r.get('overlap') or {}).get('hidden_frac')
    print(f'{s["n"]:2d} {s["layout"]:8s} {s["transport_env"]:12s} {r["value"]:9.4f} {r["ms_per_step"] / 1e3:8.2f} {sp:8.2f} {sp / s["n"]:6.2f} '
          f'{(r["roofline"]["achieved"] or 0):10.0f} {("%.2f" % ov) if ov is not None else "-":>7s}')
PY
