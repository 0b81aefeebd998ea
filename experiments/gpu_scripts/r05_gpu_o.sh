#!/bin/bash
# round-5 pass O: pipeline depth vs attention rounds on one GPU; the rank emulation (sanity run on 2 layers, then the real thing)
TAG=${1:-r05o}
timeout 300 python tools/sp_groups_probe.py > gpurun_out/${TAG}_sp_groups.txt 2>&1
timeout 300 python tools/emulate_rank.py --workload 720p --layers 2 --ranks 2 8 --fsdp-at 8 > gpurun_out/${TAG}_emulate_sanity.log 2>&1
tail -3 gpurun_out/${TAG}_emulate_sanity.log | cut -c1-700
if grep -q Traceback gpurun_out/${TAG}_emulate_sanity.log; then tail -20 gpurun_out/${TAG}_emulate_sanity.log; cat gpurun_out/${TAG}_sp_groups.txt; exit 1; fi
timeout 1200 python tools/emulate_rank.py --workload 1080p --ranks 2 4 8 --fsdp-at 8 > gpurun_out/${TAG}_emulate_1080p.jsonl 2> gpurun_out/${TAG}_emulate_1080p.err
timeout 900 python tools/emulate_rank.py --workload 1056p --ranks 8 --fsdp-at 8 > gpurun_out/${TAG}_emulate_1056p.jsonl 2> gpurun_out/${TAG}_emulate_1056p.err
cat gpurun_out/${TAG}_sp_groups.txt
python3 - <<PY
import json
for f in ('gpurun_out/${TAG}_emulate_1080p.jsonl','gpurun_out/${TAG}_emulate_1056p.jsonl'):
    for ln in open(f):
        try: d=json.loads(ln)
        except Exception: continue
        if 'ranks' in d:
            print(d['workload'], d['ranks'], d['layout'], 'compute %.2f s'%d['compute_s_per_step'], 'groups', d['pipeline_groups'], 'exch ms', {k: round(v,1) for k,v in d['exchange_ms_per_step'].items()}, 'gather ms %.0f'%d['block_gather_ms_per_step_if_exposed'], 'eff', {k: round(v,3) for k,v in d['implied_strong_scaling_efficiency'].items()})
        else: print(d)
PY
tail -3 gpurun_out/${TAG}_emulate_1080p.err gpurun_out/${TAG}_emulate_1056p.err | cut -c1-300
