#!/bin/bash
# One-shot GPU verification used at the end of a round: parity tests, smoke, bench (the metric's 1920x832x81f
# workload), rocprof stats of the same command, HBM-traffic PMC passes of the dominant kernel.
# usage: tools/round_end_gpu.sh <tag> [workload]      (PMC passes use --layers 4: per-launch traffic is layer-independent)
TAG=${1:-r03}
WL=${2:-1080p}
R=$PWD; mkdir -p gpurun_out
if [ -z "$SKIP_PYTEST" ]; then
(python -m pytest tests -q -m gpu 2>&1 | tail -25) > gpurun_out/${TAG}_pytest_gpu.log
fi
python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1
python bench.py --workload $WL --steps ${BENCH_STEPS:-2} --warmup 1 > gpurun_out/${TAG}_bench${WL}.json.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-calibration --no-shared-prefix-leg > $R/gpurun_out/${TAG}_bench${WL}_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o f -- python $R/bench.py --workload $WL --layers 4 --steps 1 --warmup 0 --no-cpu-baseline --no-video-tail --no-pmc > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o w -- python $R/bench.py --workload $WL --layers 4 --steps 1 --warmup 0 --no-cpu-baseline --no-video-tail --no-pmc > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
for tag, name in (('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob('gpurun_out/${TAG}_pmc_%s/**/*counter_collection.csv' % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name'].split('(')[0][-40:]
            agg[kn][0] += 1; agg[kn][1] += float(r['Counter_Value'])
    with open('gpurun_out/${TAG}_pmc_%s_summary.txt' % tag, 'w') as o:
        for kn, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
            o.write(f'{name} {kn:42s} dispatches={n:5d} mean_KiB={v/n:.4e} total_KiB={v:.4e}\n')
PY
python3 tools/rocprof_summary.py $(ls gpurun_out/${TAG}_prof/*/*_results.db gpurun_out/${TAG}_prof/*_results.db 2>/dev/null | head -1) gpurun_out/${TAG}_bench${WL}_kernel_stats.txt > /dev/null 2>&1
python3 - <<PY
# HBM-side traffic of the dominant kernel (self-attention = the long dispatches of attn_hd128_m16_kernel), per launch
import csv, glob, json
L = {'720p': 75600, '1080p': 131040, '1056p': 166320}.get('${WL}', 0)
def mean_kib(tag):
    v = []
    for f in glob.glob('gpurun_out/${TAG}_pmc_%s/**/*counter_collection.csv' % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'attn_hd128_m16_kernel' in r['Kernel_Name']:
                v.append(float(r['Counter_Value']))
    if v and max(v) > 4 * min(v):      # the same kernel also serves the 512-key cross-attention: keep the long launches
        cut = (max(v) + min(v)) / 2
        v = [x for x in v if x > cut]
    return (sum(v) / len(v), len(v)) if v else (None, 0)
fe, nf = mean_kib('fetch'); wr, nw = mean_kib('write')
if fe is not None and wr is not None:
    out = {'kernel': 'attn_hd128_m16_kernel, self-attention Lq=Lk=%d, 40 heads (bench.py ${WL} workload)' % L,
           'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, two separate passes of bench.py --workload ${WL} --layers 4 --steps 1 --warmup 0 (tools/round_end_gpu.sh; the launch shape does not depend on the layer count); mean over the dispatches of the kernel',
           'dispatches': [nf, nw], 'fetch_size_kib_per_launch_raw': fe, 'write_size_kib_per_launch_raw': wr,
           'gfx950_correction': 'FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read stream (MI355X_MICROARCH.md, HBM section): doubled; WRITE_SIZE as is',
           'traffic_bytes_per_launch': 2 * fe * 1024 + wr * 1024,
           'algorithmic_bytes_per_launch': 4 * L * 5120 * 2}
    json.dump(out, open('gpurun_out/${TAG}_pmc_traffic_${WL}.json', 'w'), indent=1)
    print(json.dumps(out))
PY
rm -rf gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write   # keep only the summaries (size cap)
find gpurun_out/${TAG}_prof -name '*.db' -size +20M -delete 2>/dev/null
tail -8 gpurun_out/${TAG}_pytest_gpu.log 2>/dev/null; tail -2 gpurun_out/${TAG}_smoke.log; tail -1 gpurun_out/${TAG}_bench${WL}.json.log; cat gpurun_out/${TAG}_pmc_fetch_summary.txt gpurun_out/${TAG}_pmc_write_summary.txt
