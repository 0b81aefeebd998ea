#!/bin/bash
# One-shot GPU verification used at the end of a round: parity tests, smoke, bench, rocprof stats, HBM-traffic PMC passes.
TAG=${1:-r01}
R=$PWD; mkdir -p gpurun_out
(python -m pytest tests -q -m gpu 2>&1 | tail -15) > gpurun_out/${TAG}_pytest_gpu.log
python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1
python bench.py --steps 1 --warmup 1 > gpurun_out/${TAG}_bench720p.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o bench720p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench720p_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o w -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
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
rm -rf gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write   # keep only the summaries (size cap)
tail -4 gpurun_out/${TAG}_pytest_gpu.log; tail -2 gpurun_out/${TAG}_smoke.log; tail -1 gpurun_out/${TAG}_bench720p.log; cat gpurun_out/${TAG}_pmc_fetch_summary.txt gpurun_out/${TAG}_pmc_write_summary.txt
