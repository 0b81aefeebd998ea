#!/bin/bash
# PMC passes over the WanVAE decode (tools/bench_vae.py, full 1920x832 spatial size, few frames) — separate rocprofv3 runs,
# counters only with --kernel-trace (never combined with sys/hip traces).   usage: tools/pmc_vae.sh <outdir-under-gpurun_out> [frames]
OUT=${1:-pmc_vae}; FR=${2:-9}
R=$PWD; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/$OUT/p$i -o p$i -- python $R/tools/bench_vae.py --frames $FR > $R/gpurun_out/$OUT/p$i.log 2>&1
done
cd $R
python3 - <<PY > gpurun_out/$OUT/summary.txt
import csv, glob, collections
print('# rocprofv3 --pmc passes of tools/bench_vae.py --frames $FR (1920x832), totals over ALL dispatches of each kernel')
print('# units: SQ_WAVE_CYCLES/SQ_WAIT_*/SQ_ACTIVE_INST_* quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES cycles; FETCH/WRITE_SIZE KiB (FETCH_SIZE x2 on gfx950)')
tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob('gpurun_out/$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
        t = tot[kn][r['Counter_Name']]
        t[0] += 1; t[1] += float(r['Counter_Value'])
for f in sorted(glob.glob('gpurun_out/$OUT/p1/**/*kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]
        dur[kn][0] += 1; dur[kn][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for kn in sorted(tot, key=lambda k: -dur[k][1]):
    if dur[kn][1] < 1000: continue
    print(f'== {kn}: dispatches={dur[kn][0]} total_us={dur[kn][1]:.0f}')
    c = {k: v[1] for k, v in tot[kn].items()}
    for k in sorted(c): print(f'   {k:28s} {c[k]:.4e}')
    if 'GRBM_GUI_ACTIVE' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c and c['GRBM_GUI_ACTIVE'] > 0:
        cyc = c['GRBM_GUI_ACTIVE'] / 8          # summed over the 8 XCDs
        print(f'   derived: shader cycles {cyc:.4e} -> effective clock {cyc / dur[kn][1] / 1e3:.2f} GHz; MFMA pipe busy {c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) * 100:.1f} % of 1024 SIMDs')
    if 'TCC_HIT_sum' in c: print(f'   derived: L2 hit rate {c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) * 100:.1f} %')
PY
rm -rf gpurun_out/$OUT/p?        # raw csv: size cap
cat gpurun_out/$OUT/summary.txt
