#!/bin/bash
# counters of the vendor library's GEMM kernel and of mg_gemm_bf16 on the same operands (tools/bench_lib_gemm.py), three
# separate rocprofv3 --pmc passes (never combined with other traces): clocks, matrix-pipe occupancy, L2 hit rate, fabric traffic
R=$PWD; OUT=$R/gpurun_out/${1:-r04k}_pmc_lib_gemm.txt; : > $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/plg[0-9]*
i=0
# PMC_SETS="a;b;c" overrides the counter passes (one rocprofv3 run per set)
IFS=';' read -ra SETS <<< "${PMC_SETS:-FETCH_SIZE GRBM_GUI_ACTIVE;SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY;TCC_HIT_sum TCC_MISS_sum;WRITE_SIZE}"
for P in "${SETS[@]}"; do
  i=$((i+1)); rm -rf /tmp/plg$i
  MG_GEMM_VARIANT=${MG_GEMM_VARIANT:-0} rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/plg$i -o p -- python $R/tools/bench_lib_gemm.py 131040 > /tmp/plg$i.log 2>&1
done
cd $R
python3 - <<PY >> $OUT
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('/tmp/plg*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        kn = 'vendor ' + r['Kernel_Name'][:60] if r['Kernel_Name'].startswith('Custom_') else ('mg ' + r['Kernel_Name'][:40] if 'gemm_bf16' in r['Kernel_Name'] else None)
        if kn is None: continue
        k = (kn, r['Grid_Size'], r['Counter_Name'])
        agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
for f in glob.glob('/tmp/plg1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        kn = 'vendor' if r['Kernel_Name'].startswith('Custom_') else ('mg' if 'gemm_bf16' in r['Kernel_Name'] else None)
        if kn is None: continue
        dur[(kn, r.get('Grid_Size', r.get('Grid_Size_X', '?')))][0] += 1; dur[(kn, r.get('Grid_Size', r.get('Grid_Size_X', '?')))][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for (kn, gs, cn), (n, v) in sorted(agg.items()):
    print(f'{kn:68s} grid={gs:>8s} {cn:26s} dispatches={n:3d} mean={v/n:.4e}')
for (kn, gs), (n, v) in sorted(dur.items()):
    print(f'{kn:8s} grid={gs:>8s} dispatches={n:3d} mean_us={v/n:.1f}')
PY
cat $OUT
