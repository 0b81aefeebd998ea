#!/bin/bash
# PMC passes for one kernel of the selftest (separate rocprofv3 runs; never combined with sys/hip traces)
# usage: tools/pmc_attn.sh <selftest-mode> <outdir-under-gpurun_out>
MODE=${1:-attn1}; OUT=${2:-pmc_attn}
R=$PWD; cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out/$OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
P2="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL"
P3="FETCH_SIZE GRBM_GUI_ACTIVE"
P4="WRITE_SIZE"
P5="TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/$OUT/p$i -o p$i -- $R/moviigen1.1_amd/lib/mg_selftest $MODE > $R/gpurun_out/$OUT/p$i.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/$OUT/p*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:48], r['Counter_Name'])
        agg[k][0] += 1; agg[k][1] += float(r['Counter_Value'])
    for (kn, cn), (n, v) in sorted(agg.items()):
        if 'attn' in kn or 'gemm' in kn:
            print(f'{kn:48s} {cn:28s} dispatches={n:3d} mean={v/n:.4e}')
PY
