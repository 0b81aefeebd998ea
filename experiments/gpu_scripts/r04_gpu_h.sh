#!/bin/bash
# which library kernels serve the four GEMM shapes (names carry the Tensile configuration), with their durations
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/libg -o g -- python $OLDPWD/tools/bench_lib_gemm.py > $OLDPWD/gpurun_out/r04h_lib_gemm_prof.log 2>&1
cd $OLDPWD
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('/tmp/libg/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        k = (r['Kernel_Name'], r.get('Grid_Size', r.get('Grid_Size_X', '?')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '?')), r.get('LDS_Block_Size', '?'), r.get('VGPR_Count', '?'), r.get('Accum_VGPR_Count', '?'), r.get('SGPR_Count', '?'))
        agg[k][0] += 1; agg[k][1] += d
with open('gpurun_out/r04h_lib_gemm_kernels.txt', 'w') as o:
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        o.write(f'calls={n:4d} total_us={t:12.1f} avg_us={t/n:10.1f} grid={k[1]} wg={k[2]} lds={k[3]} vgpr={k[4]} agpr={k[5]} sgpr={k[6]}\n    {k[0]}\n')
print(open('gpurun_out/r04h_lib_gemm_kernels.txt').read())
PY
