#!/bin/bash
# round-5 pass S: the attention kernel of the working tree against the previous commit's build (moviigen1.1_amd/lib_alt/: same selftest, its own .so next to it
# by rpath $ORIGIN), alternating processes on one box
NEW=moviigen1.1_amd/lib/mg_selftest; OLD=moviigen1.1_amd/lib_alt/mg_selftest
OUT=gpurun_out/${1:-r05s}_attn_ab.log
: > $OUT
for r in 1 2 3; do
  echo "== old" >> $OUT; timeout 200 $OLD attnab 131040 8 ${DATA:-0} 1 10 2>&1 | grep -E "PASS|FAIL" >> $OUT
  echo "== new" >> $OUT; timeout 200 $NEW attnab 131040 8 ${DATA:-0} 1 10 2>&1 | grep -E "PASS|FAIL" >> $OUT
done
timeout 200 $NEW w64prof 75584 8 0 1 2>&1 | grep "wave 0" >> $OUT
grep -E "==|PASS|FAIL|wave" $OUT | cut -c1-150
