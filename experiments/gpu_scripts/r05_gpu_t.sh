#!/bin/bash
# round-5 pass T: attention with the carried-state steady loop against the previous commit's build: correctness on several shapes (incl. ragged / short /
# the general entry / lse), then alternating timing on one box
NEW=moviigen1.1_amd/lib/mg_selftest; OLD=moviigen1.1_amd/lib_alt/mg_selftest
OUT=gpurun_out/${1:-r05t}_attn_ab.log
: > $OUT
timeout 300 $NEW 2>&1 | grep -E "attn|SELFTEST|FAIL" | head -20 >> $OUT
timeout 300 $NEW attnab 75600 4 1 1 0 10 2>&1 | grep -E "PASS|FAIL" >> $OUT
timeout 300 $NEW attnab 33000 2 2 1 0 10 2>&1 | grep -E "PASS|FAIL" >> $OUT
for r in 1 2 3; do
  echo "== old" >> $OUT; timeout 200 $OLD attnab 131040 8 0 1 10 2>&1 | grep -E "PASS|FAIL" >> $OUT
  echo "== new" >> $OUT; timeout 200 $NEW attnab 131040 8 0 1 10 2>&1 | grep -E "PASS|FAIL" >> $OUT
done
timeout 200 $NEW w64prof 75584 8 0 1 2>&1 | grep "wave 0" >> $OUT
grep -E "==|PASS|FAIL|wave|SELFTEST|attn" $OUT | cut -c1-170
