#!/bin/bash
# is the chip at its power cap in both hot kernels?  rocm-smi power / clock samples every 0.25 s while GEMM variant 8, variant 11 and the
# attention kernel run alone, back to back (each ~9 s of kernel time after the operand setup; random operands: power depends on the data)
TAG=${1:-r04w}
S=moviigen1.1_amd/lib/mg_selftest
OUT=gpurun_out/${TAG}_power.log
sample() {  # $1 = label; runs until the file /tmp/stop_sampling exists
  while [ ! -e /tmp/stop_sampling ]; do
    echo "$1 $(date +%s.%N | cut -c1-14) $(timeout 5 rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E 'Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level|Temperature \(Sensor junction\)|Sensor edge' | sed 's/.*: //' | tr '\n' '|')"
    sleep 0.25
  done
}
run() {  # label, command...
  rm -f /tmp/stop_sampling; lbl=$1; shift
  sample $lbl >> $OUT & SP=$!
  "$@" > /tmp/run_$lbl.log 2>&1
  touch /tmp/stop_sampling; wait $SP
  grep -E "powerloop" /tmp/run_$lbl.log | tail -2 >> $OUT
}
timeout 10 rocm-smi --showpower --showclocks --showmaxpower --showtemp > gpurun_out/${TAG}_smi_idle.txt 2>&1
: > $OUT
run gemm8 timeout 100 $S powerloop gemm 8 9
run gemm11 timeout 100 $S powerloop gemm 11 9
run attn timeout 100 $S powerloop attn 0 9
run gemm8b timeout 100 $S powerloop gemm 8 9
run gemm11b timeout 100 $S powerloop gemm 11 9
tail -5 gpurun_out/${TAG}_smi_idle.txt; grep -c . $OUT; grep powerloop $OUT; for l in gemm8 gemm11 attn; do grep "^$l " $OUT | tail -6 | head -3; done
