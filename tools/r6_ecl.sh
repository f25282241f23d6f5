#!/bin/bash
# Round 6: the C passes of a group under the cardinal second-derivative phase (EARLYC) in the LOOPED pair kernel too
out=gpurun_out/r6_ecl.txt
: > $out
ASSET_HIP_LIB=exp_build/ecl/lib.so python tools/quick_check.py reentry LGL7 0 1 7 257 2049 10000 10241 12345 30011 60001 100003 2>&1 | grep -v amdgpu.ids >> $out
export QT_REPS=3
for rep in 1 2 3; do
for n in 15000 30000 100000 1000000; do
  [ $n -ge 1000000 ] && export QT_ITERS=20 QT_WARMUP=5 || export QT_ITERS=200 QT_WARMUP=100
  for lib in exp_build/h2/lib.so exp_build/ecl/lib.so; do
    ASSET_HIP_LIB=$lib python tools/quick_time.py reentry LGL7 $n 2>&1 | grep -v amdgpu.ids >> $out
  done
done
done
cat $out
