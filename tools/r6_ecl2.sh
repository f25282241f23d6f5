#!/bin/bash
out=gpurun_out/r6_ecl2.txt
: > $out
ASSET_HIP_LIB=exp_build/ecl31/lib.so python tools/quick_check.py reentry LGL5 1 1 7 257 2049 10000 30011 100003 2>&1 | grep -v amdgpu.ids >> $out
ASSET_HIP_LIB=exp_build/ecl20/lib.so python tools/quick_check.py reentry LGL3 0 1 7 257 2049 10000 30011 100003 2>&1 | grep -v amdgpu.ids >> $out
export QT_REPS=3
for rep in 1 2; do
for n in 30000 100000 1000000; do
  [ $n -ge 1000000 ] && export QT_ITERS=20 QT_WARMUP=5 || export QT_ITERS=200 QT_WARMUP=100
  for lib in exp_build/base31/lib.so exp_build/ecl31/lib.so; do
    ASSET_HIP_LIB=$lib python tools/quick_time.py reentry LGL5 $n 1 2>&1 | grep -v amdgpu.ids >> $out
  done
  for lib in exp_build/base20/lib.so exp_build/ecl20/lib.so; do
    ASSET_HIP_LIB=$lib python tools/quick_time.py reentry LGL3 $n 0 2>&1 | grep -v amdgpu.ids >> $out
  done
done
done
cat $out
