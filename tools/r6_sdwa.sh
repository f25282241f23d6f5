#!/bin/bash
# Round 6: byte offsets in the lane records of the row-wise dense part (one v_add_u32_sdwa per slot read instead of unpack + shift-add)
# side libraries: exp_build/isa (before) / isa2 (after): tu_reentry_lgl4_0; isatb / isa2tb: tu_twobody_lt_lgl3_1
out=gpurun_out/r6_sdwa.txt
: > $out
ASSET_HIP_LIB=exp_build/isa2/lib.so python tools/quick_check.py reentry LGL7 0 1 2 3 7 64 257 2049 2560 7000 10000 12345 30011 100003 2>&1 | grep -v amdgpu.ids >> $out
ASSET_HIP_LIB=exp_build/isa2tb/lib.so python tools/quick_check.py twobody_lt LGL5 1 1 2 3 7 64 257 2049 7000 10000 12345 30011 100003 2>&1 | grep -v amdgpu.ids >> $out
export QT_REPS=3
for rep in 1 2 3; do
for n in 10000 100000 1000000; do
  [ $n -ge 1000000 ] && export QT_ITERS=20 QT_WARMUP=5 || export QT_ITERS=200 QT_WARMUP=100
  for lib in exp_build/isa/lib.so exp_build/isa2/lib.so; do
    ASSET_HIP_LIB=$lib python tools/quick_time.py reentry LGL7 $n 2>&1 | grep -v amdgpu.ids >> $out
  done
done
for n in 10000 100000; do
  export QT_ITERS=200 QT_WARMUP=100
  for lib in exp_build/isatb/lib.so exp_build/isa2tb/lib.so; do
    ASSET_HIP_LIB=$lib python tools/quick_time.py twobody_lt LGL5 $n 1 2>&1 | grep -v amdgpu.ids >> $out
  done
done
done
cat $out
