#!/bin/bash
# Round 6: H pass with the unit entry of DI_i[:, r] read as a column of H^_i + the cardinal Hessian entry as the accumulators' initial value
# (side library exp_build/h2/lib.so: tu_reentry_lgl4_0; exp_build/h2tb/lib.so: tu_twobody_lt_lgl3_1) against the library as built.
out=gpurun_out/r6_h2.txt
: > $out
ASSET_HIP_LIB=exp_build/h2/lib.so python tools/quick_check.py reentry LGL7 0 1 2 3 7 64 257 2049 2560 7000 10000 12345 30011 100003 2>&1 | grep -v amdgpu.ids >> $out
ASSET_HIP_LIB=exp_build/h2tb/lib.so python tools/quick_check.py twobody_lt LGL5 1 1 2 3 7 64 257 2049 7000 10000 12345 30011 100003 2>&1 | grep -v amdgpu.ids >> $out
export QT_REPS=3 QT_WARMUP=100
for rep in 1 2; do
for n in 5000 10000 100000 1000000; do
  [ $n -ge 1000000 ] && export QT_ITERS=20 QT_WARMUP=5 || export QT_ITERS=200 QT_WARMUP=100
  python tools/quick_time.py reentry LGL7 $n 2>&1 | grep -v amdgpu.ids >> $out
  ASSET_HIP_LIB=exp_build/h2/lib.so python tools/quick_time.py reentry LGL7 $n 2>&1 | grep -v amdgpu.ids >> $out
done
for n in 10000 100000; do
  export QT_ITERS=200 QT_WARMUP=100
  python tools/quick_time.py twobody_lt LGL5 $n 1 2>&1 | grep -v amdgpu.ids >> $out
  ASSET_HIP_LIB=exp_build/h2tb/lib.so python tools/quick_time.py twobody_lt LGL5 $n 1 2>&1 | grep -v amdgpu.ids >> $out
done
done
cat $out
