#!/bin/bash
# Round 6: the looped form in FOUR-wave workgroups (sixteen segments through one ODE chain: ResDims / lgl_resident_body QUAD, K_RESLQ) against the looped pair kernel.
# side library exp_build/quad/lib.so (tu_reentry_lgl4_0); ASSET_HIP_LQUAD_MIN=1 takes the quad form from one group per workgroup on.
out=gpurun_out/r6_quad.txt
: > $out
export ASSET_HIP_TUNING=1 ASSET_HIP_LIB=exp_build/quad/lib.so
ASSET_HIP_LQUAD_MIN=1 python tools/quick_check.py reentry LGL7 0 8192 8200 10241 12345 16384 16400 30011 60001 100003 2>&1 | grep -v "amdgpu.ids\|asset_hip:" >> $out
export QT_REPS=3
for rep in 1 2; do
for n in 15000 30000 100000 1000000; do
  [ $n -ge 1000000 ] && export QT_ITERS=20 QT_WARMUP=5 || export QT_ITERS=200 QT_WARMUP=100
  for q in 0 1; do
    echo -n "lquad_min=$q " >> $out
    ASSET_HIP_LQUAD_MIN=$q python tools/quick_time.py reentry LGL7 $n 2>&1 | grep -v "amdgpu.ids\|asset_hip:" >> $out
  done
done
done
cat $out
