#!/bin/bash
# Round 6, after UNITC: looped pair kernel (1) against looped single-wave kernel (99) on large meshes, shapes whose default form is the row-wise one
out=gpurun_out/r6_forms4.txt
: > $out
export ASSET_HIP_TUNING=1 QT_REPS=3 QT_WARMUP=30 QT_ITERS=50
for spec in "twobody_lt LGL5 B1" "twobody_lt LGL5 B0" "twobody_lt LGL3 B0" "twobody_lt LGL3 B1" "reentry LGL5 B1" "reentry LGL3 B1" "brachistochrone LGL7 B0" "brachistochrone LGL5 B1"; do
  set -- $spec
  for n in 80000 100000 200000 1000000; do
    for f in 1 99; do
      echo -n "lpair_min=$f " >> $out
      ASSET_HIP_LPAIR_MIN=$f python tools/quick_time.py $1 $2 $n ${3#B} 2>&1 | grep -v "amdgpu.ids\|asset_hip:" >> $out
    done
  done
done
cat $out
