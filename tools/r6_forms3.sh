#!/bin/bash
# Round 6, after UNITC: the looped pair kernel against the looped single-wave kernel on shapes whose default form is the row-wise one
out=gpurun_out/r6_forms3.txt
: > $out
export ASSET_HIP_TUNING=1 QT_REPS=3 QT_WARMUP=100
for spec in "twobody_lt LGL5 B1" "twobody_lt LGL3 B0" "reentry LGL7 B1" "brachistochrone LGL7 B0"; do
  set -- $spec
  for n in 12000 15000 20000 30000 40000 60000; do
    for f in 1 2 3 99; do
      echo -n "lpair_min=$f " >> $out
      ASSET_HIP_LPAIR_MIN=$f python tools/quick_time.py $1 $2 $n ${3#B} 2>&1 | grep -v "amdgpu.ids\|asset_hip:" >> $out
    done
  done
done
cat $out
