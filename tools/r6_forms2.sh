#!/bin/bash
# Round 6, after UNITC: where the launcher's thresholds between the two forms of the dense part belong now (registry.h: alt_min, lpair_min).
# rows = the row-wise kernels forced (ASSET_HIP_ALT_MIN=1 / ASSET_HIP_LPAIR_MIN=1), tiles = the tile kernels forced (99).
out=gpurun_out/r6_forms2.txt
: > $out
export ASSET_HIP_TUNING=1 QT_REPS=3 QT_WARMUP=100
for mode in LGL7 LGL5 LGL3; do
  for n in 1000 2500 4000 5000 6000 7500 10000; do
    for f in 1 99; do
      echo -n "alt_min=$f " >> $out
      ASSET_HIP_ALT_MIN=$f python tools/quick_time.py reentry $mode $n 2>&1 | grep -v "amdgpu.ids\|asset_hip:" >> $out
    done
  done
  for n in 15000 20000 30000 40000 60000; do
    for f in 1 99; do
      echo -n "lpair_min=$f " >> $out
      ASSET_HIP_LPAIR_MIN=$f python tools/quick_time.py reentry $mode $n 2>&1 | grep -v "amdgpu.ids\|asset_hip:" >> $out
    done
  done
done
cat $out
