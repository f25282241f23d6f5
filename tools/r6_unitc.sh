#!/bin/bash
# Round 6: the C pass for unit multiplier vectors + the gradient from the H passes (defect_rowdpp.h: UNITC) against the general C pass.
# Side libraries of one translation unit each, built here (no GPU needed) by:
#   python tools/build_one.py tu_reentry_lgl4_0 exp_build/libunitc.so ; ... -DASSET_RD_UNITC=0 -> exp_build/base/libbase.so
out=gpurun_out/r6_unitc.txt
mkdir -p gpurun_out
: > $out
for lib in exp_build/libunitc.so; do
  ASSET_HIP_LIB=$lib python tools/quick_check.py reentry LGL7 0 1 2 3 7 64 257 2049 6144 7000 10000 12345 30011 100003 >> $out 2>&1
done
for rep in 1 2; do
for n in 7500 10000 100000 1000000; do
  for lib in exp_build/base/libbase.so exp_build/libunitc.so; do
    QT_ITERS=$([ $n -ge 1000000 ] && echo 20 || echo 200) ASSET_HIP_LIB=$lib python tools/quick_time.py reentry LGL7 $n >> $out 2>&1
  done
done
done
cat $out
