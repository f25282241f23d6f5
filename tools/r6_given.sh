#!/bin/bash
# Round 6: the row-wise dense part behind the unit kernels of a heavy right-hand side (ResDims::RD_GIVEN) against the tile form (the library as
# built before the change).  Side libraries: exp_build/bt (tu_betts_lowthrust_lgl4_0), bt5 (lgl3_0), bt31 (lgl2_1).
out=gpurun_out/r6_given.txt
: > $out
ASSET_HIP_LIB=exp_build/bt/lib.so python tools/quick_check.py betts_lowthrust LGL7 0 1 2 3 7 64 257 1031 2049 4500 5000 9001 2>&1 | grep -v amdgpu.ids >> $out
ASSET_HIP_LIB=exp_build/bt5/lib.so python tools/quick_check.py betts_lowthrust LGL5 0 1 2 3 7 64 257 1000 1031 2049 4500 2>&1 | grep -v amdgpu.ids >> $out
ASSET_HIP_LIB=exp_build/bt31/lib.so python tools/quick_check.py betts_lowthrust LGL3 1 1 2 3 7 64 257 1031 2049 4500 2>&1 | grep -v amdgpu.ids >> $out
export QT_REPS=3 QT_WARMUP=100
for rep in 1 2; do
for n in 1000 5000 20000; do
  python tools/quick_time.py betts_lowthrust LGL7 $n 2>&1 | grep -v amdgpu.ids >> $out
  ASSET_HIP_LIB=exp_build/bt/lib.so python tools/quick_time.py betts_lowthrust LGL7 $n 2>&1 | grep -v amdgpu.ids >> $out
  python tools/quick_time.py betts_lowthrust LGL5 $n 2>&1 | grep -v amdgpu.ids >> $out
  ASSET_HIP_LIB=exp_build/bt5/lib.so python tools/quick_time.py betts_lowthrust LGL5 $n 2>&1 | grep -v amdgpu.ids >> $out
done
done
cat $out
