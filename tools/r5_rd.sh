#!/bin/bash
# round 5: the row-wise dense part of the resident kernel (defect_rowdpp.h) against the tile form, side libraries of one translation unit
set -x
for v in "$@"; do
  ASSET_HIP_LIB=exp_build/$v/lib.so python tools/quick_check.py reentry LGL7 0 1 2 3 7 64 257 2049 10000 2>&1 | tail -3
  ASSET_HIP_LIB=exp_build/$v/lib.so python tools/quick_time.py reentry LGL7 10000 2>&1 | tail -1
  ASSET_HIP_LIB=exp_build/$v/lib.so python tools/quick_time.py reentry LGL7 5000 2>&1 | tail -1
done
