#!/bin/bash
# parity of every compiled shape that runs the row-wise dense part, all kinds, one-group / looped / looped-pair mesh sizes
out=gpurun_out/r6_unitc_check.txt
: > $out
for spec in "reentry LGL7 0" "reentry LGL7 1" "reentry LGL5 0" "reentry LGL5 1" "reentry LGL3 0" "reentry LGL3 1" "twobody_lt LGL5 1" "twobody_lt LGL5 0" "twobody_lt LGL3 0" "twobody_lt LGL3 1" "twobody_lt LGL7 0" "twobody_lt LGL7 1" "brachistochrone LGL7 0" "brachistochrone LGL7 1" "brachistochrone LGL5 1" "brachistochrone LGL3 0"; do
  python tools/quick_check.py $spec 1 2 3 7 64 257 2049 7000 10000 12345 30011 100003 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
