#!/bin/bash
# round 6: the launcher's choice between the two forms of the dense part (ResDims::RD_ALT) against the tile form alone
# (ASSET_HIP_TUNING=1 ASSET_HIP_NO_ALT_FORM=1), by mesh size; parity of the chosen form first.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_altform.txt
mkdir -p $R/gpurun_out; : > $OUT
python3 $R/tools/quick_check.py reentry LGL7 0 1 5 700 6143 6144 7500 10000 10240 10241 30011 60003 >> $OUT 2>&1
python3 $R/tools/quick_check.py reentry LGL5 0 3 7500 10000 100003 >> $OUT 2>&1
python3 $R/tools/quick_check.py reentry LGL3 0 2 6000 7500 33000 140000 >> $OUT 2>&1
tm() { QT_REPS=4 python3 $R/tools/quick_time.py reentry $1 $2 0 2>&1 | tail -1 | sed "s|^default|$3|" >> $OUT; }
for m in LGL3 LGL5 LGL7; do
  for n in 5000 7500 10000 15000 30000 100000 1000000; do
    [ $n -ge 100000 ] && export QT_ITERS=30 || export QT_ITERS=200
    tm $m $n chosen
    ASSET_HIP_TUNING=1 ASSET_HIP_NO_ALT_FORM=1 tm $m $n tile-only 2>/dev/null
  done
done
cat $OUT
