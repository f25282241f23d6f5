#!/bin/bash
# round 6: with the J | H block layout, the row-wise dense part (rd) against the tile form on the shapes the tile form kept in round 5
# (Reentry without segment parameters), by mesh size.  Side libraries exp_build/<tu>_{rd,tile}/lib.so (tools/build_one.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_forms.txt
mkdir -p $R/gpurun_out; : > $OUT
tm() { # lib ode mode nseg
  ASSET_HIP_LIB=$R/exp_build/$1/lib.so QT_REPS=${QT_REPS:-5} QT_ITERS=${QT_ITERS:-200} python3 $R/tools/quick_time.py $2 $3 $4 0 2>&1 | tail -1 | sed "s|$R/exp_build/||" >> $OUT
}
for n in 1000 2500 5000 7500 10000 15000 30000; do
  for m in "lgl2 LGL3" "lgl3 LGL5" "lgl4 LGL7"; do set -- $m
    for f in tile rd; do tm reentry_$1_0_$f reentry $2 $n; done
  done
done
export QT_REPS=3 QT_ITERS=50
for n in 100000 1000000; do
  for m in "lgl3 LGL5" "lgl4 LGL7"; do set -- $m
    for f in tile rd; do tm reentry_$1_0_$f reentry $2 $n; done
  done
done
cat $OUT
