#!/bin/bash
# round 5: row-wise dense part (rd) against the tile form (base) per shape, side libraries exp_build/v_<tu>_{rd,base}
t() { # tu ode mode blocked
  for v in rd base; do
    L=exp_build/v_$1_$v/lib.so
    [ -f $L ] || continue
    ASSET_HIP_LIB=$L python tools/quick_check.py $2 $3 $4 1 3 7 257 10000 2>&1 | tail -1 | sed 's/sizes.*worst/worst/'
    ASSET_HIP_LIB=$L QT_REPS=4 python tools/quick_time.py $2 $3 10000 $4 | tail -1
  done
}
t reentry_lgl2_0 reentry LGL3 0
t reentry_lgl3_0 reentry LGL5 0
t reentry_lgl3_1 reentry LGL5 1
t reentry_lgl4_1 reentry LGL7 1
t twobody_lt_lgl2_0 twobody_lt LGL3 0
t twobody_lt_lgl2_1 twobody_lt LGL3 1
t twobody_lt_lgl3_0 twobody_lt LGL5 0
t brachistochrone_lgl4_0 brachistochrone LGL7 0
t brachistochrone_lgl3_1 brachistochrone LGL5 1
