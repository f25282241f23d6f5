#!/bin/bash
# parity sweep of the shipped library: every compiled (ODE, transcription, control mode), every kind, ragged mesh sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for ode in brachistochrone reentry twobody_lt betts_lowthrust synthetic32; do
  for mode in Trapezoidal LGL3 LGL5 LGL7; do
    for blk in 0 1; do
      sizes="1 2 3 7 64 257 2049 5003"
      [ $ode = synthetic32 ] && sizes="1 2 3 7 64 257 1031"
      [ $ode = betts_lowthrust ] && sizes="1 2 3 7 64 257 1031 2049 4500"
      timeout 600 python3 tools/quick_check.py $ode $mode $blk $sizes 2>&1 | tail -2
    done
  done
done
