#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r4_betts; mkdir -p $O
export ASSET_HIP_TUNING=1
for one in 0 1; do
  export ASSET_HIP_UNITS_ONE_LAUNCH=$one
  python3 tools/quick_check.py betts_lowthrust LGL5 0 1 3 7 64 1000 1031 2>&1 | tail -2 | cut -c1-300
  python3 tools/quick_check.py betts_lowthrust LGL7 1 5 100 2>&1 | tail -2 | cut -c1-300
  python3 tools/quick_check.py betts_lowthrust Trapezoidal 0 33 500 2>&1 | tail -2 | cut -c1-300
  for a in "betts_lowthrust LGL5 1000 0" "betts_lowthrust LGL5 2000 0" "betts_lowthrust LGL5 5000 0" "betts_lowthrust LGL5 10000 0" "betts_lowthrust LGL7 1000 0" "betts_lowthrust LGL7 5000 0"; do
    python3 tools/quick_time.py $a 2>&1 | tail -1 | sed "s/^/ONE_LAUNCH=$one /"
  done
done 2>&1 | grep -v "tuning knob" | tee $O/times.log
