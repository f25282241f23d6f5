#!/bin/bash
# experiments: exp_build/v_*/lib.so variants of one translation unit, REPS interleaved rounds
#   ODE=reentry MODE=LGL7 BLK=0 SIZES="10000 5000" REPS=2 bash tools/r4_exp.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r4_exp; mkdir -p $O
ODE=${ODE:-reentry}; MODE=${MODE:-LGL7}; BLK=${BLK:-0}
: > $O/times.log
if [ -z "$NOCHECK" ]; then
  for v in exp_build/v_*/lib.so; do ASSET_HIP_LIB=$R/$v python3 tools/quick_check.py $ODE $MODE $BLK 2>&1 | tail -3; done | tee -a $O/times.log
fi
for rep in $(seq 1 ${REPS:-2}); do
  for v in exp_build/v_*/lib.so; do
    for n in ${SIZES:-10000 5000}; do
      ASSET_HIP_LIB=$R/$v python3 tools/quick_time.py $ODE $MODE $n $BLK ${KIND:-4} 2>&1 | tail -1 | sed "s|$R/exp_build/||"
    done
  done
done | tee -a $O/times.log
