#!/bin/bash
# first GPU call of round 4: Trapezoidal on the resident kernel -- parity, then timings next to the old path
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r4_first; mkdir -p $O
python3 -m pytest tests/test_gpu_parity.py -x -q -k "Trapezoidal" > $O/pytest_trap.log 2>&1; tail -3 $O/pytest_trap.log
python3 -m pytest tests/test_gpu_distributed.py -x -q -k "rccl" > $O/pytest_dist.log 2>&1; tail -3 $O/pytest_dist.log
for a in "reentry Trapezoidal 10000 0" "twobody_lt Trapezoidal 10000 1" "reentry LGL7 10000 0" "reentry LGL7 5000 0" "twobody_lt LGL5 10000 1" "betts_lowthrust LGL5 1000 0" "twobody_lt LGL7 10000 0"; do
  python3 tools/quick_time.py $a 2>&1 | tail -1
done | tee $O/times.log
for a in "reentry Trapezoidal 10000 0" "twobody_lt Trapezoidal 10000 1"; do
  ASSET_HIP_TUNING=1 ASSET_HIP_NO_RESIDENT=1 python3 tools/quick_time.py $a 2>&1 | tail -1 | sed 's/^/NO_RESIDENT /'
  python3 tools/quick_time.py $a 2 2>&1 | tail -1
  python3 tools/quick_time.py $a 3 2>&1 | tail -1
done | tee -a $O/times.log
