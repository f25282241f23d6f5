#!/bin/bash
# full GPU suite, then timings of the main workloads
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r4_full; mkdir -p $O
python3 -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; grep -a "create .* ms\|passed\|failed\|FAILED" $O/pytest_gpu.log | tail -8
python3 bench.py --steps 200 --warmup 20 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | python3 -c "import sys, json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d.get('host_visible'), d.get('host_visible_assembled'), d['cpu_baseline']['ms_per_eval'])"
for a in "reentry LGL7 10000 0" "reentry LGL7 5000 0" "twobody_lt LGL5 10000 1" "reentry Trapezoidal 10000 0" "betts_lowthrust LGL5 1000 0" "betts_lowthrust LGL7 5000 0" "reentry LGL7 100000 0"; do
  python3 tools/quick_time.py $a 2>&1 | tail -1
done | tee $O/times.log
