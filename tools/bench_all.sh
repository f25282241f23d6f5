R=${GRAFT_REPO_ROOT:-/root/repo}
rm -f $R/gpurun_out/${RND:-r4}_bench_all.jsonl
# (a throw-away run first: the first process on a fresh box -- or the first after a profile collection -- has run 10 % slow: 32.96 against 29.1 us)
python3 $R/bench.py --workload reentry_lgl7_5k --no-cpu-baseline > /dev/null 2>&1
for wl in ${WLS:-reentry_lgl7_10k reentry_lgl7_5k reentry_lgl7_100k reentry_lgl7_1m betts_lgl5_1k twobody_lgl5_blocked_10k twobody_lgl5_blocked_100k multispacecraft_8x1250 synthetic32_lgl7_12500 synthetic32_lgl7_100k twobody_lgl7_10k betts_lgl7_5k brachistochrone_lgl3_40 reentry_trap_10k twobody_trap_blocked_10k}; do
  python3 $R/bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 >> $R/gpurun_out/${RND:-r4}_bench_all.jsonl
done
python3 $R/bench.py > $R/gpurun_out/${RND:-r4}_bench_default.log 2>&1
tail -1 $R/gpurun_out/${RND:-r4}_bench_default.log >> $R/gpurun_out/${RND:-r4}_bench_all.jsonl
wc -l $R/gpurun_out/${RND:-r4}_bench_all.jsonl
