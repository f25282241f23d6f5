#!/bin/bash
# round 6: GPU tests, the bench lines of every workload, the in-process shard leg, the self-launched two-rank run (gloo, one GPU), then
# rocprofv3 kernel stats + PMC passes of the workloads named; everything into gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python3 -m pytest tests -m gpu -q > gpurun_out/r6_gputest3.log 2>&1; tail -8 gpurun_out/r6_gputest3.log
RND=r6 bash tools/bench_all.sh > gpurun_out/r6_bench_all.log 2>&1
python3 bench.py --no-cpu-baseline --inprocess-shards 4 --steps 200 2>/dev/null | tail -1 > gpurun_out/r6_bench_inprocess.json
ASSET_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6_bench_selflaunch.json 2> gpurun_out/r6_bench_selflaunch.err
bash tools/collect_all_profiles.sh r6 ${PROF_WLS:-reentry_lgl7_10k reentry_lgl7_5k reentry_lgl7_100k reentry_lgl7_1m twobody_lgl5_blocked_10k twobody_lgl5_blocked_100k multispacecraft_8x1250 betts_lgl5_1k betts_lgl7_5k synthetic32_lgl7_100k twobody_lgl7_10k reentry_trap_10k} > gpurun_out/r6_profiles.log 2>&1
tail -3 gpurun_out/r6_profiles.log
python3 - <<'PY'
import json
for l in open("gpurun_out/r6_bench_all.jsonl"):
    try: d = json.loads(l)
    except Exception: continue
    r = d["roofline"]
    print(d["config"]["name"], "ms/step %.4f" % d["ms_per_step"], "frac %.3f" % r["frac"], "fp64 %.3f" % ((r.get("fp64") or {}).get("frac", float("nan"))))
PY
# compute-only time of the north-star launch (no block stores: a null KKT pointer), both forms
python3 tools/attic/dbg_nostore.py reentry LGL7 10000 > gpurun_out/r6_nostore.txt 2>&1
ASSET_HIP_TUNING=1 ASSET_HIP_NO_ALT_FORM=1 python3 tools/attic/dbg_nostore.py reentry LGL7 10000 >> gpurun_out/r6_nostore.txt 2>&1
cat gpurun_out/r6_nostore.txt
