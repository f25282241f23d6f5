#!/bin/bash
# Strong scaling on ONE node with N = 1, 2, 4, 8 GPUs (what the driver's SCALE run does for the default workload), for the workloads
# the >= 6x claim is made on (bench.py: config.scaling_claim): one JSON line per (workload, N) into gpurun_out/<round>_scale.jsonl.
#   RND=r5 bash tools/bench_scale.sh [workloads ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/${RND:-r5}_scale.jsonl
rm -f $out
wls=${@:-reentry_lgl7_10k reentry_lgl7_1m synthetic32_lgl7_100k multispacecraft_8x1250}
ngpu=$(python3 -c "import torch; print(torch.cuda.device_count())")
for wl in $wls; do
  for n in 1 2 4 8; do
    [ $n -gt $ngpu ] && continue
    if [ $n -eq 1 ]; then
      python3 $R/bench.py --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 >> $out
    else
      python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
        $R/bench.py --gpus $n --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 >> $out
    fi
  done
done
python3 - <<PY
import json
for l in open("$out"):
    try: d = json.loads(l)
    except Exception: continue
    print(d["config"]["name"], d["n_gpus"], "value %.3g" % d["value"], "without exchange %.3g" % d.get("value_without_exchange", d["value"]),
          "host-visible assembled %.3g" % (d.get("host_visible_assembled") or {}).get("segments_per_s", float("nan")))
PY
