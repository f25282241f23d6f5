#!/bin/bash
# LDS bank-conflict counters of one library variant (rocprofv3 --pmc, its own pass): ASSET_HIP_LIB=... bash tools/r4_pmc_lds.sh tag [workload]
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; wl=${2:-reentry_lgl7_10k}
O=$R/gpurun_out/pmc_$tag; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $O -o bench -- python3 $R/bench.py --no-cpu-baseline --workload $wl --steps 20 --warmup 2 > $O/log.txt 2>&1
python3 - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$O/bench_counter_collection.csv")):
    if "resident_kernel" in r["Kernel_Name"]:
        lvl=r["Kernel_Name"].split(",")[3].strip()
        acc[lvl][r["Counter_Name"]].append(float(r["Counter_Value"]))
for lvl,d in acc.items():
    print("$tag level", lvl, {k: int(sum(v)/len(v)) for k,v in d.items()})
PY
