#!/bin/bash
# round 5: HBM write bytes per launch of a side library's north-star kernel (rocprofv3 --pmc, its own run, no trace domain)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export ASSET_HIP_LIB=$R/exp_build/$v/lib.so QT_REPS=1 QT_ITERS=10 QT_WARMUP=2
  rm -rf /tmp/pmc_$v
  timeout 200 rocprofv3 --pmc WRITE_SIZE FETCH_SIZE --output-format csv -d /tmp/pmc_$v -o q -- python3 $R/tools/quick_time.py reentry LGL7 10000 > /tmp/pmc_$v.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_$v/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'resident' in r['Kernel_Name'] and 'Li2ELb0ELb0' in r['Kernel_Name'] or ('lgl_resident_kernel' in r['Kernel_Name'] and ', 2, false, false' in r['Kernel_Name']):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('$v', {k: (sum(x)/len(x), len(x)) for k,x in acc.items()})
PY
done
