#!/bin/bash
# round 6 experiment: every block column on 128-byte lines of its own (al_*, -DASSET_EXP_ALIGNED: wrong layout table, right byte counts + padding)
# against the packed J | H layout (na_*), row-wise dense part; times and WRITE_SIZE
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_aligned.txt
mkdir -p $R/gpurun_out; : > $OUT
tm() { ASSET_HIP_LIB=$R/exp_build/$1/lib.so QT_REPS=5 python3 $R/tools/quick_time.py $2 $3 $4 $5 2>&1 | tail -1 | sed "s|$R/exp_build/||" >> $OUT; }
for rep in 1 2; do
for n in 5000 10000 100000; do
  for v in na_rd al_rd; do tm $v reentry LGL7 $n 0; done
  for v in na_tb al_tb; do tm $v twobody_lt LGL5 $n 1; done
done
done
cd /tmp && export TMPDIR=/tmp
pmc() {
  export ASSET_HIP_LIB=$R/exp_build/$1/lib.so QT_REPS=1 QT_ITERS=10 QT_WARMUP=2
  rm -rf /tmp/pmc_$1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_$1 -o q -- python3 $R/tools/quick_time.py $2 $3 $4 $5 > /tmp/pmc_$1.log 2>&1
  python3 - >> $OUT <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pmc_$1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'lgl_resident_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
print('pmc $1 $2 $3 x$4', {k: (round(sum(x)/len(x)*1024/1e6, 2), len(x)) for k,x in acc.items()}, 'MB per launch')
PY
}
for v in na_rd al_rd; do pmc $v reentry LGL7 10000 0; done
for v in na_tb al_tb; do pmc $v twobody_lt LGL5 10000 1; done
cat $OUT
