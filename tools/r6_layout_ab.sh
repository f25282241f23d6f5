#!/bin/bash
# round 6: the reference's slot order (kl0) against the J | H layout (kl1), side libraries exp_build/<kl>_<form>/lib.so built by
# tools/build_one.py.  Times from HIP events (tools/quick_time.py), then WRITE_SIZE / FETCH_SIZE per launch (rocprofv3 --pmc, own run).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6_layout_ab.txt
mkdir -p $R/gpurun_out; : > $OUT
tm() { # lib ode mode nseg blocked
  ASSET_HIP_LIB=$R/exp_build/$1/lib.so QT_REPS=4 python3 $R/tools/quick_time.py $2 $3 $4 $5 2>&1 | tail -1 | sed "s|$R/exp_build/||" >> $OUT
}
for rep in 1 2; do
for v in kl0_tile kl1_tile kl0_rd kl1_rd; do tm $v reentry LGL7 10000 0; done
for v in kl0_tile kl1_tile kl0_rd kl1_rd; do tm $v reentry LGL7 5000 0; done
for v in kl0_tb kl1_tb; do tm $v twobody_lt LGL5 10000 1; done
done
for v in kl0_tile kl1_tile kl0_rd kl1_rd; do tm $v reentry LGL7 100000 0; done
for v in kl0_tb kl1_tb; do tm $v twobody_lt LGL5 100000 1; done
cd /tmp && export TMPDIR=/tmp
pmc() { # lib ode mode nseg blocked
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
print('pmc $1 $2 $3 x$4', {k: (round(sum(x)/len(x)*1024/1e6, 2), len(x)) for k,x in acc.items()}, 'MB per launch (counter in KiB, tools/summarize_profiles.py)')
PY
}
for v in kl0_tile kl1_tile kl0_rd kl1_rd; do pmc $v reentry LGL7 10000 0; done
for v in kl0_tb kl1_tb; do pmc $v twobody_lt LGL5 10000 1; done
cat $OUT
