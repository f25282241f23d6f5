#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh <tag> [workload]'): rocprofv3 kernel stats + PMC passes of bench.py.
# Outputs land in gpurun_out/<tag>/ ; tools/summarize_profiles.py turns them into profiles/<round>_*.{csv,json}.
# --pmc passes are separate runs and never combined with any trace domain other than the counter collection itself.
tag=${1:-prof}
wl=${2:-reentry_lgl7_10k}     # bench workload name
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --workload $wl"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --steps 200 --warmup 20 $3 > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $B --steps 20 --warmup 2 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $B --steps 20 --warmup 2 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/sq1 -o bench -- $B --steps 20 --warmup 2 > $O/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq2 -o bench -- $B --steps 20 --warmup 2 > $O/sq2.log 2>&1
tail -n 1 $O/stats.log | cut -c1-200
cat $O/stats/bench_kernel_stats.csv
