#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 stats + PMC passes (tools/collect_profiles.sh) for every bench workload, summarised on the box
# (tools/summarize_profiles.py); only the summaries travel back: gpurun_out/profiles_<round>/ -> copy them into profiles/.
rnd=${1:-r4}
shift
wls=${@:-reentry_lgl7_10k reentry_lgl7_5k reentry_lgl7_1m betts_lgl5_1k twobody_lgl5_blocked_10k multispacecraft_8x1250 synthetic32_lgl7_12500 twobody_lgl7_10k betts_lgl7_5k reentry_trap_10k twobody_trap_blocked_10k}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/profiles_$rnd
for wl in $wls; do
  bash $R/tools/collect_profiles.sh ${rnd}_$wl $wl > $R/gpurun_out/${rnd}_$wl.log 2>&1
  (cd $R && python3 tools/summarize_profiles.py ${rnd}_$wl $rnd $wl > gpurun_out/profiles_$rnd/${wl}_summary.log 2>&1)
  cp $R/profiles/${rnd}_${wl}_* $R/gpurun_out/profiles_$rnd/ 2>/dev/null
  rm -rf $R/gpurun_out/${rnd}_$wl
  tail -2 $R/gpurun_out/profiles_$rnd/${wl}_summary.log | head -1 | cut -c1-120
done
ls $R/gpurun_out/profiles_$rnd | wc -l
