#!/bin/bash
# experiment helper: build reduced libs (capi + reentry TU) with -DASSET_ABLATE=n into /root/repo/gpurun_ablate/
set -e
cd /root/repo/asset_asrl_amd/csrc
mkdir -p /root/repo/ablate
for n in ${ABL:-0 1 2}; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DASSET_ABLATE=$n -c gen/tu_reentry.hip -o /tmp/abl_tu_$n.o &&
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c capi.hip -o /tmp/abl_capi_$n.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/ablate/lib_$n.so /tmp/abl_tu_$n.o /tmp/abl_capi_$n.o ) &
done
wait
ls -la /root/repo/ablate
