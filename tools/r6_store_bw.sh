#!/bin/bash
# Round 6: what the write path sustains by stream length (tools/ubench_store.hip: persistent single-wave workgroups writing 8 064-byte blocks):
# 80 MB (the north-star launch: fits the 256 MB memory-side cache), 800 MB, 8 GB (the million-segment launch)
hipcc --offload-arch=gfx950 -O3 tools/ubench_store.hip -o gpurun_out/ubench_store 2>/dev/null
for n in 10000 100000 1000000; do ./gpurun_out/ubench_store $n 2>&1 | head -14; done > gpurun_out/r6_store_bw.txt
cat gpurun_out/r6_store_bw.txt
