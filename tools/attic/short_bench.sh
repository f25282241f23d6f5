#!/bin/bash
# One-off (round 6): what the driver's short run (--steps 20 --warmup 5) reports against the 1000-step default
for k in "20 5" "20 5" "20 5" "100 5" "1000 50"; do
  set -- $k
  python3 bench.py --steps $1 --warmup $2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('steps',d['steps'],'warmup',d['warmup'],'ms_per_step %.4f'%d['ms_per_step'],'launch_ms %.4f'%r['launch_ms'],'frac %.3f'%r['frac'])"
done
