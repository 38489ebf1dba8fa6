#!/bin/bash
# A/B of environment-knob settings (tuning knobs, ELEVEN_HIP_LIB=<another build>) on ONE box, alternating: bash tools/ab_pool.sh reps "ENV=.. ENV=.." "ENV=.." ...
reps=$1; shift
for i in $(seq $reps); do
  for cfg in "$@"; do
    v=$(env $cfg timeout -k 10 150 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace-phase 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['trace_lanes'], 'DEGRADED' if d['degraded'] else '')")
    echo "[$cfg] $v"
  done
done
