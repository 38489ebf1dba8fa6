#!/bin/bash
# bash abcfg.sh "<bench args>" cfg1 cfg2 ...
args=$1; shift
for i in 1 2; do
  for cfg in "$@"; do
    v=$(env $cfg timeout -k 10 300 python3 bench.py $args --no-cpu-baseline --no-trace-phase 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['config']['schedule'])")
    echo "[$args | $cfg] $v"
  done
done
