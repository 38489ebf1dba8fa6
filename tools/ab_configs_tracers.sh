#!/bin/bash
# tracer / shader split per scene: bash tools/ab_configs_tracers.sh "C5 12" "C5 11" "C5nl 12" ...   (GPU box, repo root)
for cfg in "$@"; do set -- $cfg; c=$1; extra=""; if [ "$c" = "C5nl" ]; then c=C5; extra="--no-lights"; fi
  v=$(ER_STREAM_TRACERS=$2 timeout -k 10 300 python3 bench.py --config $c $extra --steps 8 --warmup 2 --no-cpu-baseline --no-trace-phase 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['trace_lanes'])"); echo "[$cfg] $v"; done
