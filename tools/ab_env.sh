#!/bin/bash
# bench with and without an environment switch, alternating, on ONE box: bash tools/ab_env.sh reps VAR=val [bench args]
reps=$1; shift
sw=$1; shift
for i in $(seq $reps); do
  for mode in base "$sw"; do
    if [ "$mode" = base ]; then e=ER_AB_NONE=1; else e=$sw; fi
    v=$(env $e timeout -k 10 150 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace-phase "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_ms'], r['trace_ms_total'], r['shade_ms_total'])")
    echo "$mode: $v"
  done
done
