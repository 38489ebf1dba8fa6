#!/bin/bash
# second launch-shape sweep (joint trace x shade waves per CU), one box
run() { label=$1; shift
  v=$(env "$@" timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace-phase 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_ms'], r['trace_ms_total'], r['shade_ms_total'])")
  echo "$label: $v"; }
run "default" X=1
for c in "8 8" "8 6" "10 6" "10 8" "12 6" "12 4" "16 4" "9 7"; do set -- $c; run "trace $1 shade $2" ER_TRACE_WAVES_PER_CU=$1 ER_SHADE_WAVES_PER_CU=$2; done
run "default again" X=1
