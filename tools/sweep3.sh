#!/bin/bash
# Launch-shape sweep, second axis: FEWER trace waves per launch with MORE slot pools (each lane then works through more rays per
# launch, and the launches of different pools co-reside instead of queueing).  Prints Msamples/s, ms/step, per-launch ms,
# trace / shade ms totals and the share of busy lanes in the trace loop.
run() {
  label=$1; shift
  v=$(env "$@" timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace-phase 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_ms'], r['trace_ms_total'], r['shade_ms_total'], (r.get('trace_lanes') or {}).get('busy'))")
  echo "$label: $v"
}
run "default" ER_AB_NONE=1
for p in 3 4 6 8; do for t in 3 4 6; do for s in 2 3 5; do
  run "pools $p trace $t shade $s" ER_WF_POOLS=$p ER_TRACE_WAVES_PER_CU=$t ER_SHADE_WAVES_PER_CU=$s
done; done; done
run "default again" ER_AB_NONE=1
