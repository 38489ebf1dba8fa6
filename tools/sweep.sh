#!/bin/bash
# Tuning sweep of the wavefront schedule's launch shape on the GPU box (persistent trace / shade waves per CU, slot pools).
# Usage (GPU box): bash tools/sweep.sh > gpurun_out/sweep.log
run() {   # label, env...
  label=$1; shift
  v=$(env "$@" timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace-phase 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_ms'], r['trace_ms_total'], r['shade_ms_total'])")
  echo "$label: $v"
}
run "default (pools 3, trace 12, shade 5)"
run "default again"
for t in 8 10 14 16; do run "trace $t" ER_TRACE_WAVES_PER_CU=$t; done
for s in 3 4 6 8; do run "shade $s" ER_SHADE_WAVES_PER_CU=$s; done
for p in 2 4; do run "pools $p" ER_WF_POOLS=$p; done
run "pools 4 trace 10 shade 4" ER_WF_POOLS=4 ER_TRACE_WAVES_PER_CU=10 ER_SHADE_WAVES_PER_CU=4
run "pools 2 trace 14 shade 6" ER_WF_POOLS=2 ER_TRACE_WAVES_PER_CU=14 ER_SHADE_WAVES_PER_CU=6
run "no profile events" ER_BENCH_NO_PROFILE=1
