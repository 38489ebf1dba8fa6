#!/bin/bash
# Tuning sweep of the streaming schedule (er_stream.hip) on the GPU box: tracer waves of the 16, refill and batch thresholds.
run() {
  label=$1; shift
  v=$(env "$@" timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace-phase --schedule stream 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], (r.get('trace_lanes') or {}).get('busy'))")
  echo "$label: $v"
}
run "default (tracers 10, refill 12, batch 48)" ER_AB_NONE=1
for t in 8 9 11 12; do run "tracers $t" ER_STREAM_TRACERS=$t; done
for r in 4 8 24 32; do run "refill $r" ER_STREAM_REFILL_MIN=$r; done
for b in 16 32 56 64; do run "batch $b" ER_STREAM_BATCH_MIN=$b; done
run "default again" ER_AB_NONE=1
