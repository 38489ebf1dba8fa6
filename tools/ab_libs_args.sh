#!/bin/bash
# like ab_libs.sh with extra bench arguments: bash tools/ab_libs_args.sh reps "bench args" lib1 lib2 ...
reps=$1; shift
args=$1; shift
for i in $(seq $reps); do
  for L in "$@"; do
    v=$(ELEVEN_HIP_LIB=$(realpath $L) timeout -k 10 150 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace-phase $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], (r.get('trace_lanes') or {}).get('busy'))")
    echo "$(basename $L): $v"
  done
done
