#!/bin/bash
# bench with each given build of the library, alternating, on ONE box: bash tools/ab_libs.sh reps lib1 lib2 ...
reps=$1; shift
for i in $(seq $reps); do
  for L in "$@"; do
    v=$(ELEVEN_HIP_LIB=$(realpath $L) timeout -k 10 150 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace-phase 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_ms'], r['trace_ms_total'], r['shade_ms_total'])")
    echo "$(basename $L): $v"
  done
done
