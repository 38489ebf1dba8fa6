#!/bin/bash
run() { label=$1; shift
  v=$(env "$@" timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sim-world 8 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['schedule'])")
  echo "$label: $v"; }
run "default 12" X=1
for w in 4 6 8 10 16; do run "fused waves/CU $w" ER_FUSED_WAVES_PER_CU=$w; done
run "wavefront forced" X=1
