#!/bin/bash
# A/B of environment settings of ONE library on ONE box: every setting runs the same bench line, the list is repeated `reps` times so that the
# settings alternate (box-to-box noise is +-4 %, run-to-run on one box +-0.5 %).
#   bash tools/ab_env.sh "ER_NODE8_PACKED=4096 ER_NODE8_PACKED=999999999" 2 --config C4 --steps 6 --warmup 2     (a setting "-" = none)
set -o pipefail
settings=$1; reps=$2; shift 2
out=gpurun_out/ab_env; mkdir -p $out
i=0
for r in $(seq 1 $reps); do
  for s in $settings; do
    i=$((i + 1))
    e=$s; [ "$s" = "-" ] && e="ER_AB_NONE=1"
    if ! env $e timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-projection "$@" > $out/run$i.log 2> $out/run$i.err; then echo "$s FAILED"; tail -n 5 $out/run$i.err; exit 1; fi
    python3 -c "
import json
d=json.loads(open('$out/run$i.log').read().strip().splitlines()[-1]); r=d['roofline']; t=r.get('trace_lanes') or {}
print('$s', 'rep $r', d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step frac', r['frac'], 'visits', r.get('node_visits_per_ray'), 'node_bytes', d['accel']['node_bytes'] * d['accel']['nodes'])"
  done
done
