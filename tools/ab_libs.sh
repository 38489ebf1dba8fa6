#!/bin/bash
# A/B of several builds of the library on ONE box: every build runs the same bench line, the list is repeated `reps` times so that
# the builds alternate (box-to-box noise is +-4 %, run-to-run on one box +-0.5 %).
#   bash tools/ab_libs.sh "v0 hip v2" 2 --steps 20 --warmup 5        (names = elevenrender_amd/libeleven_<name>.so)
set -o pipefail
libs=$1; reps=$2; shift 2
out=gpurun_out/ab_libs; mkdir -p $out
for r in $(seq 1 $reps); do
  for l in $libs; do
    if ! ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_$l.so timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-trace-phase --no-projection "$@" > $out/$l.$r.log 2> $out/$l.$r.err; then echo "$l FAILED"; tail -n 5 $out/$l.$r.err; exit 1; fi
    python3 -c "
import json
d=json.loads(open('$out/$l.$r.log').read().strip().splitlines()[-1]); r=d['roofline']; t=r.get('trace_lanes') or {}
print('$l', 'rep $r', d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step  lanes busy/node/tri', t.get('busy'), t.get('node'), t.get('tri'), 'wave_steps', t.get('wave_steps'))"
  done
done
