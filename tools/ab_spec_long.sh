#!/bin/bash
# Round 6: speculative successors for EVERY sample of the pixels whose paths are long (er_stream.hip ST_LONG_SHIFT).  ER_STREAM_SPEC_LONG = sixteenths of
# max_bounces from which a pixel counts as long (0 = off); one box, settings alternate.   bash tools/ab_spec_long.sh "0 10 8" 2 --sim-world 8 --steps 20 --warmup 5
set -o pipefail
settings=$1; reps=$2; shift 2
out=gpurun_out/ab_spec_long; mkdir -p $out
i=0
for r in $(seq 1 $reps); do for sp in $settings; do
  i=$((i + 1))
  if ! ER_STREAM_SPEC_LONG=$sp timeout -k 10 300 python3 bench.py --repeats 3 --no-cpu-baseline --no-trace-phase --no-projection "$@" > $out/$i.log 2> $out/$i.err; then echo "long $sp FAILED"; tail -n 5 $out/$i.err; exit 1; fi
  python3 -c "
import json
d=json.loads(open('$out/$i.log').read().strip().splitlines()[-1]); s=(d.get('stream') or {}); s=(s.get('waves'), s.get('lanes_busy'), s.get('speculation'))
print('long $sp rep $r:', d['ms_per_step'], 'ms per pass', d['value'], 'Msamples/s', d['repeats']['values'], s)"
done; done
