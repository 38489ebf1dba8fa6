#!/bin/bash
# Round 6: a pixel that is behind its workgroup's most advanced one goes on in the slot it has (er_stream.hip s_front).  ER_STREAM_SPEC_KEEP = 1 + the samples it may be behind
# (0 = off); one box, settings alternate.   bash tools/ab_spec_keep.sh "0 1 2" 2 --sim-world 4 --steps 20 --warmup 5
set -o pipefail
settings=$1; reps=$2; shift 2
out=gpurun_out/ab_spec_keep; mkdir -p $out
i=0
for r in $(seq 1 $reps); do for sp in $settings; do
  i=$((i + 1))
  if ! ER_STREAM_SPEC_KEEP=$sp timeout -k 10 300 python3 bench.py --repeats 3 --no-cpu-baseline --no-trace-phase --no-projection "$@" > $out/$i.log 2> $out/$i.err; then echo "keep $sp FAILED"; tail -n 5 $out/$i.err; exit 1; fi
  python3 -c "
import json
d=json.loads(open('$out/$i.log').read().strip().splitlines()[-1]); s=(d.get('stream') or {}); s=(s.get('waves'), s.get('lanes_busy'), s.get('speculation'))
print('keep $sp rep $r:', d['ms_per_step'], 'ms per pass', d['value'], 'Msamples/s', d['repeats']['values'], s)"
done; done
