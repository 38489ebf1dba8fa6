#!/bin/bash
# Device time of ONE launch of the streaming kernel against the number of sample passes in it: slope = a marginal pass, intercept = the
# launch's fixed cost (start-up burst + drain).   bash tools/launch_time_vs_steps.sh [--sim-world 8]   (one GPU box, repo root)
set -eo pipefail
out=gpurun_out/launch_time; mkdir -p $out
for n in 1 2 4 8 16 32; do
  timeout -k 10 300 python3 bench.py --steps $n --warmup 2 --repeats 3 --no-projection --no-cpu-baseline --no-trace-phase "$@" > $out/s$n.log 2> $out/s$n.err
  python3 -c "
import json
d=json.loads(open('$out/s$n.log').read().strip().splitlines()[-1]); r=d['repeats']
print('steps $n  launch ms', sorted(r['region_ms'])[1], ' ms/step', round(sorted(r['region_ms'])[1]/$n,4), ' Msamples/s', d['value'])"
done
