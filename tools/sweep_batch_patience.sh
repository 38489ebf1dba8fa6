#!/bin/bash
# How long a shader wave of the streaming schedule waits for a full batch (ER_STREAM_BATCH_SPINS polls of ~0.2 us) before it takes a
# partial one, at one GPU's 1/8 and 1/4 share of the C2 frame, the whole frame and C1.   bash tools/sweep_batch_patience.sh "0 6 12 24 96"
set -o pipefail
list=${1:-"0 6 12 24 96 400"}
run() { # label env... -- args
  label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-trace-phase --no-projection --repeats 3 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step')"
}
for sp in $list; do run "sim8 spins=$sp" ER_STREAM_BATCH_SPINS=$sp -- --sim-world 8 --steps 20 --warmup 5; done
for sp in $list; do run "sim4 spins=$sp" ER_STREAM_BATCH_SPINS=$sp -- --sim-world 4 --steps 20 --warmup 5; done
for sp in $list; do run "full spins=$sp" ER_STREAM_BATCH_SPINS=$sp -- --steps 20 --warmup 5; done
for sp in $list; do run "C1 spins=$sp" ER_STREAM_BATCH_SPINS=$sp -- --config C1 --steps 16 --warmup 2; done
