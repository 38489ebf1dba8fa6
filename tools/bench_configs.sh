#!/bin/bash
# The other BASELINE.json scenes through bench.py (--config), and the 2-rank rehearsal of the N>1 control flow on one GPU
# (two ranks share GPU 0 and talk over gloo; the driver's real N-GPU runs use RCCL through er_gather_pass).
# Usage (GPU box): bash tools/bench_configs.sh r02   -> gpurun_out/configs_r02/*.log
set -o pipefail
tag=${1:-r02}
out=gpurun_out/configs_$tag
mkdir -p $out
timeout -k 10 200 python3 bench.py --config C1 --steps 16 --warmup 2 > $out/c1.log 2> $out/c1.err
timeout -k 10 500 python3 bench.py --config C5 --steps 8 --warmup 2 > $out/c5.log 2> $out/c5.err
timeout -k 10 300 python3 bench.py --config C5 --no-lights --steps 8 --warmup 2 --no-cpu-baseline > $out/c5_nolights.log 2> $out/c5_nolights.err
timeout -k 10 500 python3 bench.py --config C4 --steps 6 --warmup 1 > $out/c4.log 2> $out/c4.err
ER_BENCH_REHEARSAL=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 8 --warmup 2 > $out/rehearsal2.log 2> $out/rehearsal2.err
for s in 2 4 8; do
  timeout -k 10 200 python3 bench.py --sim-world $s --steps 20 --warmup 5 --no-cpu-baseline > $out/sim$s.log 2> $out/sim$s.err
done
for f in $out/*.log; do echo "== $f"; python3 -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    r=d['roofline']; print(d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'sched', d['config']['schedule'], 'n_gpus', d['n_gpus'], 'gather', d.get('gather'))
except Exception as e: print('FAILED', e)
"; done
