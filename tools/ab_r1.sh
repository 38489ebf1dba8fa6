#!/bin/bash
# round-1 tree (_r1/, `git archive 910d088`) against the current tree on ONE box: full-frame wavefront and the fused
# schedule at 1/8 of the frame (the multi-GPU rank size).
for i in 1 2 3; do
  for T in _r1 .; do
    for args in "" "--sim-world 8"; do
      v=$(cd $T && timeout -k 10 150 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['schedule'])")
      echo "$T [$args]: $v"
    done
  done
done
