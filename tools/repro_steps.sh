#!/bin/bash
# The calls that exposed the lost-ray bug of the streaming schedule's first ring protocol (a 4-sample call, then a 64-sample one,
# on the C2 frame): every line must print a bench result, none an er_wait watchdog error.  GPU box: bash tools/repro_steps.sh
for k in 24 32 48 64; do
  echo "steps $k:"; timeout -k 10 120 python3 bench.py --steps $k --warmup 4 --no-cpu-baseline --no-trace-phase 2>&1 | tail -n 1 | cut -c1-200
done
echo "steps 64 warmup 0:"; timeout -k 10 120 python3 bench.py --steps 64 --warmup 0 --no-cpu-baseline --no-trace-phase 2>&1 | tail -n 1 | cut -c1-200
