#!/bin/bash
# Repeats the small-frame tests of the streaming schedule (few pixels per workgroup: rings turn over in microseconds, the
# timing-sensitive corner).  GPU box: bash tools/flaky_small.sh [repeats]
n=${1:-20}
fail=0
for i in $(seq $n); do
  timeout -k 10 300 python -m pytest tests/test_gpu_lights.py tests/test_gpu_parity.py -m gpu -x -q -k "(lit_scenes and 256) or fused_schedule or (edge_cases and 256)" > /tmp/fl.log 2>&1 || { fail=$((fail+1)); grep -E "assert|bit-exact|FAILED" /tmp/fl.log | tail -4; }
done
echo "repeats $n, failures $fail"
