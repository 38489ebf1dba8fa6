#!/bin/bash
# The streaming schedule's parity tests once under each CORNER setting of its tuning knobs (every setting must give the same bits:
# the knobs only move work between waves and batches).  One process per setting, distinct settings, no repetition.
# GPU box, repo root: bash tools/knob_corners.sh   -> gpurun_out/knob_corners.log
set -o pipefail
mkdir -p gpurun_out
# (the files whose tests force ER_FLAG_STREAM: all schedules against each other and the oracle, lit scenes, state export / import)
for cfg in "ER_STREAM_TRACERS=13 ER_STREAM_WAVES=16 ER_STREAM_REFILL_MIN=64" "ER_STREAM_TRACERS=1 ER_STREAM_REFILL_MIN=1" "ER_STREAM_WAVES=12 ER_STREAM_TRACERS=11 ER_STREAM_BATCH_MIN=3" "ER_STREAM_TRACERS=6 ER_STREAM_FIN_MIN=1 ER_STREAM_BATCH_MIN=1" "ER_STREAM_XCD_TILES=1 ER_STREAM_SUPER_TILE=2 ER_STREAM_FIN_MIN=17" "ER_STREAM_XCD_TILES=1 ER_STREAM_SUPER_TILE=16 ER_STREAM_LEVEL_XCDS=0 ER_CAM_TRIG_ON_DEVICE=1 ER_MAT_PRE_ON_DEVICE=1 ER_TEX_POW2=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 280 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_lights.py tests/test_gpu_state.py -m gpu -q -x 2>&1 | grep -E '^E  |FAILED|passed|failed' | tail -6 || { echo "FAILED under $cfg"; exit 1; }
done
echo "all corner settings bit-exact"
