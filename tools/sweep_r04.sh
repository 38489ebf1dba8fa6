#!/bin/bash
# Round-4 knob sweeps on the GPU box (one process at a time, each under its own timeout; stops at the first failure).
#   bash tools/sweep_r04.sh sim8   -> one GPU's 1/8 share of the C2 frame: tracer/shader split at 16, 12 and 8 waves per CU
#   bash tools/sweep_r04.sh c4     -> C4: tracer/shader split, ring-visit and batch thresholds
set -o pipefail
what=${1:?sim8|c4|c4w|shares|knobs|retune|knobs13|diet|supertile|supertile2|supertile3|level|edgeauto|sharesedge|texcompact}
out=gpurun_out/sweep_r04_$what
mkdir -p $out
line() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step', 'lanes', (r.get('trace_lanes') or {}).get('busy'), d['config']['schedule'])
"; }
run() {  # name, env assignments..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  if ! env "${envs[@]}" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-trace-phase --no-projection "$@" > $out/$name.log 2> $out/$name.err; then echo "$name FAILED"; tail -n 5 $out/$name.err; exit 1; fi
  line $out/$name.log "$name ${envs[*]}"
}
if [ "$what" = sim8 ]; then
  S="--sim-world 8 --steps 20 --warmup 5"
  for t in 6 8 10 11 12 13; do run w16_t$t ER_STREAM_TRACERS=$t -- $S; done
  for t in 6 8 9 10; do run w12_t$t ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_w12.so ER_STREAM_TRACERS=$t -- $S; done
  for t in 4 5 6; do run w8_t$t ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_w8.so ER_STREAM_TRACERS=$t -- $S; done
  run w16_t12_batch16 ER_STREAM_TRACERS=12 ER_STREAM_BATCH_MIN=16 ER_STREAM_FIN_MIN=16 -- $S
  run w12_t9_batch16 ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_w12.so ER_STREAM_TRACERS=9 ER_STREAM_BATCH_MIN=16 ER_STREAM_FIN_MIN=16 -- $S
elif [ "$what" = c4 ]; then
  S="--config C4 --steps 6 --warmup 1"
  for t in 10 11 12 13; do run t$t ER_STREAM_TRACERS=$t -- $S; done
  run t12_refill4 ER_STREAM_TRACERS=12 ER_STREAM_REFILL_MIN=4 -- $S
  run t12_refill24 ER_STREAM_TRACERS=12 ER_STREAM_REFILL_MIN=24 -- $S
  run t11_batch48 ER_STREAM_TRACERS=11 ER_STREAM_BATCH_MIN=48 ER_STREAM_FIN_MIN=48 -- $S
fi
if [ "$what" = c4w ]; then      # (appended) C4 at 12 and 8 waves per CU: the shader step spills 34 / 0 registers instead of 111
  S="--config C4 --steps 6 --warmup 1"
  run w16_t12 ER_STREAM_TRACERS=12 -- $S
  for t in 8 9; do run w12_t$t ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_w12.so ER_STREAM_TRACERS=$t -- $S; done
  for t in 4 5; do run w8_t$t ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_w8.so ER_STREAM_TRACERS=$t -- $S; done
fi
if [ "$what" = shares ]; then      # one GPU's 1/2, 1/4, 1/8 share of the C2 frame at 16 and 12 waves per CU (ER_STREAM_WAVES)
  for w in 2 3 4 6 8; do
    S="--sim-world $w --steps 20 --warmup 5"
    run s${w}_w16 ER_STREAM_WAVES=16 -- $S
    run s${w}_w12 ER_STREAM_WAVES=12 -- $S
  done
  run s8_w12_t8 ER_STREAM_WAVES=12 ER_STREAM_TRACERS=8 -- --sim-world 8 --steps 20 --warmup 5
  run s8_w12_t10 ER_STREAM_WAVES=12 ER_STREAM_TRACERS=10 -- --sim-world 8 --steps 20 --warmup 5
  run s8_default X=1 -- --sim-world 8 --steps 20 --warmup 5
fi
if [ "$what" = knobs ]; then      # C2 after the split wait: batch / refill thresholds again, shader-wave priority 2 and 3
  S="--steps 20 --warmup 5"
  run base X=1 -- $S
  run batch48 ER_STREAM_BATCH_MIN=48 -- $S
  run batch32 ER_STREAM_BATCH_MIN=32 -- $S
  run fin32 ER_STREAM_FIN_MIN=32 -- $S
  run fin48 ER_STREAM_FIN_MIN=48 -- $S
  run refill8 ER_STREAM_REFILL_MIN=8 -- $S
  run refill16 ER_STREAM_REFILL_MIN=16 -- $S
  run refill20 ER_STREAM_REFILL_MIN=20 -- $S
  run prio2 ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_p2.so -- $S
  run prio3 ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_p3.so -- $S
  run base2 X=1 -- $S
fi
if [ "$what" = retune ]; then      # after -fno-slp-vectorize: the tracer / shader split per config again (fixed splits), thresholds
  for t in 11 12 13; do run c2_t$t ER_STREAM_TRACERS=$t -- --steps 20 --warmup 5; done
  for t in 10 11 12; do run c5_t$t ER_STREAM_TRACERS=$t -- --config C5 --steps 12 --warmup 3; done
  for t in 11 12 13; do run c4_t$t ER_STREAM_TRACERS=$t -- --config C4 --steps 6 --warmup 2; done
  run c2_refill8 ER_STREAM_REFILL_MIN=8 -- --steps 20 --warmup 5
  run c2_refill16 ER_STREAM_REFILL_MIN=16 -- --steps 20 --warmup 5
  run c2_batch48 ER_STREAM_BATCH_MIN=48 -- --steps 20 --warmup 5
  for s in 8; do run sim8_w12 ER_STREAM_WAVES=12 -- --sim-world 8 --steps 20 --warmup 5; run sim8_w16 ER_STREAM_WAVES=16 -- --sim-world 8 --steps 20 --warmup 5; done
fi
if [ "$what" = knobs13 ]; then      # C2 at 13 + 3 after the shader diet: thresholds again
  S="--steps 20 --warmup 5"
  run base X=1 -- $S
  run refill8 ER_STREAM_REFILL_MIN=8 -- $S
  run refill16 ER_STREAM_REFILL_MIN=16 -- $S
  run refill20 ER_STREAM_REFILL_MIN=20 -- $S
  run batch48 ER_STREAM_BATCH_MIN=48 -- $S
  run batch32 ER_STREAM_BATCH_MIN=32 -- $S
  run fin32 ER_STREAM_FIN_MIN=32 -- $S
  run fin48 ER_STREAM_FIN_MIN=48 -- $S
  run xcd0 ER_STREAM_XCD_TILES=0 -- $S
  run base2 X=1 -- $S
fi
if [ "$what" = diet ]; then      # the host-precomputed constants against their device evaluation, one library, alternating (knobs of er_api.cpp)
  for i in 1 2; do
    run c2_base_$i X=1 -- --steps 20 --warmup 5
    run c2_cam_on_device_$i ER_CAM_TRIG_ON_DEVICE=1 -- --steps 20 --warmup 5
    run c2_mat_on_device_$i ER_MAT_PRE_ON_DEVICE=1 -- --steps 20 --warmup 5
    run c2_wrap_by_division_$i ER_TEX_POW2=0 -- --steps 20 --warmup 5
    run c2_all_three_$i ER_CAM_TRIG_ON_DEVICE=1 ER_MAT_PRE_ON_DEVICE=1 ER_TEX_POW2=0 -- --steps 20 --warmup 5
  done
  run c5_base X=1 -- --config C5 --steps 12 --warmup 3
  run c5_wrap_by_division ER_TEX_POW2=0 -- --config C5 --steps 12 --warmup 3
  run c5_mat_on_device ER_MAT_PRE_ON_DEVICE=1 -- --config C5 --steps 12 --warmup 3
  run c5_all_three ER_CAM_TRIG_ON_DEVICE=1 ER_MAT_PRE_ON_DEVICE=1 ER_TEX_POW2=0 -- --config C5 --steps 12 --warmup 3
fi
if [ "$what" = supertile ]; then      # edge of the per-XCD super-tile (in 8 x 8 tiles) on the 4K frame and on C2
  for t in 8 4 16 32; do run c4_st$t ER_STREAM_SUPER_TILE=$t -- --config C4 --steps 6 --warmup 2; done
  for t in 8 4 16; do run c2_st$t ER_STREAM_SUPER_TILE=$t -- --steps 20 --warmup 5; done
  run c4_roundrobin ER_STREAM_XCD_TILES=0 -- --config C4 --steps 6 --warmup 2
fi
if [ "$what" = supertile2 ]; then      # finer: C2, C4, C5, one GPU's eighth
  for t in 8 12 16 20 24 32 48; do run c2_st$t ER_STREAM_SUPER_TILE=$t -- --steps 20 --warmup 5; done
  for t in 16 24 32 48 64; do run c4_st$t ER_STREAM_SUPER_TILE=$t -- --config C4 --steps 6 --warmup 2; done
  for t in 8 16 24 32; do run c5_st$t ER_STREAM_SUPER_TILE=$t -- --config C5 --steps 12 --warmup 3; done
  for t in 8 16 32; do run sim8_st$t ER_STREAM_SUPER_TILE=$t -- --sim-world 8 --steps 20 --warmup 5; done
  for t in 8 16 32; do run sim2_st$t ER_STREAM_SUPER_TILE=$t -- --sim-world 2 --steps 20 --warmup 5; done
fi
if [ "$what" = supertile3 ]; then      # after the deal levels the XCDs tile by tile
  for t in 8 16 24 32 48 64; do run c2_st$t ER_STREAM_SUPER_TILE=$t -- --steps 20 --warmup 5; done
  for t in 8 16 32 48 64 96; do run c4_st$t ER_STREAM_SUPER_TILE=$t -- --config C4 --steps 6 --warmup 2; done
  for t in 8 16 32 64; do run c5_st$t ER_STREAM_SUPER_TILE=$t -- --config C5 --steps 12 --warmup 3; done
  for t in 8 16 32; do run sim8_st$t ER_STREAM_SUPER_TILE=$t -- --sim-world 8 --steps 20 --warmup 5; done
  for t in 8 16 32; do run sim2_st$t ER_STREAM_SUPER_TILE=$t -- --sim-world 2 --steps 20 --warmup 5; done
fi
if [ "$what" = level ]; then      # the deal's tile-by-tile levelling of the XCDs on / off at the default super-tile edge and at 16
  for i in 1 2 3; do
    run c2_level_$i X=1 -- --steps 20 --warmup 5
    run c2_nolevel_$i ER_STREAM_LEVEL_XCDS=0 -- --steps 20 --warmup 5
  done
  for i in 1 2; do
    run c2_st16_level_$i ER_STREAM_SUPER_TILE=16 -- --steps 20 --warmup 5
    run c2_st16_nolevel_$i ER_STREAM_SUPER_TILE=16 ER_STREAM_LEVEL_XCDS=0 -- --steps 20 --warmup 5
  done
  run c4_level X=1 -- --config C4 --steps 6 --warmup 2
  run c4_nolevel ER_STREAM_LEVEL_XCDS=0 -- --config C4 --steps 6 --warmup 2
fi
if [ "$what" = edgeauto ]; then      # the deal that starts on 16 x 16-tile regions and falls back by the XCDs' measured finish times, against the fixed default edge
  show() { grep -h "XCDs finished" $out/$1.err | sed 's/^/   /'; }
  for i in 1 2; do
    run c2_auto_$i ER_STREAM_VERBOSE=1 -- --steps 20 --warmup 5; show c2_auto_$i
    run c2_edge8_$i ER_STREAM_VERBOSE=1 ER_STREAM_SUPER_TILE=8 -- --steps 20 --warmup 5; show c2_edge8_$i
  done
  run c4_auto ER_STREAM_VERBOSE=1 -- --config C4 --steps 6 --warmup 2; show c4_auto
  run c4_edge8 ER_STREAM_VERBOSE=1 ER_STREAM_SUPER_TILE=8 -- --config C4 --steps 6 --warmup 2; show c4_edge8
  run c5_auto ER_STREAM_VERBOSE=1 -- --config C5 --steps 12 --warmup 3; show c5_auto
  run c5_edge8 ER_STREAM_VERBOSE=1 ER_STREAM_SUPER_TILE=8 -- --config C5 --steps 12 --warmup 3; show c5_edge8
  run c5nl_auto ER_STREAM_VERBOSE=1 -- --config C5 --no-lights --steps 12 --warmup 3; show c5nl_auto
  run sim2_auto ER_STREAM_VERBOSE=1 -- --sim-world 2 --steps 20 --warmup 5; show sim2_auto
fi
if [ "$what" = sharesedge ]; then      # a GPU's half and quarter of the C2 frame on 8 x 8 and 16 x 16-tile regions per XCD (fixed), alternating
  for i in 1 2; do
    for w in 2 4; do
      run s${w}_st8_$i ER_STREAM_VERBOSE=1 ER_STREAM_SUPER_TILE=8 -- --sim-world $w --steps 20 --warmup 5
      run s${w}_st16_$i ER_STREAM_VERBOSE=1 ER_STREAM_SUPER_TILE=16 -- --sim-world $w --steps 20 --warmup 5
      grep -h "XCDs finished" $out/s${w}_st16_$i.err | head -2 | sed 's/^/   /'
    done
  done
fi
if [ "$what" = texcompact ]; then      # scalar-only textures kept with one channel (default) against every texture as it came, alternating
  for i in 1 2 3; do
    run c5_compact_$i X=1 -- --config C5 --steps 12 --warmup 3
    run c5_as_is_$i ER_TEX_COMPACT=0 -- --config C5 --steps 12 --warmup 3
    run c5_unfused_$i ER_TEX_FUSE=0 -- --config C5 --steps 12 --warmup 3
  done
  for i in 1 2; do
    run c5nl_compact_$i X=1 -- --config C5 --no-lights --steps 12 --warmup 3
    run c5nl_as_is_$i ER_TEX_COMPACT=0 -- --config C5 --no-lights --steps 12 --warmup 3
    run c5nl_unfused_$i ER_TEX_FUSE=0 -- --config C5 --no-lights --steps 12 --warmup 3
  done
fi
