#!/bin/bash
# A/B of two builds of the library on ONE box, alternating (box-to-box noise is +-1 ... 4 %, run-to-run on one box +-0.5 %).
#   bash tools/ab_lib.sh elevenrender_amd/libeleven_base.so elevenrender_amd/libeleven_hip.so 3 --steps 20 --warmup 5
set -o pipefail
a=$1; b=$2; reps=$3; shift 3
out=gpurun_out/ab_lib; mkdir -p $out
i=0
for r in $(seq 1 $reps); do for lib in $a $b; do
  i=$((i + 1))
  if ! ELEVEN_HIP_LIB=$PWD/$lib timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-projection --no-trace-phase "$@" > $out/$i.log 2> $out/$i.err; then echo "$lib FAILED"; tail -n 5 $out/$i.err; exit 1; fi
  python3 -c "
import json
d=json.loads(open('$out/$i.log').read().strip().splitlines()[-1])
print('$(basename $lib) rep $r:', d['value'], 'Msamples/s', d['ms_per_step'], 'ms/step', d['repeats']['values'], 'frac', d['roofline']['frac'])"
done; done
