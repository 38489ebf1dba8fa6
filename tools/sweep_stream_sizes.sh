run() {
  label=$1; shift
  v=$(timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-trace-phase "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$label: $v"
}
for res in "2560 1440" "3840 2160" "1280 720" "640 360"; do set -- $res; for s in stream wavefront fused; do run "C2 ${1}x${2} $s" --width $1 --height $2 --schedule $s; done; done
