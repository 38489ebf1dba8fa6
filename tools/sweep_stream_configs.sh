run() {
  label=$1; shift
  v=$(timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-trace-phase "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('schedule'))")
  echo "$label: $v"
}
for c in C4 C5 C1; do for s in stream wavefront fused; do run "$c $s" --config $c --schedule $s; done; done
for w in 2 4 8; do for s in stream fused wavefront; do run "C2 sim-world $w $s" --sim-world $w --schedule $s; done; done
