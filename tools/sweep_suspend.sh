run() {
label=$1; shift
v=$(env "$@" timeout -k 10 150 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace-phase 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_ms'], r['trace_ms_total'], r['shade_ms_total'], r['launches'], r['empty_launches'], r.get('trace_lanes'))")
echo "$label: $v"
}
run "suspend 0" ER_TRACE_SUSPEND=0
run "suspend 40" ER_TRACE_SUSPEND=40
run "suspend 65" ER_TRACE_SUSPEND=65
run "suspend 0 refill 4" ER_TRACE_SUSPEND=0 ER_TRACE_REFILL_MIN=4
run "suspend 0 refill 1" ER_TRACE_SUSPEND=0 ER_TRACE_REFILL_MIN=1
run "suspend 0 refill 32" ER_TRACE_SUSPEND=0 ER_TRACE_REFILL_MIN=32
run "suspend 0 pools 1" ER_TRACE_SUSPEND=0 ER_WF_POOLS=1
