for L in "$@"; do
  v=$(ELEVEN_HIP_LIB=$(realpath $L) timeout -k 10 150 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-trace-phase 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['node_visits_per_ray'], r['tri_tests_per_ray'], d['accel']['nodes'], d['accel']['build_ms'])")
  echo "$(basename $L): $v"
done
