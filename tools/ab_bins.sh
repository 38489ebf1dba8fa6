for l in hip b32 b64 hip b32 b64; do
  ELEVEN_HIP_LIB=$PWD/elevenrender_amd/libeleven_$l.so timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-trace-phase --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$l', d['value'], 'visits', r['node_visits_per_ray'], 'tests', r['tri_tests_per_ray'], 'nodes', d['accel']['nodes'], 'build_ms', d['accel']['build_ms'])"
done
