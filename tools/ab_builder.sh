set -o pipefail
for cfg in "C2 --steps 20 --warmup 5" "C4 --config C4 --steps 6 --warmup 4"; do
  set -- $cfg; name=$1; shift
  for b in ${BUILDERS:-host gpu host gpu}; do
    extra=""; [ $b = gpu ] && extra="--gpu-build"; [ $b = lbvh ] && { extra="--gpu-build"; export ER_GPU_BUILD_LBVH=1; }; [ $b != lbvh ] && unset ER_GPU_BUILD_LBVH
    ER_GPU_BUILD_VERBOSE=1 timeout -k 10 500 python3 bench.py "$@" $extra --repeats 3 --no-cpu-baseline --no-projection > gpurun_out/abb_${name}_${b}.log 2> gpurun_out/abb_${name}_${b}.err || { echo FAILED $name $b; tail -3 gpurun_out/abb_${name}_${b}.err; exit 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/abb_${name}_${b}.log').read().strip().splitlines()[-1]); r=d['roofline']; a=d['accel']
print('$name $b', d['value'], 'Msamples/s  visits', r['node_visits_per_ray'], 'tests', r['tri_tests_per_ray'], 'nodes', a['nodes'], 'depth', a['max_depth'], 'build_ms', a['build_ms'], 'upload_ms', a['upload_ms'])"
    grep "er_gpu_build" gpurun_out/abb_${name}_${b}.err | head -3 || true
  done
done
