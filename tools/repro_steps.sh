for k in 24 32 48 64; do
  echo "steps $k:"; timeout -k 10 120 python3 bench.py --steps $k --warmup 4 --no-cpu-baseline --no-trace-phase 2>&1 | tail -n 1 | cut -c1-200
done
echo "steps 64 warmup 0:"; timeout -k 10 120 python3 bench.py --steps 64 --warmup 0 --no-cpu-baseline --no-trace-phase 2>&1 | tail -n 1 | cut -c1-200
