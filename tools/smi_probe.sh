#!/bin/bash
# power / clock samples while the C2 bench runs 3 x 64 steps (background sampler = one extra process, no GPU context)
( for i in $(seq 1 400); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|GPU use" | tr '\n' ' '; echo; sleep 0.05; done ) > gpurun_out/smi_samples.txt &
SMI=$!
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-trace-phase --steps 200 --warmup 20 > gpurun_out/smi_bench.log 2>&1
kill $SMI 2>/dev/null
wait $SMI 2>/dev/null
sort gpurun_out/smi_samples.txt | uniq -c | sort -rn | head -12
