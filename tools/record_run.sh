#!/bin/bash
# Record run for profiles/ (one GPU box, repo root):   bash tools/record_run.sh r06 [bench|pmc_c2|pmc_c4|pmc_c5|all]
# (a gpurun call lasts 20 minutes at most: the stages are run as separate calls)
#   1. the bench line as the driver types it (C2), and rocprofv3 --kernel-trace --stats of the same command;
#   2. the other BASELINE scenes (C4, C5, C5 without the extensions, C1), one GPU's 1/2, 1/4, 1/8 share of the C2 frame, and the
#      two-rank rehearsal of `bench.py --gpus 2` (no launcher around it: the file starts its own ranks);
#   3. the PMC passes of the final kernel: C2 full set, C4 and C5 core set (tools/pmc_passes.sh; separate --pmc passes, kernel trace only).
# Every step under its own timeout; the script stops at the first failure (no GPU step is started after a failed one).
# Afterwards, in the build container:  python tools/pmc_traffic.py gpurun_out/pmc_<tag>_c2 <tag> C2  (and C4, C5),
#   python tools/pmc_table.py gpurun_out/pmc_<tag>_c2 > profiles/<tag>_pmc_c2_stream_kernel.txt, and copy the logs.
set -eo pipefail
tag=${1:-r06}
stage=${2:-all}
out=gpurun_out/record_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ $stage = bench ] || [ $stage = all ]; then
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/stats.log 2>&1
python3 tools/kernel_launches.py $out/stats/run_kernel_trace.csv er_stream_kernel > $out/kernel_launches.txt
echo "stats done"
timeout -k 10 300 python3 bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-projection > $out/bench_64.log 2> $out/bench_64.err
timeout -k 10 200 python3 bench.py --config C1 --steps 16 --warmup 2 > $out/bench_c1.log 2> $out/bench_c1.err
timeout -k 10 500 python3 bench.py --config C5 --steps 12 --warmup 3 > $out/bench_c5.log 2> $out/bench_c5.err
timeout -k 10 300 python3 bench.py --config C5 --no-lights --steps 12 --warmup 3 --no-cpu-baseline --no-projection > $out/bench_c5_nolights.log 2> $out/bench_c5_nolights.err
timeout -k 10 600 python3 bench.py --config C4 --steps 6 --warmup 4 > $out/bench_c4.log 2> $out/bench_c4.err
echo "configs done"
for s in 2 4 8; do
  timeout -k 10 200 python3 bench.py --sim-world $s --steps 20 --warmup 5 --no-cpu-baseline --no-projection > $out/bench_sim$s.log 2> $out/bench_sim$s.err
done
ER_BENCH_REHEARSAL=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > $out/bench_rehearsal2.log 2> $out/bench_rehearsal2.err
echo "shares + rehearsal done"
fi
if [ $stage = pmc_c2 ] || [ $stage = all ]; then bash tools/pmc_passes.sh ${tag}_c2 full > $out/pmc_c2.log 2>&1 || { tail -20 $out/pmc_c2.log; exit 1; }; fi
if [ $stage = pmc_c4 ] || [ $stage = all ]; then bash tools/pmc_passes.sh ${tag}_c4 core --config C4 > $out/pmc_c4.log 2>&1 || { tail -20 $out/pmc_c4.log; exit 1; }; fi
if [ $stage = pmc_c5 ] || [ $stage = all ]; then bash tools/pmc_passes.sh ${tag}_c5 core --config C5 > $out/pmc_c5.log 2>&1 || { tail -20 $out/pmc_c5.log; exit 1; }; fi
echo "pmc done"
for f in $out/bench*.log; do python3 -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('$f', d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'of measured', r['frac_of_measured'], 'n_gpus', d['n_gpus'], 'sched', d['config']['schedule'], (d.get('cpu_baseline') or {}).get('value'))
except Exception as e: print('$f FAILED', e)
"; done
