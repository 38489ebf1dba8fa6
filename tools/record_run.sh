#!/bin/bash
# Record run for profiles/: default bench line, rocprofv3 kernel statistics of the same command, and the HBM
# traffic counters (separate --pmc passes, kernel trace only).  Run on the GPU box from the repo root:
#   gpurun -- 'bash tools/record_run.sh r02 [bench args]'   then   python tools/make_profile_summary.py gpurun_out/record_r02 r02
set -eo pipefail
tag=${1:-r02}
shift || true
extra="$@"     # extra bench.py arguments for every run, e.g. --schedule wavefront
out=gpurun_out/record_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python3 bench.py $extra > $out/bench.log 2> $out/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py $extra > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o run -- python3 bench.py $extra --steps 4 --warmup 1 --no-cpu-baseline --no-trace-phase > $out/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o run -- python3 bench.py $extra --steps 4 --warmup 1 --no-cpu-baseline --no-trace-phase > $out/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_valu -o run -- python3 bench.py $extra --steps 4 --warmup 1 --no-cpu-baseline --no-trace-phase > $out/pmc_valu.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2 -o run -- python3 bench.py $extra --steps 4 --warmup 1 --no-cpu-baseline --no-trace-phase > $out/pmc_l2.log 2>&1
ls -R $out | head -40
cat $out/bench.log
