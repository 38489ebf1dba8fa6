#!/bin/bash
# Read / write request sizes the L2s send to the fabric (TCC_EA0_RDREQ by size), streaming against wavefront schedule.  GPU box, repo root.
# Counters in their own passes with --kernel-trace only; the first pass that fails ends the script with its exit code and its log's tail
# (an `|| echo failed` here once hid an aborted pass: VERDICT r5).
set -eo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_rdreq
mkdir -p $out
pass() {  # name, counters, bench args...
  name=$1; counters=$2; shift 2
  if ! timeout -k 10 200 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $out/$name -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-trace-phase "$@" > $out/$name.log 2>&1; then
    echo "pmc_request_sizes: pass $name FAILED"; tail -n 8 $out/$name.log; exit 1
  fi
}
pass stream "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
pass wf "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" --schedule wavefront
pass streamw "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
ls $out
