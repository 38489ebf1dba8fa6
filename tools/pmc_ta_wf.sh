#!/bin/bash
# The same front-end counters per ROLE: the wavefront schedule runs traversal (er_wf_trace) and shading (er_wf_shade) as separate
# kernels on the same code (er_trav.h, er_bounce.inc), so its per-kernel rows split what er_stream_kernel's single row cannot.
set -o pipefail
tag=${1:-r03}
out=gpurun_out/pmc_ta_wf_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() {
  name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-trace-phase --schedule wavefront > $out/$name.log 2>&1 || { echo "pass $name FAILED (see $out/$name.log)"; exit 1; }
  echo "pass $name done"
}
pass tcp2 TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE
pass ta1 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
ls $out
