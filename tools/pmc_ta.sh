#!/bin/bash
# PMC passes on the vector-memory front end (TA address unit, TCP tag pipe, TD data return) of the streaming kernel: is the CU's
# one-line-per-cycle address path the bound of the scattered node / triangle gathers?  Two or three counters per pass (a TA pass
# with four is refused: "exceeds the capabilities of the hardware").  GPU box, repo root: bash tools/pmc_ta.sh r03
set -o pipefail
tag=${1:-r03}
out=gpurun_out/pmc_ta_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() {
  name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o run -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-trace-phase --schedule stream > $out/$name.log 2>&1 || echo "pass $name failed"
  echo "pass $name done"
}
pass ta1 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass ta3 TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
pass tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE
pass tcp4 TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE
pass td TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE
ls $out
