#!/bin/bash
# rocprofv3 --pmc passes of bench.py's dominant kernel (kernel trace only, one counter set per pass; never combined with other
# trace domains).  GPU box, repo root:
#     bash tools/pmc_passes.sh TAG SET [bench.py arguments ...]       -> gpurun_out/pmc_TAG/<pass>/run_counter_collection.csv
#     SET = full (16 passes) | core (11 passes: wave time split, lanes, L2, fabric bytes, address unit, L1 stalls)
#         | check (every counter set of `full` on a one-kernel torch process instead of the bench: says which sets the hardware
#           accepts, in seconds each, before any bench run is spent on them; a refused set is reported and the check goes on --
#           the process it aborts has done nothing)
# then   python tools/pmc_table.py gpurun_out/pmc_TAG > profiles/<round>_pmc_<what>.txt
#
# Every pass holds at most what the hardware takes in one pass (MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ 8, TCC 4 with
# FETCH_SIZE = 3 and WRITE_SIZE = 2, GRBM 2; TA / TD / TCP: two or three of one block -- round 3's scripts asked for four TA
# counters and for three TD counters in one pass, rocprofiler refused ("Request exceeds the capabilities of the hardware to
# collect") and ABORTED the process after it had initialised the GPU, and `|| echo` hid it).  A pass that fails now stops the
# script: its log tail is printed, the exit code is 1 and NO further GPU step is started.
set -o pipefail
tag=${1:?tag}; set_=${2:?full|core}; shift 2
out=gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--steps 6 --warmup 2 --repeats 1 --no-projection --no-cpu-baseline --no-trace-phase $*"
pass() {
  name=$1; shift
  if [ "$set_" = check ]; then
    if timeout -k 10 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/check_$name -o run -- python3 -c "import torch; x = torch.ones(4096, device='cuda'); print(float((x + 1).sum()))" > $out/check_$name.log 2>&1 \
       && [ -s $out/check_$name/run_counter_collection.csv ]; then echo "set $name accepted: $*"; else echo "set $name REFUSED: $*"; grep -m 3 -i "error\|exceeds" $out/check_$name.log; fi
    return 0
  fi
  if ! timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o run -- python3 bench.py $args > $out/$name.log 2>&1; then
    echo "PMC pass $name FAILED (counters: $*); log tail:"; tail -n 15 $out/$name.log
    exit 1
  fi
  if [ ! -s $out/$name/run_counter_collection.csv ]; then echo "PMC pass $name wrote no counters"; tail -n 15 $out/$name.log; exit 1; fi
  echo "pass $name done"
}
pass sq  SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVES SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
pass sq3 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
pass write WRITE_SIZE GRBM_GUI_ACTIVE
pass tcp_stall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass ta1 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
pass tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE
if [ "$set_" = full ] || [ "$set_" = check ]; then
  pass icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass ta3 TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
  pass tcp4 TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE
  pass td1 TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE
  pass td2 TD_LOAD_WAVEFRONT_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE
fi
ls $out
