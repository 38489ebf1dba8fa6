#!/bin/bash
# Memory-pipeline counters of the wavefront kernels (separate --pmc passes, kernel trace only; kernels run serialised
# under counter collection, so every figure is "this kernel alone on the chip").  On the GPU box, from the repo root:
#   gpurun -- 'bash tools/pmc_mem.sh r02'   then   python tools/pmc_mem_summary.py gpurun_out/pmc_mem_r02
set -o pipefail
tag=${1:-r02}
out=gpurun_out/pmc_mem_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() {
  name=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-trace-phase > $out/$name.log 2>&1 || { echo "pass $name FAILED (see $out/$name.log)"; exit 1; }
  echo "pass $name done"
}
pass ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp_stall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp_req TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum GRBM_GUI_ACTIVE
pass tcc TCC_BUSY_avr TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY
pass sq2 SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE
pass td TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
ls $out
