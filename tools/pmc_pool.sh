#!/bin/bash
# PMC passes that compare the streaming schedule's two tracers (kernel trace only).  GPU box, repo root: bash tools/pmc_pool.sh tag
set -o pipefail
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for pool in 0 1; do
  out=gpurun_out/pmc_pool_${tag}_$pool
  mkdir -p $out
  pass() {
    name=$1; shift
    ER_STREAM_POOL=$pool timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -o run -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-trace-phase --schedule stream > $out/$name.log 2>&1 || echo "pass $name failed"
    echo "pool=$pool pass $name done"
  }
  pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
  pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE
  pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE
done
