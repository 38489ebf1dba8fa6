#!/bin/bash
# C4 (10 000 blobs, 4K frame) at several tessellations of the blob: the same frame, the same rays to first order, another working set of
# nodes + triangle records (57 MB at 100 triangles per blob ... 573 MB at 1 000 = BASELINE config 4, 1.15 GB at 2 000).  Per point: the
# bench line (Msamples/s, rays/s, visits and tests per ray), and four --pmc passes of the same command (kernel trace only, one counter set
# per pass): L2 hit rate, fabric bytes (FETCH_SIZE / WRITE_SIZE), read latency seen by the L1.  GPU box, repo root:
#     bash tools/c4_working_set_sweep.sh "100 250 500 1000"    -> gpurun_out/c4_ws/<tris>/..., one summary line per point on stdout
# A pass that fails ends the script (exit 1, log tail printed); no further GPU step is started.
set -o pipefail
points=${1:-"100 250 500 1000"}
out=gpurun_out/c4_ws; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for t in $points; do
  d=$out/$t; mkdir -p $d
  args="--config C4 --blob-tris $t --steps 6 --warmup 2 --no-projection --no-cpu-baseline"
  if ! timeout -k 10 400 python3 bench.py $args --repeats 3 > $d/bench.log 2> $d/bench.err; then echo "bench at $t FAILED"; tail -n 8 $d/bench.err; exit 1; fi
  pass() {
    name=$1; shift
    if ! timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d/$name -o run -- python3 bench.py $args --repeats 1 --no-trace-phase > $d/$name.log 2>&1; then
      echo "PMC pass $name at $t FAILED"; tail -n 12 $d/$name.log; exit 1; fi
    [ -s $d/$name/run_counter_collection.csv ] || { echo "PMC pass $name at $t wrote no counters"; exit 1; }
  }
  pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE
  pass fetch FETCH_SIZE GRBM_GUI_ACTIVE
  pass write WRITE_SIZE GRBM_GUI_ACTIVE
  pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
  pass tcp3 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_READ_sum GRBM_GUI_ACTIVE
  python3 tools/c4_working_set_point.py $d $t
done
