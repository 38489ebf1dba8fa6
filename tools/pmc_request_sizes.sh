cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_rdreq
mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/stream -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-trace-phase > $out/stream.log 2>&1 || echo failed stream
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/wf -o run -- python3 bench.py --schedule wavefront --steps 4 --warmup 1 --no-cpu-baseline --no-trace-phase > $out/wf.log 2>&1 || echo failed wf
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $out/streamw -o run -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-trace-phase > $out/streamw.log 2>&1 || echo failed streamw
ls $out
