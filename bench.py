#!/usr/bin/env python3
"""bench.py -- Msamples/s of the per-sample path-tracing hot path on MI355X.

Default workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): synthetic 1M random-triangle soup + 2048x1024
procedural sky HDRI, 1920x1080, max_bounces 8, seed 12345.  `--config C1|C4|C5` selects the other BASELINE scenes
(their numbers are reported with the same fields; `config.workload` names what ran).  A STEP is one sample pass (one
renderingKernel per pixel, reference src/kernel.cpp:689-700); the 256 spp of the config are 256 such steps.
`value` = executed bounce-loop iterations (the reference's `for (i...)` at src/kernel.cpp:508, counted on the device)
per second, whole job, scene and framebuffers resident in HBM when the timed region starts.

Roofline (dominant kernel: the traversal).  Algorithmic bytes follow SURVEY.md 8(d): 64 B per node visit + 36 B per
triangle test, event counts from an instrumented replay of the same samples.
  roofline.achieved = ALL traversal bytes of the timed region / the wall time of the timed region (nothing is scaled by
      a concurrency factor: the slot pools' launches overlap each other and the shade launches, so this is a lower
      bound on the rate "during traversal");
  roofline.trace_phase = the same bytes / the summed duration of the trace launches in a second pass of the same steps
      with ONE slot pool on ONE stream, where launches cannot overlap and the division is exact -- the rate while the
      traversal kernel has the chip to itself;
  roofline.peak = 8 TB/s (HBM3E spec, MI355X_MICROARCH.md); roofline.peak_measured = a streaming copy / read kernel timed
      in this job (er_measure_hbm_peak).

With --gpus N the frame's 8x8 pixel tiles are sharded over N ranks (one process per GPU: under torch.distributed.run when
the caller used it, otherwise `bench.py --gpus N` starts its own N ranks as a child process, launch_ranks()); there is no collective in the timed region -- the RCCL framebuffer gather (all five planes,
through the library's own C++ entry er_gather_pass) happens once after it and is reported as readback_ms.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_TAG = "r06"     # profiles/<tag>_pmc_traffic.json: HBM bytes per launch from the rocprofv3 PMC passes


def trace_bytes(c):
    """Algorithmic bytes of the traversal, SURVEY.md 8(d): 64 B per node visit + 36 B per triangle test (its
    per-pixel-sample formula restricted to the terms the trace kernel executes).  V and T are this build's own
    event counts (8-wide nodes, two-triangle leaves)."""
    return 64 * c["node_visits"] + 36 * c["tri_tests"]


def trace_bytes_layout(c):
    """The same events priced at this build's record sizes (80-byte ErNode8, 48-byte triangle record, 40 bytes
    of ray in / result out per ray) -- what the kernel actually asks the memory system for."""
    return 80 * c["node_visits"] + 48 * c["tri_tests"] + 40 * c["rays"]


def path_bytes(c, hdri_texels):
    """SURVEY.md 8(d), whole per-sample path: 64 V + 36 T + 112 H + 12 X + 4 ceil(log2 P) S + 112 per pixel-sample
    (H shaded hits, X texel fetches, S HDRI CDF samples of a P-texel HDRI; the trailing 112 B is the framebuffer,
    RNG and sample-count read-modify-write of one finished path)."""
    cdf_steps = max(1, math.ceil(math.log2(max(2, hdri_texels))))
    return (trace_bytes(c) + 112 * c["shaded_hits"] + 12 * c["texel_fetches"] + 4 * cdf_steps * c["hdri_samples"]
            + 112 * c["paths"])


def host_cores():
    """Cores this job may really use: the affinity mask, cut to the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(scene, max_bounces, flags, threads, budget_s=16.0, one_core_budget_s=7.0, one_core=True):
    """The oracle (a port of the reference algorithm: fixed-depth-18 BVH, unordered unpruned traversal) timed on this
    box's host cores, on a bounded sample of the same workload: once on `threads` threads, once on ONE."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    W, H = scene.x_res, scene.y_res

    def leg(nthreads, budget):
        o = oracle.Oracle(scene, math_mode=oracle.MATH_LIBM, max_bounces=max_bounces, threads=nthreads, flags=flags)

        def run_rows(rows):
            t0 = time.perf_counter()
            c0 = o.counters()["bounce_samples"]
            for y in rows:
                o.render(1, y * W, (y + 1) * W)
            return time.perf_counter() - t0, o.counters()["bounce_samples"] - c0

        n_probe = max(1, min(8, H // 8))
        probe_rows = list(range(H // (2 * n_probe), H, max(1, H // n_probe)))[:n_probe]     # rows spread over the frame
        t_probe, s_probe = run_rows(probe_rows)
        per_row = t_probe / len(probe_rows)
        n_rows = int(max(0, min(H - len(probe_rows), (budget - t_probe) / max(per_row, 1e-6))))
        rows = []
        if n_rows > 0:
            stride = max(1, H // n_rows)
            rows = [y for y in range(stride // 2, H, stride) if y not in probe_rows][:n_rows]
        t_main, s_main = run_rows(rows) if rows else (0.0, 0)
        build_s = o.build_seconds
        o.close()
        total_rows = len(probe_rows) + len(rows)
        return (s_probe + s_main) / (t_probe + t_main) / 1e6, total_rows, t_probe + t_main, build_s

    v_all, rows_all, t_all, build_s = leg(threads, budget_s)
    out = {"value": round(v_all, 6), "unit": "Msamples/s", "cores": threads, "kind": "port",
           "sample": f"{rows_all} rows x {W} px of the same frame at 1 spp (rows spread over the image), {t_all:.1f} s of work on "
                     f"{threads} threads (all cores this job may use: affinity mask cut to the cgroup quota; --cpu-threads overrides); "
                     f"reference-BVH build {build_s:.1f} s not timed"}
    if one_core:       # (a second oracle, i.e. a second reference-style BVH build: skipped where that build is the expensive part)
        v_one, rows_one, t_one, _ = leg(1, one_core_budget_s)
        out["one_core"] = {"value": round(v_one, 6), "cores": 1, "sample": f"{rows_one} rows x {W} px, {t_one:.1f} s on 1 thread"}
    return out


def launch_ranks(n):
    """`bench.py --gpus N` run without a launcher: start the N ranks (one process per GPU) with torch.distributed.run as a CHILD
    process, pass its output through (rank 0 prints the JSON line) and return its exit code.  Runs before this process has
    imported torch or touched HIP: nothing that has initialised the GPU is ever replaced (no exec), and the parent never
    initialises it at all.  Reference shape reproduced by the N ranks: one pass per sample over all pixels
    (src/kernel.cpp:680-706), the read-back of src/Managers.cpp:287-302 after the gather."""
    import socket
    import subprocess
    rehearsal = os.environ.get("ER_BENCH_REHEARSAL") == "1" or os.environ.get("ER_BENCH_DRY_RUN") == "1"
    if not rehearsal:
        import torch   # device_count() does not initialise the GPU on this image
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py --gpus {n}: this node shows {have} GPU(s); one process per GPU needs {n} "
                  f"(ER_BENCH_REHEARSAL=1 rehearses the control flow with all ranks on GPU 0 over gloo)", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def dry_run(rank, world):
    """ER_BENCH_DRY_RUN=1: the ranks join (gloo), agree on who is there, rank 0 prints one line; nothing touches a GPU.
    Lets the CPU test-suite see that `bench.py --gpus N` really becomes N ranks."""
    import torch
    import torch.distributed as dist
    joined = [rank]
    if world > 1:
        dist.init_process_group(backend="gloo")
        t = torch.zeros(world, dtype=torch.int64)
        t[rank] = rank + 1
        dist.all_reduce(t)
        joined = [int(v) - 1 for v in t.tolist()]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_joined": joined}), flush=True)
    return 0


CONFIGS = {
    # name: (description, default max_bounces, config spp)
    "C1": ("Cornell box, 12 triangles, 256x256", 4, 16),
    "C2": ("1M random-triangle soup + 2048x1024 sky HDRI, 1920x1080", 8, 256),
    "C4": ("10M triangles (10 000 instances of a 1 000-triangle smooth blob, flattened) + sky HDRI, 3840x2160", 8, 1024),
    "C5": ("1M-triangle soup, 64 textured materials (192 value-noise textures), 256 point lights, MIS on, 1920x1080", 16, 256),
}


def make_scene(args, scenes, abi):
    """The BASELINE.json scene `--config` names; returns (scene, extension flags, workload text)."""
    ext = 0
    if args.config == "C1":
        w, h = args.width or 256, args.height or 256
        sc = scenes.cornell(w, h)
        what = f"C1: {CONFIGS['C1'][0].replace('256x256', f'{w}x{h}')}"
    elif args.config == "C2":
        w, h = args.width or 1920, args.height or 1080
        sc = scenes.soup(args.tris, w, h, seed=12345)
        what = f"C2: {args.tris}-triangle random soup + 2048x1024 sky HDRI, {w}x{h}"
    elif args.config == "C4":
        w, h = args.width or 3840, args.height or 2160
        sc = scenes.blob_instances(tris_per_blob=args.blob_tris, x_res=w, y_res=h)
        what = f"C4: {sc.tri_count} triangles (10 000 instances of a {args.blob_tris}-triangle smooth-normal blob, flattened) + sky HDRI, {w}x{h}"
    else:
        w, h = args.width or 1920, args.height or 1080
        sc = scenes.torture(args.tris, w, h, seed=12345)
        ext = abi.FLAG_POINT_LIGHTS | abi.FLAG_MIS
        if args.no_lights:
            ext = 0
        what = (f"C5: {args.tris}-triangle soup, 64 textured materials, " +
                ("256 point lights + MIS (build-defined extensions, parity unpinned)" if ext else "reference behaviour (lights ignored, no MIS)") + f", {w}x{h}")
    return sc, ext, what


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2", help="BASELINE.json scene (default C2, the one the metric is quoted on)")
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--max-bounces", type=int, default=0, help="0 -> the config's own")
    ap.add_argument("--no-lights", action="store_true", help="C5 without the point-light / MIS extensions (reference behaviour)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=16.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline's all-core leg (0 = every core this job may use)")
    ap.add_argument("--no-trace-phase", action="store_true", help="skip the single-pool pass that measures the traversal phase alone")
    ap.add_argument("--per-step-launch", action="store_true", help="one launch per step instead of one launch for all K steps")
    ap.add_argument("--schedule", choices=["auto", "wavefront", "megakernel", "stream"], default="auto")
    ap.add_argument("--gpu-build", action="store_true", help="force the device BVH build (ER_FLAG_GPU_BUILD; the default for scenes of >= 20 000 triangles since round 5)")
    ap.add_argument("--host-build", action="store_true", help="force the host BVH build (ER_FLAG_HOST_BUILD)")
    ap.add_argument("--blob-tris", type=int, default=1000, help="(C4, diagnostic) triangles per blob: the same 10 000 blobs at another tessellation -- the same frame with "
                    "another working set (tools/c4_working_set_sweep.sh); 1000 is BASELINE config 4")
    ap.add_argument("--sim-world", type=int, default=0, help="(diagnostic) render only rank --sim-rank's tiles of this many, no collective")
    ap.add_argument("--sim-rank", type=int, default=0)
    ap.add_argument("--no-projection", action="store_true", help="skip the `projected` block (N = 1 only): rank 0's share of a 2-, 4- and 8-way tile split rendered alone on this GPU")
    ap.add_argument("--repeats", type=int, default=5, help="the timed region (exactly --steps steps between two barriers) is run this many times back to back; "
                    "`value` is the median repeat, all of them are in `repeats`")
    args = ap.parse_args()
    args.repeats = max(1, args.repeats)
    max_bounces = args.max_bounces or CONFIGS[args.config][1]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))       # `python3 bench.py --gpus N` as typed: this process only starts the N ranks

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    if os.environ.get("ER_BENCH_DRY_RUN") == "1":
        sys.exit(dry_run(rank, world))

    import torch   # device sync, torch.distributed (RCCL) -- plumbing only
    import numpy as np
    from elevenrender_amd import abi, render, scenes

    dist = None
    # ER_BENCH_REHEARSAL=1: several ranks share GPU 0 and talk over gloo (RCCL refuses two ranks on one GPU); used to
    # rehearse the N>1 control flow on a one-GPU box.  The driver's real runs use one GPU per rank and RCCL.
    rehearsal = os.environ.get("ER_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    coll_dev = "cpu" if rehearsal else "cuda"
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    scene, ext_flags, workload = make_scene(args, scenes, abi)
    shard_rank, shard_world = (args.sim_rank, args.sim_world) if (args.sim_world > 1 and world == 1) else (rank, world)
    sched_flag = {"auto": 0, "wavefront": abi.FLAG_WAVEFRONT, "megakernel": abi.FLAG_MEGAKERNEL, "stream": abi.FLAG_STREAM}[args.schedule]
    base_flags = sched_flag | ext_flags | (abi.FLAG_GPU_BUILD if args.gpu_build else 0) | (abi.FLAG_HOST_BUILD if args.host_build else 0)

    def manager(extra_flags):
        rm_ = render.RenderingManager(render.RenderParameters(sampleTarget=CONFIGS[args.config][2], max_bounces=max_bounces, device=f"hip:{local_rank}",
                                                              rank=shard_rank, world=shard_world, flags=base_flags | extra_flags))
        rm_.start_rendering(scene)
        return rm_

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def timed_region():
        """warm-up + the timed region.  Library errors are caught around the library calls only, so that every rank still
        reaches every barrier; the ranks then agree on whether the region succeeded."""
        # (ER_BENCH_NO_PROFILE=1, diagnostic: time the region without the per-launch HIP events of ER_FLAG_PROFILE)
        out = {"ok": 1, "error": None}
        try:
            out["rm"] = manager(0 if os.environ.get("ER_BENCH_NO_PROFILE") == "1" else abi.FLAG_PROFILE)
            out["accel"] = out["rm"].accel_info()
            out["warmup_ms"] = None
            if args.warmup > 0:
                out["rm"].render(args.warmup, blocking=False)
                out["warmup_ms"] = out["rm"].wait()       # device time of the warm-up call (one launch in the single-kernel schedules)
            out["c_before"] = out["rm"].counters()
        except abi.ErError as e:
            out["ok"], out["error"] = 0, str(e)
        # The timed region, `--repeats` times back to back: each repeat is EXACTLY `--steps` steps bracketed by barrier + synchronize on
        # both sides (one launch each in the single-kernel schedules), with its own counters and device time.  One 100-ms launch in a
        # 30-s job is a thin measurement: `value` is the median repeat and the line carries all of them.
        out["reps"] = []
        for _ in range(args.repeats):
            rep = {}
            if out["ok"]:
                try:
                    rep["c_before"] = out["rm"].counters()
                except abi.ErError as e:
                    out["ok"], out["error"] = 0, str(e)
            sync_all()
            rep["t0"] = time.perf_counter()
            if out["ok"]:
                try:
                    if args.per_step_launch:
                        for _ in range(args.steps):
                            out["rm"].render(1, blocking=False)
                    else:
                        out["rm"].render(args.steps, blocking=False)
                    rep["kernel_ms"] = out["rm"].wait()
                    rep["prof"] = out["rm"].profile()   # per-kernel device time of the timed region (HIP events on the streams the launches ran on)
                    if os.environ.get("ER_BENCH_SIMULATE_STREAM_FAILURE") == "1" and sched_flag != abi.FLAG_WAVEFRONT:
                        raise abi.ErError(-4, "simulated failure of the timed region (ER_BENCH_SIMULATE_STREAM_FAILURE)")   # tests the fallback below
                except abi.ErError as e:
                    out["ok"], out["error"] = 0, str(e)
            sync_all()
            rep["elapsed"] = time.perf_counter() - rep["t0"]
            if out["ok"]:
                rep["c_after"] = out["rm"].counters()
            out["reps"].append(rep)
        ok = out["ok"]
        if dist is not None:
            t = torch.tensor([ok], dtype=torch.int32, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        out["ok_all"] = ok
        return out

    region = timed_region()
    schedule_fallback = None
    if not region["ok_all"] and sched_flag in (0, abi.FLAG_STREAM):
        # The streaming schedule ends a call with an error instead of hanging if a workgroup stops making progress (its
        # watchdog).  The bench then measures the wavefront schedule and says so, rather than reporting nothing.
        print(f"rank {rank}: timed region failed ({region['error']}); repeating it with the wavefront schedule", file=sys.stderr)
        schedule_fallback = region["error"] or "another rank failed"
        if region.get("rm") is not None:
            region["rm"].close()
        base_flags = abi.FLAG_WAVEFRONT | ext_flags | (abi.FLAG_GPU_BUILD if args.gpu_build else 0) | (abi.FLAG_HOST_BUILD if args.host_build else 0)
        sched_flag = abi.FLAG_WAVEFRONT
        region = timed_region()
    if not region["ok_all"]:
        raise RuntimeError(f"timed region failed: {region['error']}")
    rm, accel = region["rm"], region["accel"]
    launches = args.steps if args.per_step_launch else 1
    reps = region["reps"]
    K = len(reps)
    keys = ("bounce_samples", "paths", "rays")
    mine_counts = [[r["c_after"][k] - r["c_before"][k] for k in keys] for r in reps]          # this rank, per repeat
    rep_elapsed = [r["elapsed"] for r in reps]
    rep_totals = [list(m) for m in mine_counts]
    every = None
    if dist is not None:
        # what every rank did, so that a scaling run explains itself: device time of its timed region, its samples and rays -- per repeat
        mine = torch.tensor([[r["kernel_ms"]] + [float(v) for v in m] for r, m in zip(reps, mine_counts)], dtype=torch.float64, device=coll_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = [e.tolist() for e in every]                      # [rank][repeat][kernel_ms, samples, paths, rays]
        t = torch.tensor(rep_elapsed, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # a repeat lasts as long as its slowest rank
        rep_elapsed = t.tolist()
        tot = torch.tensor(mine_counts, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        rep_totals = [[int(v) for v in row] for row in tot.tolist()]
    rep_values = [rep_totals[i][0] / rep_elapsed[i] / 1e6 for i in range(K)]
    order = sorted(range(K), key=lambda i: rep_values[i])
    mid = order[(K - 1) // 2]                                    # the median repeat (the lower middle one of an even count): a repeat that ran, not an average
    c_before, c_after = reps[mid]["c_before"], reps[mid]["c_after"]
    kernel_ms, prof, elapsed = reps[mid]["kernel_ms"], reps[mid]["prof"], rep_elapsed[mid]
    samples, paths, rays = rep_totals[mid]
    per_rank = None
    if every is not None:
        em = [e[mid] for e in every]
        per_rank = {"kernel_ms": [round(e[0], 3) for e in em],
                    "ms_per_step_min": round(min(e[0] for e in em) / args.steps, 4), "ms_per_step_max": round(max(e[0] for e in em) / args.steps, 4),
                    "bounce_samples": [int(e[1]) for e in em], "paths": [int(e[2]) for e in em], "rays": [int(e[3]) for e in em]}

    # ---- framebuffer combine (once per read-back, outside the timed region) ----
    # It runs on a worker thread with a deadline: RCCL from the library's C++ side has never run with more than one rank on the
    # machines this was built on, and a communicator that never forms must not cost the run its measurement -- after
    # ER_BENCH_GATHER_TIMEOUT seconds (180) rank 0 prints the line with `gather` saying so and the process exits with code 5.
    t_rb = time.perf_counter()
    gstate = {"path": None, "ms": None, "error": None}

    def do_gather():
        from elevenrender_amd import dist as erdist
        try:
            if rehearsal:
                erdist.gather_all_planes_torch(dist, rm, rank, world)     # gloo on the host: the test harness path
                gstate["path"] = "torch.distributed (gloo rehearsal)"
                return
            # the production combine: RCCL communicator made by the library's C++ side (er_comm_create), then per plane
            # er_gather_pass = pack -> ncclSend / ncclRecv over xGMI -> unpack.  If the C++ path cannot start on this
            # node (RCCL not loadable ...) every rank falls back to the torch.distributed gather so that the line is
            # not lost; `gather` says which path ran.
            comm = None
            try:
                comm = erdist.NativeComm(dist, rank, world, local_rank)   # (the ranks agree inside before anything collective)
            except Exception as e:
                print(f"[rank {rank}] native RCCL communicator not available: {e}", file=sys.stderr)
            if comm is not None:
                sync_all()
                t_g = time.perf_counter()
                for p in range(abi.PASS_COUNT):
                    comm.gather_pass(rm, p)
                sync_all()
                gstate["ms"] = (time.perf_counter() - t_g) * 1e3      # pack + send/recv over xGMI + unpack of all five planes, slowest rank
                gstate["path"] = "er_gather_pass (RCCL from the C++ side, 5 planes)"
                comm.close()
            else:
                erdist.gather_all_planes_torch(dist, rm, rank, world)
                gstate["path"] = "torch.distributed.gather (fallback: er_comm_create failed)"
            torch.cuda.synchronize()
        except Exception as e:      # (reported in the line; the throughput measurement above stands)
            gstate["error"] = f"{type(e).__name__}: {e}"

    gather_hung = False
    if dist is not None:
        import threading
        th = threading.Thread(target=do_gather, daemon=True)
        th.start()
        th.join(float(os.environ.get("ER_BENCH_GATHER_TIMEOUT", "180")))
        gather_hung = th.is_alive()
        if gather_hung:
            gstate["path"] = "NOT COMPLETED: the framebuffer gather did not return within its deadline (the throughput figures above do not depend on it)"
        elif gstate["error"]:
            gstate["path"] = "FAILED: " + gstate["error"]
    gather_path, gather_ms = gstate["path"], gstate["ms"]
    gather_ok = dist is None or (not gather_hung and not gstate["error"])
    beauty_mean = None
    if rank == 0 and gather_ok:
        img = rm.get_pass("beauty")
        assert np.isfinite(img).all()
        beauty_mean = float(img[..., :3].mean())
    readback_ms = (time.perf_counter() - t_rb) * 1e3

    result = None
    if rank == 0:
        # ---- roofline: algorithmic bytes from an instrumented replay of the same samples ----
        inst = manager(abi.FLAG_COUNTERS)
        n_inst = min(2, args.steps)
        inst.render(n_inst)
        ci = inst.counters()
        inst.close()
        hdri_texels = scene.hdri[1] * scene.hdri[2]
        my_samples = c_after["bounce_samples"] - c_before["bounce_samples"]
        my_rays = c_after["rays"] - c_before["rays"]
        # per-ray statistics of the instrumented replay scale the timed region's ray count
        trace_b = trace_bytes(ci) / max(1, ci["rays"]) * my_rays
        layout_b = trace_bytes_layout(ci) / max(1, ci["rays"]) * my_rays
        path_b = path_bytes(ci, hdri_texels) / max(1, ci["bounce_samples"]) * my_samples
        t_launches = max(1, prof["trace_launches"])
        trace_ms_avg = prof["trace_ms"] / t_launches
        sched = {abi.FLAG_WAVEFRONT: "wavefront", abi.FLAG_MEGAKERNEL: "megakernel", abi.FLAG_STREAM: "stream"}.get(prof["schedule"], "?")
        kernel_name = {"wavefront": "er_wf_trace", "megakernel": "er_render_kernel", "stream": "er_stream_kernel"}.get(sched, "?")
        if sched not in ("wavefront", "stream"):
            layout_b = path_b + (layout_b - trace_b)
            trace_b = path_b      # the single kernel of these schedules does the whole path
        wall_s = max(kernel_ms * 1e-3, 1e-9)        # device time of the whole timed region (HIP events on the library's stream)
        achieved = trace_b / wall_s / 1e9           # ALL traversal bytes of the region / its wall time -- no concurrency factor
        per_launch = (trace_b / t_launches) / (trace_ms_avg * 1e-3) / 1e9 if trace_ms_avg > 0 else 0.0

        # ---- trace phase alone: the same steps with ONE slot pool on ONE stream (launch durations are then exclusive) ----
        trace_phase = None
        if sched == "wavefront" and not args.no_trace_phase:
            os.environ["ER_WF_POOLS"] = "1"
            try:
                one = manager(abi.FLAG_PROFILE | (0 if sched_flag else abi.FLAG_WAVEFRONT))
            finally:
                del os.environ["ER_WF_POOLS"]
            if args.warmup > 0:
                one.render(args.warmup)
            c0 = one.counters()
            torch.cuda.synchronize()
            one.render(args.steps, blocking=False)
            one_ms = one.wait()
            p1 = one.profile()
            c1 = one.counters()
            one.close()
            rays1 = c1["rays"] - c0["rays"]
            b1 = trace_bytes(ci) / max(1, ci["rays"]) * rays1
            tp = b1 / max(p1["trace_ms"] * 1e-3, 1e-9) / 1e9
            trace_phase = {"schedule": "wavefront, 1 slot pool, 1 stream (launches do not overlap: bytes / summed trace-launch time is exact)",
                           "achieved": round(tp, 2), "frac": round(tp / HBM_PEAK_GBS, 4),
                           "trace_ms_total": round(p1["trace_ms"], 3), "shade_ms_total": round(p1["shade_ms"], 3), "region_ms": round(one_ms, 3),
                           "trace_launches": p1["trace_launches"], "empty_launches": p1["empty_launches"],
                           "Msamples_per_s": round((c1["bounce_samples"] - c0["bounce_samples"]) / max(one_ms * 1e-3, 1e-9) / 1e6, 2)}

        # ---- measured HBM peak, same job ----
        try:
            copy_gbs, read_gbs = render.measure_hbm_peak(local_rank)
        except Exception as e:     # measurement helper only; the line stays valid without it
            copy_gbs, read_gbs = None, None
            print(f"er_measure_hbm_peak failed: {e}", file=sys.stderr)
        peak_measured = max(copy_gbs or 0.0, read_gbs or 0.0) or None

        traffic = None
        traffic_source = None
        # HBM-side bytes from rocprofv3 PMC passes of THIS command line (tools/pmc_passes.sh + tools/pmc_traffic.py; counters cannot be
        # collected inside a timed run): replayed, and only for the exact workload they were recorded on
        standard = args.tris == 1_000_000 and args.blob_tris == 1000 and not (args.width or args.height) and not args.max_bounces and world == 1 and shard_world == 1
        cfg_key = args.config + ("_nolights" if args.config == "C5" and args.no_lights else "")
        if sched == "stream" and standard:
            for tag in (PROFILE_TAG, "r04"):
                tf = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
                if not os.path.exists(tf):
                    continue
                try:
                    tj = json.load(open(tf))
                    per_step = (tj.get(cfg_key) or {}).get("er_stream_kernel_hbm_bytes_per_step") or (tj.get("er_stream_kernel_hbm_bytes_per_step") if cfg_key == "C2" else None)
                    if per_step:
                        # one launch of the streaming kernel runs all K steps of the call; the PMC passes measured bytes per step
                        traffic = per_step * args.steps / launches
                        traffic_source = (f"profiles/{os.path.basename(tf)} [{cfg_key}]: bytes per step from PMC passes of an earlier run of this command x the "
                                          f"steps of this launch, replayed (not measured in this run)")
                        break
                except Exception:
                    traffic = None
        elif sched == "wavefront" and standard and args.config == "C2":
            tf = os.path.join(ROOT, "profiles", "r02wf_pmc_traffic.json")
            if os.path.exists(tf):
                traffic = json.load(open(tf)).get("er_wf_trace_hbm_bytes_per_launch")
                traffic_source = f"profiles/{os.path.basename(tf)}: PMC passes of an earlier run of this command, replayed (not measured in this run)"
        value = samples / elapsed / 1e6
        result = {
            "metric": "Msamples/sec (rays traced x bounces) at 1920x1080, 1M-tri scene",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # true = the default schedule's timed region ended in an error (its watchdog) and `value` was measured with the
            # wavefront schedule instead: a correctness event of the product's default path; the process then exits with code 3
            "degraded": schedule_fallback is not None,
            "config": {"workload": f"{workload}, max_bounces {max_bounces}, 1 step = 1 spp pass (config total {CONFIGS[args.config][2]} spp), seed 12345",
                       "sharding": f"8x8 pixel tiles, (tx+ty) % {shard_world}", "calls_in_timed_region": launches,
                       "schedule": sched, "schedule_fallback": schedule_fallback},
            "paths_per_s": round(paths / elapsed, 1), "rays_per_s": round(rays / elapsed, 1),
            "mean_path_length": round(samples / max(1, paths), 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         # the same counter bytes as a rate over the timed region: what the L2s asked of the fabric (Infinity-Cache hits
                         # are counted, MI355X_MICROARCH.md), against the spec peak and the peak measured in this job
                         # (wavefront: `traffic` is per trace launch of one pool -- the shade launches' bytes are not in it)
                         "traffic_GBps": round(traffic * (t_launches if sched == "wavefront" else launches) / wall_s / 1e9, 1) if traffic else None,
                         "traffic_frac": round(traffic * (t_launches if sched == "wavefront" else launches) / wall_s / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                         # "L2/MALL-resident traffic must be reported separately from HBM".  What the traversal touches -- wide nodes +
                         # 48-byte triangle records -- against the 256 MB Infinity Cache: a working set inside it is served by the L2s and that
                         # cache (FETCH_SIZE, and so `traffic`, counts its hits: MI355X_MICROARCH.md), and "bound: hbm" then names the roofline the
                         # metric is defined against, not the unit that limits the kernel (the tracer waves' instruction
                         # streams; profiles/r06_c4_working_set_sweep.log: 57 MB -> 1 147 MB of the same frame costs 19 % of the rate)
                         "resident": {"traversal_working_set_MB": round((accel["node_count"] * accel["node_bytes"] + 48.0 * scene.tri_count) / 1e6, 1),
                                      "infinity_cache_MB": 256,
                                      "where": "inside the Infinity Cache" if accel["node_count"] * accel["node_bytes"] + 48.0 * scene.tri_count <= 256e6 else "beyond the Infinity Cache"},
                         "kernel": kernel_name,
                         "definition": "achieved = all traversal bytes of the timed region (64 B/node visit + 36 B/triangle test, SURVEY 8d) / device time of the region"
                                       + ("; the streaming schedule is ONE kernel (traversal and shading waves side by side): region time = its launch duration, nothing overlaps it" if sched == "stream" else ""),
                         "peak_measured": round(peak_measured, 1) if peak_measured else None,
                         "peak_measured_copy": round(copy_gbs, 1) if copy_gbs else None, "peak_measured_read": round(read_gbs, 1) if read_gbs else None,
                         "frac_of_measured": round(achieved / peak_measured, 4) if peak_measured else None,
                         "trace_phase": trace_phase,
                         "launches": prof["trace_launches"], "empty_launches": prof["empty_launches"], "empty_ms": round(prof["empty_ms"], 3),
                         "pools_side_by_side": max(1, int(prof.get("concurrency", 1))),
                         "achieved_per_launch": round(per_launch, 2),
                         "achieved_own_layout": round(achieved * layout_b / max(trace_b, 1.0), 2),
                         "avg_launch_ms": round(trace_ms_avg, 5), "algorithmic_bytes_per_launch": round(trace_b / t_launches, 1),
                         # rocprofv3 --stats also sees the warm-up call: for the one-kernel-per-call schedules its row is
                         # (warm-up launch, timed launch), i.e. MaxNs = avg_launch_ms and AverageNs = their mean
                         "warmup_launch_ms": round(region["warmup_ms"], 3) if region.get("warmup_ms") is not None else None,
                         "trace_ms_total": round(prof["trace_ms"], 3), "shade_ms_total": round(prof["shade_ms"], 3),
                         "region_ms": round(kernel_ms, 3),
                         # SURVEY 8(d)'s FULL per-pixel-sample formula (64 V + 36 T + 112 H + 12 X + 4 ceil(log2 P) S + 112) over the same region: the
                         # streaming kernel runs the whole path, so this is its algorithmic rate by the survey's definition; `achieved` / `frac` keep
                         # to the traversal terms alone (north_star: "during BVH traversal"), as in every earlier round
                         "whole_path_GBps": round(path_b / wall_s / 1e9, 2), "frac_whole_path": round(path_b / wall_s / 1e9 / HBM_PEAK_GBS, 4),
                         "node_visits_per_ray": round(ci["node_visits"] / max(1, ci["rays"]), 2),
                         "tri_tests_per_ray": round(ci["tri_tests"] / max(1, ci["rays"]), 2),
                         # lane occupancy of er_wf_trace's loop in the instrumented replay: share of the 64 lanes that held a
                         # ray / ran the node part / ran the triangle part, per loop iteration of a wave
                         "trace_lanes": None if not ci.get("trace_wave_steps") else {
                             "wave_steps": ci["trace_wave_steps"],
                             "busy": round(ci["trace_busy_lanes"] / (64.0 * ci["trace_wave_steps"]), 4),
                             "node": round(ci["trace_node_lanes"] / (64.0 * ci["trace_wave_steps"]), 4),
                             "tri": round(ci["trace_tri_lanes"] / (64.0 * ci["trace_wave_steps"]), 4)}},
            "accel": {"nodes": accel["node_count"], "node_bytes": accel["node_bytes"], "leaves": accel["leaf_count"],
                      "max_depth": accel["max_depth"], "build_ms": round(accel["build_ms"], 1), "builder": "device binned SAH" if accel["builder"] else "host binned SAH", "upload_ms": round(accel["upload_ms"], 2)},
            # every repeat of the timed region (each exactly `steps` steps between two barriers): `value` is the median one
            "repeats": {"k": K, "values": [round(v, 3) for v in rep_values], "median_index": mid, "min": round(min(rep_values), 3), "max": round(max(rep_values), 3),
                        "spread": round((max(rep_values) - min(rep_values)) / max(rep_values[mid], 1e-12), 5),
                        "region_ms": [round(r["kernel_ms"], 3) for r in reps], "elapsed_ms": [round(e * 1e3, 3) for e in rep_elapsed]},
            "readback_ms": round(readback_ms, 2), "gather": gather_path, "beauty_mean": beauty_mean,
            # the streaming schedule's configuration of this rank's share and its speculation counts over the render (zero on whole frames:
            # speculative samples exist in the kernel's form 2 only, i.e. on shares of at most 2 304 pixels per CU)
            "stream": (lambda si: {"pixels_per_cu": si["pixels_per_cu"], "form": si["form"], "waves": si["waves"], "tracers": si["tracers"], "large_regions": bool(si["large_regions"]), "lanes_busy": round(si["lanes_busy"], 4),
                                   "speculation": {"started": si["spec_started"], "right": si["spec_right"], "wrong": si["spec_wrong"]}})(rm.stream_info()) if sched == "stream" else None,
            # N > 1: per-rank device time of the timed region (a rank whose tiles hold longer paths shows here), and the framebuffer
            # combine: wall time of the five er_gather_pass calls on the slowest rank and the bytes the root received
            "ranks": per_rank,
            "gather_ms": round(gather_ms, 3) if gather_ms is not None else None,
            "gather_bytes": (int(scene.x_res) * int(scene.y_res) * 16 * 5 * (world - 1) // world) if world > 1 else None,
        }
        if world == 1 and shard_world == 1 and not args.no_projection:
            # What the first SCALE record can be read against: rank 0's share of an N-way split of the SAME frame ((tx + ty) % N tiles),
            # rendered alone on this GPU with the same steps and warm-up -- every rank of a real N-GPU run does this much work side by side
            # (the soup's shares are alike; a frame of uneven cost has a slowest rank, which this does not show), then one gather.
            proj = {}
            for n in (2, 4, 8):
                try:
                    pm = render.RenderingManager(render.RenderParameters(sampleTarget=CONFIGS[args.config][2], max_bounces=max_bounces, device=f"hip:{local_rank}",
                                                                         rank=0, world=n, flags=base_flags))
                    pm.start_rendering(scene)
                    if args.warmup > 0:
                        pm.render(args.warmup)
                    ms = []
                    for _ in range(3):
                        pm.render(args.steps, blocking=False)
                        ms.append(pm.wait())
                    si = pm.stream_info()
                    pm.close()
                    share_ms = sorted(ms)[1] / args.steps
                    proj[str(n)] = {"share_ms_per_step": round(share_ms, 4), "speedup_vs_1": round((kernel_ms / args.steps) / share_ms, 3),
                                    "value": round(value * (kernel_ms / args.steps) / share_ms, 1),
                                    # what the share ran as: owned pixels per CU, waves per workgroup and how many of them trace, the share of their
                                    # lanes that held a ray in the last launch (the whole frame: roofline.trace_lanes.busy of the instrumented replay),
                                    # the deal; `latency_tracer`: no second tracer role exists (priced and not built)
                                    # `form`: 0 the whole-frame kernel, 1 pixels that are behind keep their slots, 2 that and speculative samples
                                    "pixels_per_cu": si["pixels_per_cu"], "form": si["form"], "waves": si["waves"], "tracers": si["tracers"], "lanes_busy": round(si["lanes_busy"], 4),
                                    "large_regions": bool(si["large_regions"]), "latency_tracer": False,
                                    # round 6: speculative sample pipelining of the kernel's form 2 (DESIGN.md sections 5 and 7): samples started beside the
                                    # pixel's sample in flight, and how many of those guesses of the RNG state were right / wrong
                                    "speculation": {"started": si["spec_started"], "right": si["spec_right"], "wrong": si["spec_wrong"]}}
                except abi.ErError as e:
                    proj[str(n)] = {"error": str(e)}
            result["projected"] = {"what": "one GPU's share (rank 0 of N) of this frame rendered alone on this GPU, median of 3 launches of `steps` steps; "
                                           "speedup = whole-frame device time / share device time; excludes the framebuffer gather (one per read-back, outside the timed region)",
                                   "n_gpus": proj}
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or host_cores()
            # C4: ONE reference-style BVH build of the 10 M triangles (the expensive part, not timed), then the same bounded sample
            # of rows on all cores; the one-core leg would need a second build and is left out there
            result["cpu_baseline"] = cpu_baseline(scene, max_bounces, ext_flags, threads, args.cpu_budget, one_core=args.config != "C4")
            result["gpu_over_cpu"] = round(value / max(result["cpu_baseline"]["value"], 1e-12), 1)
    if not gather_ok:
        # a rank is (or may be) stuck inside a collective: say what was measured and leave without touching the process group again
        if rank == 0:
            print(json.dumps(result), flush=True)
        else:
            time.sleep(float(os.environ.get("ER_BENCH_EXIT_GRACE", "45")))      # (a launcher ends every rank when one leaves with an error: rank 0 prints its line first)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(5)
    rm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if schedule_fallback is not None:
        sys.exit(3)      # the line above is a valid measurement of the fallback schedule, but the default one failed: not a success


if __name__ == "__main__":
    main()
