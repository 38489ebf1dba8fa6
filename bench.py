#!/usr/bin/env python3
"""bench.py -- Msamples/s of the per-sample path-tracing hot path on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d "C2"): synthetic 1M random-triangle soup +
2048x1024 procedural sky HDRI, 1920x1080, max_bounces 8, seed 12345.  A STEP is one sample
pass (one renderingKernel per pixel, reference src/kernel.cpp:689-700); the 256 spp of the
config are 256 such steps.  `value` = executed bounce-loop iterations (the reference's
`for (i...)` at src/kernel.cpp:508, counted on the device) per second, whole job, scene and
framebuffers resident in HBM when the timed region starts.

With --gpus N the frame's 8x8 pixel tiles are sharded over N ranks (one process per GPU,
launched by torch.distributed.run); there is no collective in the timed region -- the RCCL
framebuffer gather happens once after it and is reported as readback_ms.

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


def cpu_baseline(scene, max_bounces, budget_s=18.0):
    """The oracle (a port of the reference algorithm: fixed-depth-18 BVH, unordered unpruned
    traversal) timed on this box's host cores, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    cores = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16))   # the GPU box's CPU share for one GPU is 16
    o = oracle.Oracle(scene, math_mode=oracle.MATH_LIBM, max_bounces=max_bounces, threads=cores)
    W, H = scene.x_res, scene.y_res

    def run_rows(rows):
        t0 = time.perf_counter()
        c0 = o.counters()["bounce_samples"]
        for y in rows:
            o.render(1, y * W, (y + 1) * W)
        return time.perf_counter() - t0, o.counters()["bounce_samples"] - c0

    probe_rows = list(range(H // 16, H, H // 8))[:8]          # 8 rows spread over the frame
    t_probe, s_probe = run_rows(probe_rows)
    per_row = t_probe / len(probe_rows)
    n_rows = int(max(8, min(H - len(probe_rows), budget_s / max(per_row, 1e-6))))
    stride = max(1, H // n_rows)
    rows = [y for y in range(stride // 2, H, stride) if y not in probe_rows][:n_rows]
    t_main, s_main = run_rows(rows)
    total_rows = len(probe_rows) + len(rows)
    value = (s_probe + s_main) / (t_probe + t_main) / 1e6
    build_s = o.build_seconds
    o.close()
    return {"value": round(value, 6), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{total_rows} rows x {W} px of the same 1M-tri frame at 1 spp (rows spread over the image), "
                      f"{t_probe + t_main:.1f} s of work on {cores} threads; reference-BVH build {build_s:.1f} s not timed"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--max-bounces", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=18.0)
    ap.add_argument("--per-step-launch", action="store_true", help="one launch per step instead of one launch for all K steps")
    ap.add_argument("--schedule", choices=["auto", "wavefront", "fused", "megakernel"], default="auto")
    ap.add_argument("--gpu-build", action="store_true", help="build the BVH on the GPU (ER_FLAG_GPU_BUILD) instead of the host SAH build")
    ap.add_argument("--sim-world", type=int, default=0, help="(diagnostic) render only rank --sim-rank's tiles of this many, no collective")
    ap.add_argument("--sim-rank", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

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

    scene = scenes.soup(args.tris, args.width, args.height, seed=12345)
    shard_rank, shard_world = (args.sim_rank, args.sim_world) if (args.sim_world > 1 and world == 1) else (rank, world)
    sched_flag = {"auto": 0, "wavefront": abi.FLAG_WAVEFRONT, "fused": abi.FLAG_FUSED, "megakernel": abi.FLAG_MEGAKERNEL}[args.schedule]
    pars = render.RenderParameters(sampleTarget=256, max_bounces=args.max_bounces, device=f"hip:{local_rank}",
                                   rank=shard_rank, world=shard_world,
                                   flags=abi.FLAG_PROFILE | sched_flag | (abi.FLAG_GPU_BUILD if args.gpu_build else 0))
    rm = render.RenderingManager(pars)
    rm.start_rendering(scene)
    accel = rm.accel_info()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        rm.render(args.warmup)
    c_before = rm.counters()
    sync_all()
    t0 = time.perf_counter()
    if args.per_step_launch:
        for _ in range(args.steps):
            rm.render(1, blocking=False)
    else:
        rm.render(args.steps, blocking=False)
    kernel_ms = rm.wait()
    prof = rm.profile()      # per-kernel device time of the timed region (HIP events on the library's stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    c_after = rm.counters()
    launches = args.steps if args.per_step_launch else 1

    samples = c_after["bounce_samples"] - c_before["bounce_samples"]
    paths = c_after["paths"] - c_before["paths"]
    rays = c_after["rays"] - c_before["rays"]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([samples, paths, rays], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        samples, paths, rays = (int(v) for v in tot.tolist())

    # ---- framebuffer combine (once per read-back, outside the timed region) ----
    t_rb = time.perf_counter()
    if dist is not None:
        from elevenrender_amd import dist as erdist
        rows = [rm.owned_count(r) for r in range(world)]
        mine = torch.zeros((rows[rank], 4), dtype=torch.float32, device="cuda")
        rm.pack_owned(abi.PASS_BEAUTY, mine.data_ptr())

        def unpack(r, t):
            t = t.contiguous().cuda()
            torch.cuda.synchronize()
            rm.unpack_owned(abi.PASS_BEAUTY, r, t.data_ptr())
        erdist.gather_plane(dist, rank, world, mine.to(coll_dev), max(rows), unpack)
        torch.cuda.synchronize()
    beauty_mean = None
    if rank == 0:
        img = rm.get_pass("beauty")
        assert np.isfinite(img).all()
        beauty_mean = float(img[..., :3].mean())
    readback_ms = (time.perf_counter() - t_rb) * 1e3

    result = None
    if rank == 0:
        # ---- roofline: algorithmic bytes per launch from an instrumented replay of the same samples ----
        inst = render.RenderingManager(render.RenderParameters(sampleTarget=256, max_bounces=args.max_bounces,
                                                               device=f"hip:{local_rank}", rank=shard_rank, world=shard_world,
                                                               flags=abi.FLAG_COUNTERS | sched_flag | (abi.FLAG_GPU_BUILD if args.gpu_build else 0)))
        inst.start_rendering(scene)
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
        sched = {abi.FLAG_WAVEFRONT: "wavefront", abi.FLAG_FUSED: "fused", abi.FLAG_MEGAKERNEL: "megakernel"}.get(prof["schedule"], "?")
        kernel_name = {"wavefront": "er_wf_trace", "fused": "er_fused_kernel", "megakernel": "er_render_kernel"}.get(sched, "?")
        if sched != "wavefront":
            trace_b = path_b      # the single kernel of these schedules does the whole path
            layout_b = path_b + (layout_b - trace_bytes(ci) / max(1, ci["rays"]) * my_rays)
        # the wavefront schedule runs `conc` slot pools side by side, each on its own stream: launches overlap, so
        # the rate the kernel sustains is conc x (bytes of one launch / duration of one launch).  Conservative: a
        # trace launch also shares the chip with the other pools' shade launches for part of its duration.
        conc = max(1, int(prof.get("concurrency", 1)))
        per_launch = (trace_b / t_launches) / (trace_ms_avg * 1e-3) / 1e9 if trace_ms_avg > 0 else 0.0
        achieved = per_launch * conc
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # HBM bytes per launch from rocprofv3 PMC passes
        if os.path.exists(tf) and world == 1 and args.tris == 1_000_000:
            try:
                traffic = json.load(open(tf)).get("er_wf_trace_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        value = samples / elapsed / 1e6
        result = {
            "metric": "Msamples/sec (rays traced x bounces) at 1920x1080, 1M-tri scene",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2: {args.tris}-triangle random soup + 2048x1024 sky HDRI, {args.width}x{args.height}, "
                                   f"max_bounces {args.max_bounces}, 1 step = 1 spp pass (config total 256 spp), seed 12345",
                       "sharding": f"8x8 pixel tiles, (tx+ty) % {shard_world}", "calls_in_timed_region": launches,
                       "schedule": sched},
            "paths_per_s": round(paths / elapsed, 1), "rays_per_s": round(rays / elapsed, 1),
            "mean_path_length": round(samples / max(1, paths), 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": kernel_name, "launches": prof["trace_launches"], "concurrent_launches": conc,
                         "achieved_per_launch": round(per_launch, 2),
                         "achieved_own_layout": round(per_launch * conc * layout_b / max(trace_b, 1.0), 2),
                         "avg_launch_ms": round(trace_ms_avg, 5), "algorithmic_bytes_per_launch": round(trace_b / t_launches, 1),
                         "trace_ms_total": round(prof["trace_ms"], 3), "shade_ms_total": round(prof["shade_ms"], 3),
                         "whole_path_GBps": round(path_b / (kernel_ms * 1e-3) / 1e9, 2) if kernel_ms > 0 else None,
                         "node_visits_per_ray": round(ci["node_visits"] / max(1, ci["rays"]), 2),
                         "tri_tests_per_ray": round(ci["tri_tests"] / max(1, ci["rays"]), 2)},
            "accel": {"nodes": accel["node_count"], "node_bytes": accel["node_bytes"], "leaves": accel["leaf_count"],
                      "max_depth": accel["max_depth"], "build_ms": round(accel["build_ms"], 1), "builder": "device linear BVH" if accel["builder"] else "host binned SAH", "upload_ms": round(accel["upload_ms"], 2)},
            "readback_ms": round(readback_ms, 2), "beauty_mean": beauty_mean,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(scene, args.max_bounces, args.cpu_budget)
            result["gpu_over_cpu"] = round(value / max(result["cpu_baseline"]["value"], 1e-12), 1)
    rm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
