"""Multi-GPU host logic on CPU: tile ownership is a partition, and the framebuffer combine
(pack -> torch.distributed.gather -> unpack, the code bench.py runs over RCCL) reassembles the
full frame, exercised with the gloo backend at world_size 2 and 3."""
import os
import socket

import numpy as np
import pytest

from elevenrender_amd import dist as erdist


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("res", [(1920, 1080), (64, 64), (50, 37), (7, 5)])
def test_tile_ownership_is_a_partition(world, res):
    x_res, y_res = res
    seen = np.zeros(x_res * y_res, np.int32)
    counts = []
    for r in range(world):
        idx = erdist.tile_pixel_index(erdist.owned_tiles(r, world, x_res, y_res), x_res, y_res)
        valid = idx[idx >= 0]
        seen[valid] += 1
        counts.append(valid.size)
    assert (seen == 1).all()                       # every pixel owned exactly once
    if world > 1 and x_res >= 64:
        assert max(counts) - min(counts) <= 0.02 * x_res * y_res   # balanced within 2 %


def _worker(rank, world, port, x_res, y_res, out_q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # each rank "renders" only the pixels it owns: value = f(pixel), everything else stays at setup value
    full = np.zeros((y_res, x_res, 4), np.float32)
    full[..., 3] = 1.0
    truth = np.zeros_like(full)
    yy, xx = np.mgrid[0:y_res, 0:x_res]
    truth[..., 0] = xx * 0.5 + yy
    truth[..., 1] = (xx * yy) % 7
    truth[..., 2] = -xx
    truth[..., 3] = 1.0
    mine_idx = erdist.tile_pixel_index(erdist.owned_tiles(rank, world, x_res, y_res), x_res, y_res)
    v = mine_idx[mine_idx >= 0]
    full.reshape(-1, 4)[v] = truth.reshape(-1, 4)[v]
    compact = torch.from_numpy(erdist.pack_owned_host(full, rank, world))
    rows = [erdist.owned_tiles(r, world, x_res, y_res).size * 64 for r in range(world)]

    def unpack(src_rank, t):
        erdist.unpack_owned_host(full, t.numpy(), src_rank, world)
    erdist.gather_plane(dist, rank, world, compact, max(rows), unpack)
    ok = True
    if rank == 0:
        ok = bool((full == truth).all())
    t = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    out_q.put((rank, ok, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,res", [(2, (100, 60)), (3, (64, 40))])
def test_gloo_framebuffer_gather(world, res):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, res[0], res[1], q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and red == 1.0 for (_, ok, red) in results), results


def test_bench_gpus_n_starts_its_own_n_ranks():
    """`python3 bench.py --gpus 2` with no launcher around it must become TWO ranks (it used to render the whole
    frame on one GPU and print n_gpus 1).  ER_BENCH_DRY_RUN=1: the ranks join over gloo and rank 0 reports who is there --
    nothing touches a GPU, so this runs here.  Under a launcher (WORLD_SIZE set) the same file must NOT start ranks again."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["ER_BENCH_DRY_RUN"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3"], env=env, capture_output=True, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-1500:], p.stderr[-1500:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_joined"] == [0, 1]
    assert "torch.distributed.run" in p.stderr and "--nproc-per-node=2" in p.stderr
    # one rank, no launcher: no child process
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 1 and "torch.distributed.run" not in p.stderr
