"""Multi-GPU host logic: pixel-tile ownership and the framebuffer combine.

The path shards by independent pixels: rank r of `world` renders the 8x8 tiles
(tx, ty) with (tx + ty) % world == r -- the same rule the library applies (er_api.cpp,
ErScene::tiles_of).  No collective runs per sample; one gather per read-back moves each
rank's owned pixels to rank 0 (ownership is disjoint, so the "reduce" is a gather).
The collective itself goes through torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).
"""
import numpy as np

TILE = 8


def owned_tiles(rank, world, x_res, y_res):
    """tile ids (ty * tiles_x + tx), ascending -- identical to ErScene::tiles_of."""
    tiles_x, tiles_y = (x_res + TILE - 1) // TILE, (y_res + TILE - 1) // TILE
    ty, tx = np.divmod(np.arange(tiles_x * tiles_y, dtype=np.int64), tiles_x)
    return np.nonzero((tx + ty) % world == rank)[0].astype(np.int64)


def tile_pixel_index(tiles, x_res, y_res):
    """[n_tiles, 64] pixel indices (y * x_res + x) in lane order, -1 where the lane is outside the image."""
    tiles_x = (x_res + TILE - 1) // TILE
    ty, tx = np.divmod(np.asarray(tiles, np.int64), tiles_x)
    lane = np.arange(64, dtype=np.int64)
    px = tx[:, None] * TILE + (lane & 7)[None, :]
    py = ty[:, None] * TILE + (lane >> 3)[None, :]
    idx = py * x_res + px
    idx[(px >= x_res) | (py >= y_res)] = -1
    return idx


def pack_owned_host(plane, rank, world):
    """numpy twin of er_pack_kernel: plane [y,x,4] -> [n_owned_tiles*64, 4] (zeros outside the image)."""
    y_res, x_res = plane.shape[:2]
    idx = tile_pixel_index(owned_tiles(rank, world, x_res, y_res), x_res, y_res).reshape(-1)
    out = np.zeros((idx.size, 4), plane.dtype)
    ok = idx >= 0
    out[ok] = plane.reshape(-1, 4)[idx[ok]]
    return out


def unpack_owned_host(plane, compact, src_rank, world):
    """numpy twin of er_unpack_kernel (in place)."""
    y_res, x_res = plane.shape[:2]
    idx = tile_pixel_index(owned_tiles(src_rank, world, x_res, y_res), x_res, y_res).reshape(-1)
    ok = idx >= 0
    plane.reshape(-1, 4)[idx[ok]] = compact[: idx.size][ok]


def gather_plane(dist, rank, world, my_compact, max_rows, unpack):
    """Gather every rank's compact owned-pixel buffer to rank 0 and scatter them into rank 0's plane.

    my_compact: torch tensor [rows_r, 4] on the collective's device; max_rows: max over ranks of rows
    (buffers are padded to it -- torch.distributed.gather needs equal shapes); unpack(src_rank, tensor)
    writes one rank's pixels into the full plane.  Returns nothing; rank 0's plane is complete afterwards.
    """
    import torch
    buf = torch.zeros((max_rows, 4), dtype=my_compact.dtype, device=my_compact.device)
    buf[: my_compact.shape[0]] = my_compact
    gathered = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if buf.is_cuda:
        torch.cuda.synchronize()   # unpack runs on the library's own stream: wait for RCCL's
    if rank == 0:
        for r in range(world):
            if r != 0:
                unpack(r, gathered[r])
