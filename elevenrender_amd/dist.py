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


def gather_all_planes_torch(dist, rm, rank, world, planes=(0, 1, 2, 3, 4)):
    """All pass planes through torch.distributed (the test-harness path: gloo on the CPU tests / rehearsal; the
    production combine is NativeComm below).  `rm` is a render.RenderingManager begun with this rank / world."""
    import torch
    rows = [rm.owned_count(r) for r in range(world)]
    on_cuda = dist.get_backend() == "nccl"
    for p in planes:
        # torch.empty: er_pack_owned writes every row, and it runs on the library's own stream -- a torch.zeros fill
        # on torch's stream could land after the pack and blank it (round-1 advice)
        mine = torch.empty((rows[rank], 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        rm.pack_owned(p, mine.data_ptr())          # synchronises the library's stream before returning

        def unpack(r, t, p=p):
            t = t.contiguous().cuda()
            torch.cuda.synchronize()
            rm.unpack_owned(p, r, t.data_ptr())
        gather_plane(dist, rank, world, mine if on_cuda else mine.cpu(), max(rows), unpack)


class NativeComm:
    """The library's own RCCL communicator (include/eleven_hip.h: er_comm_*, er_gather_pass).  Only the 128-byte
    unique id travels through the host's channel -- here a torch.distributed broadcast; the pixel data moves from the
    C++ side with ncclSend / ncclRecv over xGMI."""

    def __init__(self, dist, rank, world, device):
        import ctypes as C
        import torch
        from . import abi
        self.lib = abi.load()
        self.handle = C.c_void_p()
        on_gpu = dist.get_backend() == "nccl"
        dev = "cuda" if on_gpu else "cpu"
        # 1. every rank checks locally that RCCL can be loaded at all (er_comm_unique_id dlopens it); the ranks agree
        #    BEFORE anybody enters the collective ncclCommInitRank, so that a rank without RCCL cannot strand the others
        ident = (C.c_uint8 * 128)()
        rc = self.lib.er_comm_unique_id(ident)
        ok = torch.tensor([1 if rc == abi.ER_OK else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) != 1:
            raise abi.ErError(rc if rc != abi.ER_OK else abi.ER_ERR_NO_DEVICE,
                              "er_comm_unique_id failed on at least one rank: " + self.lib.er_last_error().decode("utf-8", "replace"))
        # 2. rank 0's id travels through the host's channel (every other rank's probe id is discarded)
        t = torch.tensor(list(ident), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0)
        ident = (C.c_uint8 * 128)(*t.cpu().tolist())
        # 3. collective: ncclCommInitRank on every rank
        abi.check(self.lib.er_comm_create(ident, rank, world, device, C.byref(self.handle)))

    def gather_pass(self, rm, pass_id, root=0):
        from . import abi
        abi.check(self.lib.er_gather_pass(rm.handle, pass_id, self.handle, root))

    def close(self):
        if self.handle:
            self.lib.er_comm_destroy(self.handle)
            self.handle = None
