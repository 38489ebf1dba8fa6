"""The framebuffer combine from the C++ side (include/eleven_hip.h: er_comm_*, er_gather_pass; csrc/er_collective.cpp).

A one-GPU box cannot run RCCL with more than one rank (RCCL refuses two ranks on one device), so:
  * the pack -> exchange -> unpack logic of er_gather_pass is driven for 3 ranks in ONE process over the loopback
    transport of include/eleven_hip_debug.h -- the same code path, only the wire differs;
  * the RCCL transport itself (dlopen, ncclGetUniqueId, ncclCommInitRank, destroy) is exercised with a communicator of
    size 1.  The N-GPU run is bench.py --gpus N, which the driver launches on a whole node.
"""
import ctypes as C
import numpy as np
import pytest

from elevenrender_amd import abi, render, scenes
from elevenrender_amd import dist as erdist
from test_gpu_parity import gpu_render

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sched", [0, abi.FLAG_WAVEFRONT, abi.FLAG_STREAM])
def test_gather_pass_over_the_loopback_transport_reassembles_the_frame(sched, monkeypatch):
    lib = abi.load()
    sc = scenes.soup(5000, 100, 76, seed=9, hdri_size=(64, 32))     # 100x76: partial tiles on both edges
    full = gpu_render(sc, 5, max_bounces=8, flags=sched)
    one = render.RenderingManager(render.RenderParameters(max_bounces=8, flags=sched))   # the single-GPU frame, denoised
    one.start_rendering(sc)
    one.render(5)
    one.denoise()
    full_denoised = one.get_pass("denoise")
    one.close()
    world, root = 3, 1                                              # a root that is not rank 0
    comms = (C.c_void_p * world)()
    abi.check(lib.er_debug_comm_create_local(world, comms))
    rms = []
    for r in range(world):
        rm = render.RenderingManager(render.RenderParameters(max_bounces=8, rank=r, world=world, flags=sched))
        rm.start_rendering(sc)
        rm.render(5, blocking=False)            # no wait: the gather is ordered after the samples on the scene's stream
        rms.append(rm)
    # denoising a sharded frame needs the whole BEAUTY and NORMAL planes: refused until they have been gathered
    assert lib.er_denoise(rms[root].handle, 0, 0.0) == abi.ER_ERR_STATE
    assert b"gather BEAUTY and NORMAL" in lib.er_last_error()
    for p in range(abi.PASS_COUNT):
        for r in [x for x in range(world) if x != root] + [root]:
            abi.check(lib.er_gather_pass(rms[r].handle, p, comms[r], root))
    for name, p in abi.PASS_NAMES.items():
        got = rms[root].get_pass(name)
        assert (got.view(np.uint32) == full[name].view(np.uint32)).all(), name
    # ... then the root denoises the gathered frame exactly as one GPU would its own; the other ranks still cannot
    rms[root].denoise()
    assert (rms[root].get_pass("denoise").view(np.uint32) == full_denoised.view(np.uint32)).all()
    assert lib.er_denoise(rms[0].handle, 0, 0.0) == abi.ER_ERR_STATE
    # a non-root rank still holds only its own pixels
    other = rms[0].get_pass("beauty")
    assert (other.view(np.uint32) != full["beauty"].view(np.uint32)).any()
    # the root before its peers have sent: a state error, not a hang (the in-process receive's timeout, shortened for this
    # ONE call only: the library reads the variable per call, every other test keeps the 120 s default)
    monkeypatch.setenv("ER_LOCAL_RECV_TIMEOUT_S", "2")
    assert lib.er_gather_pass(rms[root].handle, 0, comms[root], root) == abi.ER_ERR_STATE
    assert b"has not sent within" in lib.er_last_error()
    monkeypatch.delenv("ER_LOCAL_RECV_TIMEOUT_S")
    # rank / world of scene and communicator must match
    assert lib.er_gather_pass(rms[0].handle, 0, comms[2], root) == abi.ER_ERR_INVALID_ARG
    for rm in rms:
        rm.close()
    for c in comms:
        lib.er_comm_destroy(c)


def test_in_process_transport_is_a_fifo_two_passes_sent_before_the_first_receive():
    """a non-root rank that gathers pass A and then pass B before the root has received A must not lose A (one
    mailbox slot per (src, dst) used to free A as "an unclaimed older message": the root's gather of A then received B's
    pixels, sizes being equal, and its gather of B timed out)."""
    lib = abi.load()
    sc = scenes.soup(3000, 72, 56, seed=17, hdri_size=(64, 32))
    full = gpu_render(sc, 4, max_bounces=8)
    world, root = 2, 0
    comms = (C.c_void_p * world)()
    abi.check(lib.er_debug_comm_create_local(world, comms))
    rms = []
    for r in range(world):
        rm = render.RenderingManager(render.RenderParameters(max_bounces=8, rank=r, world=world))
        rm.start_rendering(sc)
        rm.render(4)
        rms.append(rm)
    for p in (abi.PASS_BEAUTY, abi.PASS_NORMAL):          # rank 1 sends both planes ...
        abi.check(lib.er_gather_pass(rms[1].handle, p, comms[1], root))
    for p in (abi.PASS_BEAUTY, abi.PASS_NORMAL):          # ... before the root receives the first
        abi.check(lib.er_gather_pass(rms[root].handle, p, comms[root], root))
    for name in ("beauty", "normal"):
        assert (rms[root].get_pass(name).view(np.uint32) == full[name].view(np.uint32)).all(), name
    for rm in rms:
        rm.close()
    for c in comms:
        lib.er_comm_destroy(c)


def test_rccl_transport_initialises_and_a_one_rank_gather_is_a_no_op():
    lib = abi.load()
    ident = (C.c_uint8 * 128)()
    abi.check(lib.er_comm_unique_id(ident))
    assert any(ident)                            # RCCL filled it in
    comm = C.c_void_p()
    abi.check(lib.er_comm_create(ident, 0, 1, 0, C.byref(comm)))
    sc = scenes.cornell(32, 24)
    rm = render.RenderingManager(render.RenderParameters())
    rm.start_rendering(sc)
    rm.render(2)
    before = rm.get_pass("beauty")
    abi.check(lib.er_gather_pass(rm.handle, abi.PASS_BEAUTY, comm, 0))
    assert (rm.get_pass("beauty").view(np.uint32) == before.view(np.uint32)).all()
    assert lib.er_gather_pass(rm.handle, abi.PASS_BEAUTY, comm, 3) == abi.ER_ERR_INVALID_ARG
    rm.close()
    lib.er_comm_destroy(comm)


def test_native_communicator_beside_torch_distributed_rccl():
    """bench.py --gpus N creates the library's RCCL communicator in a process where torch.distributed's "nccl" backend
    (PyTorch's own RCCL) is already initialised; only the 128-byte id goes through torch.  Rehearsed here with one rank:
    both RCCL users coexist, the id broadcast works on the GPU, the gather is a no-op."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from elevenrender_amd import dist as erdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)                                   # PyTorch's RCCL is live
        comm = erdist.NativeComm(dist, 0, 1, 0)
        rm = render.RenderingManager(render.RenderParameters())
        rm.start_rendering(scenes.cornell(32, 24))
        rm.render(2)
        before = rm.get_pass("beauty")
        for p in range(abi.PASS_COUNT):
            comm.gather_pass(rm, p)
        assert (rm.get_pass("beauty").view(np.uint32) == before.view(np.uint32)).all()
        comm.close()
        rm.close()
        dist.all_reduce(t)                                   # ... and still is
        assert float(t[0]) == 1.0
    finally:
        dist.destroy_process_group()


def test_c3_eight_ranks_of_the_c2_frame_gathered_equal_the_single_gpu_frame():
    """BASELINE config 3 as far as one GPU can run it: the 1M-triangle C2 scene at 1920x1080, its 8x8 tiles dealt to EIGHT ranks
    ((tx + ty) % 8, one ErScene per rank, all on this box's one GPU in one process), every rank renders its share -- an eighth of
    the frame is 1 012 pixels per CU, the 12-wave form of the streaming kernel -- and every plane is gathered to rank 0 through the
    library's own er_gather_pass over the in-process transport (the RCCL transport refuses two ranks on one device; the wire is the
    only thing that differs from the 8-GPU run).  The gathered frame must equal the single-GPU frame bit for bit in all five planes;
    the ranks' paths must add up to the frame, once.  Reference shape: one pass per sample over all pixels (src/kernel.cpp:680-706),
    read-back of src/Managers.cpp:287-302."""
    lib = abi.load()
    sc = scenes.soup(1_000_000, 1920, 1080, seed=12345)
    spp, mb = 2, 8
    full = gpu_render(sc, spp, max_bounces=mb)
    world, root = 8, 0
    comms = (C.c_void_p * world)()
    abi.check(lib.er_debug_comm_create_local(world, comms))
    rms = []
    for r in range(world):
        rm = render.RenderingManager(render.RenderParameters(max_bounces=mb, rank=r, world=world))
        rm.start_rendering(sc)
        rm.render(spp, blocking=False)
        rms.append(rm)
    paths = 0
    for rm in rms:
        rm.wait()
        paths += rm.counters()["paths"]
        assert rm.profile()["schedule"] == abi.FLAG_STREAM
    assert paths == 1920 * 1080 * spp
    # (round 6) the buffers of the gather are the exact ones of an 8-way split of a 1080p frame -- 4 050 or 4 051
    # tiles x 64 pixels x 16 bytes per rank, 4.1 MB -- allocated by the FIRST plane's gather and reused by the other four: the same
    # device pointers after every plane (a five-plane read-back from seven peers was 35 hipMalloc / hipFree pairs until round 5)
    def buffers(r, peer):
        ip, ib, mp, mb_ = C.c_void_p(), C.c_uint64(), C.c_void_p(), C.c_uint64()
        abi.check(lib.er_debug_gather_buffers(rms[r].handle, peer, C.byref(ip), C.byref(ib), C.byref(mp), C.byref(mb_)))
        return ip.value, ib.value, mp.value, mb_.value
    tiles_of = [len(erdist.owned_tiles(r, world, 1920, 1080)) for r in range(world)]
    assert sum(tiles_of) == 240 * 135 and max(tiles_of) - min(tiles_of) <= 1
    first = None
    for p in range(abi.PASS_COUNT):
        for r in [x for x in range(world) if x != root] + [root]:
            abi.check(lib.er_gather_pass(rms[r].handle, p, comms[r], root))
        now = {("in", peer): buffers(root, peer)[:2] for peer in range(world) if peer != root}
        now.update({("mine", r): buffers(r, root)[2:] for r in range(world) if r != root})
        if first is None:
            first = now
            for peer in range(world):
                if peer != root:
                    assert now[("in", peer)] == (now[("in", peer)][0], tiles_of[peer] * 64 * 16) and now[("in", peer)][0]
                    assert now[("mine", peer)] == (now[("mine", peer)][0], tiles_of[peer] * 64 * 16) and now[("mine", peer)][0]
            assert buffers(root, root)[:2] == (None, 0)          # the root receives nothing from itself
        assert now == first, p                                   # pointer-stable across the five planes
    for name in abi.PASS_NAMES:
        got = rms[root].get_pass(name)
        assert (got.view(np.uint32) == full[name].view(np.uint32)).all(), name
    for rm in rms:
        rm.close()
    for c in comms:
        lib.er_comm_destroy(c)


def test_rccl_carries_bytes_from_the_c_side_a_self_exchange_through_the_transport_table():
    """"RCCL from C++ has never carried a byte".  RCCL refuses two ranks on one device, so on a one-GPU box the only
    exchange it allows is a rank's send to ITSELF: group start, ncclSend(self), ncclRecv(self), group end -- through the same
    transport table, dlopen'd symbols, datatype and stream arguments er_gather_pass uses -- with one peer buffer of an 8-way 1080p
    split (4 051 tiles x 64 pixels x 16 bytes) and a single float4; the bytes received must be the bytes sent.  The in-process
    transport goes through the same hook."""
    lib = abi.load()
    ident = (C.c_uint8 * 128)()
    abi.check(lib.er_comm_unique_id(ident))
    comm = C.c_void_p()
    abi.check(lib.er_comm_create(ident, 0, 1, 0, C.byref(comm)))
    for nbytes in (16, 4051 * 64 * 16):
        ms = C.c_double()
        abi.check(lib.er_debug_comm_loopback(comm, nbytes, C.byref(ms)))
        print(f"RCCL self exchange of {nbytes} bytes: {ms.value:.3f} ms")
    lib.er_comm_destroy(comm)
    comms = (C.c_void_p * 2)()
    abi.check(lib.er_debug_comm_create_local(2, comms))
    ms = C.c_double()
    abi.check(lib.er_debug_comm_loopback(comms[1], 4051 * 64 * 16, C.byref(ms)))
    assert lib.er_debug_comm_loopback(comms[1], 6, C.byref(ms)) == abi.ER_ERR_INVALID_ARG
    for c in comms:
        lib.er_comm_destroy(c)
