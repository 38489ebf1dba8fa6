// er_collective.h -- the framebuffer combine of the tile-sharded multi-GPU render, done from the C++ side.
//
// north_star: "pixel tiles shard embarrassingly across the 8 GPUs of one node with an RCCL reduce over xGMI only for
// the final framebuffer accumulate".  Tile ownership is disjoint, so the "reduce" is a gather: every rank packs the
// pixels it owns into a compact device buffer (er_pack_kernel), the non-root ranks ncclSend it to the root, the root
// ncclRecv's the 7 buffers inside one group (xGMI is point to point: seven direct links carry them side by side; a ring
// collective would be per-link bound and move 7x the bytes) and scatters each into its full plane (er_unpack_kernel).
// One call per read-back (reference hook: RenderingManager::get_pass, src/Managers.cpp:287-302), none per sample.
//
// The wire is behind a small transport table so that the pack -> exchange -> unpack logic of er_gather_pass can be
// driven in ONE process on ONE GPU by a loopback transport (include/eleven_hip_debug.h: er_debug_comm_create_local);
// the production transport is RCCL, resolved at run time with dlopen so that the library has no link-time dependency
// on it and shares the process's RCCL when the host (e.g. PyTorch) has already loaded one.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

struct ErTransport {
    // all calls are made by the rank that owns `self`; buffers are device pointers; `stream` orders the transfer
    int (*group_start)(void* self);
    int (*group_end)(void* self);
    int (*send)(void* self, const void* dev_buf, size_t bytes, uint32_t peer, hipStream_t stream);
    int (*recv)(void* self, void* dev_buf, size_t bytes, uint32_t peer, hipStream_t stream);
    void (*destroy)(void* self);
    const char* name;
};

struct ErComm {
    ErTransport t;
    void* self;
    uint32_t rank, world;
    int device;
};
