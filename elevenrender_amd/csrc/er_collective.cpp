// er_collective.cpp -- er_comm_* / er_gather_pass of include/eleven_hip.h: see er_collective.h.
#include "er_collective.h"

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <rccl/rccl.h>   // types and prototypes only: every call goes through the table below (dlopen, no link dependency)

#include "er_scene.h"

using namespace erh;

namespace {

// ---- RCCL, resolved at run time ----
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("ER_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
            r.error = dlerror();
        }
        if (!r.lib) return;
        bool ok = true;
        auto sym = [&](const char* name) {
            void* p = dlsym(r.lib, name);
            if (!p) { ok = false; r.error = std::string("missing symbol ") + name; }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(r.lib); r.lib = nullptr; }
    });
    return r;
}

int nccl_fail(const char* what, ncclResult_t e) {
    Rccl& r = rccl();
    return fail(ER_ERR_HIP, std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error"));
}

struct RcclSelf { ncclComm_t comm = nullptr; };

int rccl_group_start(void*) { ncclResult_t e = rccl().GroupStart(); return e == ncclSuccess ? ER_OK : nccl_fail("ncclGroupStart", e); }
int rccl_group_end(void*) { ncclResult_t e = rccl().GroupEnd(); return e == ncclSuccess ? ER_OK : nccl_fail("ncclGroupEnd", e); }
int rccl_send(void* self, const void* buf, size_t bytes, uint32_t peer, hipStream_t st) {
    ncclResult_t e = rccl().Send(buf, bytes, ncclUint8, (int)peer, ((RcclSelf*)self)->comm, st);
    return e == ncclSuccess ? ER_OK : nccl_fail("ncclSend", e);
}
int rccl_recv(void* self, void* buf, size_t bytes, uint32_t peer, hipStream_t st) {
    ncclResult_t e = rccl().Recv(buf, bytes, ncclUint8, (int)peer, ((RcclSelf*)self)->comm, st);
    return e == ncclSuccess ? ER_OK : nccl_fail("ncclRecv", e);
}
void rccl_destroy(void* self) {
    RcclSelf* s = (RcclSelf*)self;
    if (s->comm) (void)rccl().CommDestroy(s->comm);
    delete s;
}

// ---- in-process transport: the `world` ranks of ONE process (one host driving several GPUs from one process -- or several
// ranks on one GPU, as the tests do).  A send parks a device copy of the buffer (on the sender's device) in the shared
// mailbox; the matching recv waits for it and copies it into the receiver's buffer with a peer copy (the runtime moves it over
// xGMI when the devices are peers).  Ranks may call in any order and from different threads. ----
struct Mailbox {
    std::mutex mtx;
    std::condition_variable cv;
    struct Msg { void* copy; size_t bytes; int device; };
    // (src, dst) -> device copies in the order they were sent.  A FIFO, like the wire it stands for: a rank that gathers pass A
    // and then pass B before the root has received A leaves both parked, and the root's receives take A, then B -- an
    // unclaimed message is never dropped or overtaken (with one slot per pair B replaced A and the root's gather of
    // A silently received B's pixels).
    std::map<std::pair<uint32_t, uint32_t>, std::deque<Msg>> slots;
    uint64_t bytes_moved = 0, messages = 0;
    ~Mailbox() {
        for (auto& kv : slots)
            for (auto& m : kv.second) (void)hipFree(m.copy);
    }
};
struct LocalSelf {
    std::shared_ptr<Mailbox> box;
    uint32_t rank;
};
int local_nop(void*) { return ER_OK; }
int local_send(void* self, const void* buf, size_t bytes, uint32_t peer, hipStream_t st) {
    LocalSelf* s = (LocalSelf*)self;
    void* copy = nullptr;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipMalloc(&copy, std::max<size_t>(bytes, 1)));
    hipError_t e = hipMemcpyAsync(copy, buf, bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(copy); return fail(ER_ERR_HIP, std::string("in-process send: ") + hipGetErrorString(e)); }
    {
        std::lock_guard<std::mutex> lk(s->box->mtx);
        s->box->slots[std::make_pair(s->rank, peer)].push_back(Mailbox::Msg{copy, bytes, dev});
        s->box->bytes_moved += bytes;
        s->box->messages++;
    }
    s->box->cv.notify_all();
    return ER_OK;
}
int local_recv(void* self, void* buf, size_t bytes, uint32_t peer, hipStream_t st) {
    LocalSelf* s = (LocalSelf*)self;
    Mailbox::Msg msg{};
    {
        std::unique_lock<std::mutex> lk(s->box->mtx);
        const auto key = std::make_pair(peer, s->rank);
        // (read per call, so that a test can shorten it around ONE negative case without changing it for the whole process)
        const char* wait_env = getenv("ER_LOCAL_RECV_TIMEOUT_S");
        const int wait_s = wait_env ? std::max(0, atoi(wait_env)) : 120;
        if (!s->box->cv.wait_for(lk, std::chrono::seconds(wait_s), [&] { auto f = s->box->slots.find(key); return f != s->box->slots.end() && !f->second.empty(); }))
            return fail(ER_ERR_STATE, "in-process recv: rank " + std::to_string(peer) + " has not sent within " + std::to_string(wait_s) + " s (every rank must call er_gather_pass)");
        auto& q = s->box->slots[key];
        msg = q.front();
        q.pop_front();
    }
    int rc = ER_OK;
    if (msg.bytes != bytes) rc = fail(ER_ERR_STATE, "in-process recv: size mismatch (" + std::to_string(msg.bytes) + " sent, " + std::to_string(bytes) + " expected)");
    if (rc == ER_OK) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = dev == msg.device ? hipMemcpyAsync(buf, msg.copy, bytes, hipMemcpyDeviceToDevice, st) : hipMemcpyPeerAsync(buf, dev, msg.copy, msg.device, bytes, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(ER_ERR_HIP, std::string("in-process recv: ") + hipGetErrorString(e));
    }
    (void)hipFree(msg.copy);
    return rc;
}
void local_destroy(void* self) { delete (LocalSelf*)self; }

}  // namespace

static int er_comm_unique_id_impl(uint8_t* id) {
    if (!id) return fail(ER_ERR_INVALID_ARG, "er_comm_unique_id: NULL argument");
    Rccl& r = rccl();
    if (!r.lib) return fail(ER_ERR_NO_DEVICE, "er_comm_unique_id: RCCL is not loadable (" + r.error + ")");
    static_assert(sizeof(ncclUniqueId) == ER_COMM_ID_BYTES, "ER_COMM_ID_BYTES must be RCCL's NCCL_UNIQUE_ID_BYTES");
    ncclUniqueId u;
    ncclResult_t e = r.GetUniqueId(&u);
    if (e != ncclSuccess) return nccl_fail("ncclGetUniqueId", e);
    memcpy(id, &u, sizeof(u));
    return ER_OK;
}

static int er_comm_create_impl(const uint8_t* id, uint32_t rank, uint32_t world, int device, ErComm** out) {
    if (!id || !out) return fail(ER_ERR_INVALID_ARG, "er_comm_create: NULL argument");
    *out = nullptr;
    if (world == 0 || rank >= world) return fail(ER_ERR_INVALID_ARG, "er_comm_create: rank >= world");
    Rccl& r = rccl();
    if (!r.lib) return fail(ER_ERR_NO_DEVICE, "er_comm_create: RCCL is not loadable (" + r.error + ")");
    if (device < 0 || device >= er_device_count()) return fail(ER_ERR_NO_DEVICE, "er_comm_create: device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<RcclSelf> self(new RcclSelf());
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t e = r.CommInitRank(&self->comm, (int)world, u, (int)rank);
    if (e != ncclSuccess) return nccl_fail("ncclCommInitRank", e);
    ErComm* c = new ErComm();
    c->t = ErTransport{rccl_group_start, rccl_group_end, rccl_send, rccl_recv, rccl_destroy, "rccl"};
    c->self = self.release();
    c->rank = rank; c->world = world; c->device = device;
    *out = c;
    return ER_OK;
}

static int er_comm_create_local_impl(uint32_t world, ErComm** out) {
    if (!out || world == 0) return fail(ER_ERR_INVALID_ARG, "er_comm_create_local: bad argument");
    auto box = std::make_shared<Mailbox>();
    for (uint32_t r = 0; r < world; r++) {
        ErComm* c = new ErComm();
        c->t = ErTransport{local_nop, local_nop, local_send, local_recv, local_destroy, "in-process"};
        c->self = new LocalSelf{box, r};
        c->rank = r; c->world = world; c->device = -1;
        out[r] = c;
    }
    return ER_OK;
}

// pack -> exchange -> unpack of ONE plane (ER_PASS_COUNT = the sample-count plane).  Every rank of the communicator
// calls it with its own scene; after it the root's full plane holds every rank's pixels.
static int er_gather_pass_impl(ErScene* s, int pass, ErComm* c, uint32_t root) {
    if (!s || !c) return fail(ER_ERR_INVALID_ARG, "er_gather_pass: NULL argument");
    if (pass < 0 || pass >= ER_PASS_COUNT) return fail(ER_ERR_INVALID_ARG, "er_gather_pass: pass out of range");
    if (root >= c->world) return fail(ER_ERR_INVALID_ARG, "er_gather_pass: root >= world");
    std::lock_guard<std::mutex> lk(s->mtx);
    if (!s->begun) return fail(ER_ERR_STATE, "er_gather_pass: er_render_begin has not succeeded");
    if (s->params.world != c->world || s->params.rank != c->rank)
        return fail(ER_ERR_INVALID_ARG, "er_gather_pass: the scene's rank/world differ from the communicator's");
    if (c->world == 1) return ER_OK;     // the plane is already whole
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st = s->stream;          // ordered after every sample enqueued so far
    int rc;
    if (c->rank != root) {
        const size_t n = (size_t)s->dev.owned_tile_count * 64;
        DevBuf<float4>& mine = s->d_gather_mine;          // (kept on the scene: allocated by the first gather)
        if (mine.n < n || !mine.p) { if ((rc = upload(mine, (const float4*)nullptr, n, st)) != ER_OK) return rc; }
        er_launch_pack(s->dev, s->d_owned.p, s->dev.owned_tile_count, pass, mine.p, st);
        HIP_TRY(hipGetLastError());
        if ((rc = c->t.group_start(c->self)) != ER_OK) return rc;
        rc = c->t.send(c->self, mine.p, n * sizeof(float4), root, st);
        int rc2 = c->t.group_end(c->self);
        if (rc != ER_OK) return rc;
        if (rc2 != ER_OK) return rc2;
        HIP_TRY(hipStreamSynchronize(st));   // the next gather packs into the same buffer
        return er_scene_stream_status(s, "er_gather_pass");   // (what was sent is incomplete if the streaming schedule stopped early)
    }
    // root: one receive buffer per peer, all receives in ONE group (seven xGMI links side by side), then the scatters
    // (receive buffers and the peers' tile tables live on the scene: the first gather allocates and uploads them, every later one
    // only packs, receives and scatters)
    std::vector<float4*> in(c->world, nullptr);
    std::vector<size_t> counts(c->world, 0);
    for (uint32_t r = 0; r < c->world; r++) {
        if (r == root) continue;
        auto it = s->d_rank_tiles.find(r);
        if (it == s->d_rank_tiles.end()) {
            std::vector<uint32_t> t = s->tiles_of(r, c->world);
            ScopedDevBuf<uint32_t> b;
            if ((rc = upload(b, t.data(), t.size(), st)) != ER_OK) return rc;
            HIP_TRY(hipStreamSynchronize(st));   // t goes out of scope
            it = s->d_rank_tiles.emplace(r, DevBuf<uint32_t>(b)).first;
            b.p = nullptr;
        }
        counts[r] = it->second.n * 64;
        DevBuf<float4>& buf = s->d_gather_in[r];
        if (buf.n < counts[r] || !buf.p) { if ((rc = upload(buf, (const float4*)nullptr, counts[r], st)) != ER_OK) return rc; }
        in[r] = buf.p;
    }
    if ((rc = c->t.group_start(c->self)) != ER_OK) return rc;
    for (uint32_t r = 0; r < c->world && rc == ER_OK; r++)
        if (r != root) rc = c->t.recv(c->self, in[r], counts[r] * sizeof(float4), r, st);
    int rc2 = c->t.group_end(c->self);
    if (rc != ER_OK) return rc;
    if (rc2 != ER_OK) return rc2;
    for (uint32_t r = 0; r < c->world; r++) {
        if (r == root) continue;
        const DevBuf<uint32_t>& tiles = s->d_rank_tiles[r];
        er_launch_unpack(s->dev, tiles.p, (uint32_t)tiles.n, pass, in[r], st);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    for (uint32_t r = 0; r < c->world; r++) if (r != root) s->unpacked[pass].insert(r);
    return er_scene_stream_status(s, "er_gather_pass");
}

static int er_debug_gather_buffers_impl(ErScene* s, uint32_t peer, void** in_ptr, uint64_t* in_bytes, void** mine_ptr, uint64_t* mine_bytes) {
    if (!s) return fail(ER_ERR_INVALID_ARG, "er_debug_gather_buffers: NULL scene");
    std::lock_guard<std::mutex> lk(s->mtx);
    auto it = s->d_gather_in.find(peer);
    if (in_ptr) *in_ptr = it == s->d_gather_in.end() ? nullptr : (void*)it->second.p;
    if (in_bytes) *in_bytes = it == s->d_gather_in.end() ? 0 : (uint64_t)it->second.n * sizeof(float4);
    if (mine_ptr) *mine_ptr = (void*)s->d_gather_mine.p;
    if (mine_bytes) *mine_bytes = (uint64_t)s->d_gather_mine.n * sizeof(float4);
    return ER_OK;
}

static int er_debug_comm_loopback_impl(ErComm* c, uint64_t bytes, double* ms) {
    if (!c || bytes == 0 || bytes % 4 != 0) return fail(ER_ERR_INVALID_ARG, "er_debug_comm_loopback: bad argument");
    if (c->device >= 0) HIP_TRY(hipSetDevice(c->device));
    ScopedDevBuf<uint32_t> src, dst;
    const size_t n = (size_t)(bytes / 4);
    std::vector<uint32_t> pat(n), back(n, 0u);
    for (size_t i = 0; i < n; i++) pat[i] = (uint32_t)(i * 2654435761u) ^ 0x5bd1e995u;
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { if (s) (void)hipStreamDestroy(s); } } guard{st};
    int rc;
    if ((rc = upload(src, pat.data(), n, st)) != ER_OK) return rc;
    if ((rc = upload(dst, (const uint32_t*)nullptr, n, st)) != ER_OK) return rc;
    HIP_TRY(hipMemsetAsync(dst.p, 0, bytes, st));
    HIP_TRY(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = c->t.group_start(c->self)) != ER_OK) return rc;
    rc = c->t.send(c->self, src.p, (size_t)bytes, c->rank, st);
    int rc1 = rc == ER_OK ? c->t.recv(c->self, dst.p, (size_t)bytes, c->rank, st) : ER_OK;
    int rc2 = c->t.group_end(c->self);
    if (rc != ER_OK) return rc;
    if (rc1 != ER_OK) return rc1;
    if (rc2 != ER_OK) return rc2;
    HIP_TRY(hipStreamSynchronize(st));
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    HIP_TRY(hipMemcpy(back.data(), dst.p, bytes, hipMemcpyDeviceToHost));
    if (back != pat) return fail(ER_ERR_STATE, std::string("er_debug_comm_loopback: the bytes received over the ") + c->t.name + " transport differ from the bytes sent");
    return ER_OK;
}

extern "C" {
int er_debug_gather_buffers(ErScene* s, uint32_t peer, void** in_ptr, uint64_t* in_bytes, void** mine_ptr, uint64_t* mine_bytes) {
    return guarded("er_debug_gather_buffers", [&]() -> int { return er_debug_gather_buffers_impl(s, peer, in_ptr, in_bytes, mine_ptr, mine_bytes); });
}
int er_debug_comm_loopback(ErComm* c, uint64_t bytes, double* ms) {
    return guarded("er_debug_comm_loopback", [&]() -> int { return er_debug_comm_loopback_impl(c, bytes, ms); });
}
int er_comm_unique_id(uint8_t* id) { return guarded("er_comm_unique_id", [&]() -> int { return er_comm_unique_id_impl(id); }); }
int er_comm_create(const uint8_t* id, uint32_t rank, uint32_t world, int device, ErComm** out) {
    return guarded("er_comm_create", [&]() -> int { return er_comm_create_impl(id, rank, world, device, out); });
}
void er_comm_destroy(ErComm* c) {
    if (!c) return;
    if (c->t.destroy) c->t.destroy(c->self);
    delete c;
}
int er_gather_pass(ErScene* s, int pass, ErComm* c, uint32_t root) {
    return guarded("er_gather_pass", [&]() -> int { return er_gather_pass_impl(s, pass, c, root); });
}
int er_comm_create_local(uint32_t world, ErComm** out) {
    return guarded("er_comm_create_local", [&]() -> int { return er_comm_create_local_impl(world, out); });
}
int er_debug_comm_create_local(uint32_t world, ErComm** out) { return er_comm_create_local(world, out); }      // (the name the round-2 tests use)
}  // extern "C"
