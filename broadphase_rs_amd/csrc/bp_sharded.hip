// bp_sharded.hip -- the multi-GPU Layer behind the C ABI of include/bp_sharded.h (host code only).
//
// One bp_sharded = one rank: a `local` layer (encode target) and a `shard` layer (sort / scan target), both plain
// bp_layer handles driven through the single-GPU C ABI, plus the exchange steps in between.  The transport sits behind
// three collectives (all-gather of small host vectors, all-to-all-v and all-gather-v of device buffers) with two
// implementations: RCCL (one rank per process / GPU; librccl is dlopen-ed) and LOCAL (ranks are host threads of one
// process; device-to-device copies).  Pipeline and invariants: broadphase_rs_amd/dist.py (same steps, same names).
#include "bp_sharded.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace {

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------
typedef struct { char internal[128]; } rccl_uid;
typedef void* rccl_comm;
enum { RCCL_INT8 = 0, RCCL_INT64 = 4 };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(rccl_uid*) = nullptr;
    int (*CommInitRank)(rccl_comm*, int, rccl_uid, int) = nullptr;
    int (*CommDestroy)(rccl_comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, rccl_comm, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, rccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommAbort)(rccl_comm) = nullptr;                 // optional
    int (*CommGetAsyncError)(rccl_comm, int*) = nullptr;   // optional
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // BP_RCCL_LIBRARY: another library with RCCL's entry points (tests/cpp/fake_rccl.cpp: ranks as PROCESSES on one GPU, bytes
        // through shared memory -- the engine's multi-rank code over a real process boundary where the pool has one GPU per box)
        if (const char* over = getenv("BP_RCCL_LIBRARY")) {
            api.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (api.lib) break;
            }
        }
        if (!api.lib) { api.why = std::string("dlopen librccl: ") + (dlerror() ? dlerror() : "not found"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(api.lib, n); if (!p) api.why = std::string("librccl lacks ") + n; return p; };
        api.GetUniqueId = (int (*)(rccl_uid*))sym("ncclGetUniqueId");
        api.CommInitRank = (int (*)(rccl_comm*, int, rccl_uid, int))sym("ncclCommInitRank");
        api.CommDestroy = (int (*)(rccl_comm))sym("ncclCommDestroy");
        api.AllGather = (int (*)(const void*, void*, size_t, int, rccl_comm, hipStream_t))sym("ncclAllGather");
        api.Send = (int (*)(const void*, size_t, int, int, rccl_comm, hipStream_t))sym("ncclSend");
        api.Recv = (int (*)(void*, size_t, int, int, rccl_comm, hipStream_t))sym("ncclRecv");
        api.GroupStart = (int (*)())sym("ncclGroupStart");
        api.GroupEnd = (int (*)())sym("ncclGroupEnd");
        api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
        const std::string keep = api.why;  // the two below are optional: their absence is not an error
        api.CommAbort = (int (*)(rccl_comm))dlsym(api.lib, "ncclCommAbort");
        api.CommGetAsyncError = (int (*)(rccl_comm, int*))dlsym(api.lib, "ncclCommGetAsyncError");
        api.why = keep;
    });
    return &api;
}
bool rccl_ok(const RcclApi* a) {
    return a->lib && a->GetUniqueId && a->CommInitRank && a->CommDestroy && a->AllGather && a->Send && a->Recv && a->GroupStart && a->GroupEnd;
}

thread_local std::string g_group_error;

}  // namespace

enum { GRP_LOCAL = 0, GRP_RCCL = 1 };

struct bp_group {
    int kind = GRP_LOCAL;
    int world = 1;
    std::string err;
    // ---- LOCAL: ranks are threads of this process ------------------------------------------------------------------
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    unsigned long long generation = 0;
    bool broken = false;                                  // a rank failed inside a collective: everyone bails out (RCCL: the communicator was aborted)
    std::vector<std::vector<int64_t>> host_slot;          // all-gather of host vectors: slot per rank
    struct Pub { const void* ptr = nullptr; std::vector<uint64_t> offs; uint64_t n = 0; };
    std::vector<Pub> pub;                                 // device buffers published for an exchange
    // ---- RCCL: this process is ONE rank ------------------------------------------------------------------------------
    int rank = 0, device = 0;
    rccl_comm comm = nullptr;
    hipStream_t stream = nullptr;
    void* d_small = nullptr;                              // staging of the small host vectors
    size_t small_cap = 0;
    void* h_small = nullptr;
};

namespace {

int gfail(bp_group* g, int code, const std::string& what) {
    if (g) g->err = what;
    g_group_error = what;
    return code;
}

// reusable barrier of the LOCAL group; returns false when the group was broken by a failing rank
bool local_barrier(bp_group* g) {
    std::unique_lock<std::mutex> lk(g->mu);
    if (g->broken) return false;
    const unsigned long long gen = g->generation;
    if (++g->waiting == g->world) {
        g->waiting = 0;
        ++g->generation;
        g->cv.notify_all();
    } else {
        g->cv.wait(lk, [&] { return g->generation != gen || g->broken; });
    }
    return !g->broken;
}
void local_break(bp_group* g) {
    std::lock_guard<std::mutex> lk(g->mu);
    g->broken = true;
    g->cv.notify_all();
}

// RCCL: this rank cannot go on with a collective sequence (or a peer did not show up in time).  The communicator is
// aborted -- transfers in flight with this rank end instead of waiting for it -- and the group refuses further use: the
// host tears the job down (or builds a new group); the process image is never restarted from here.
void rccl_poison(bp_group* g) {
    if (g->kind != GRP_RCCL || g->broken) return;
    g->broken = true;
    const RcclApi* a = rccl_api();
    if (g->comm && a->CommAbort) { (void)a->CommAbort(g->comm); g->comm = nullptr; }
}
// wait for the group's stream, but not for ever: a peer that failed between two collectives never posts its side.  Polls the
// stream and the communicator's asynchronous error state; BP_SHARDED_TIMEOUT_MS (default 60000) bounds the wait.
int grp_wait(bp_group* g) {
    const RcclApi* a = rccl_api();
    static const long timeout_ms = getenv("BP_SHARDED_TIMEOUT_MS") ? atol(getenv("BP_SHARDED_TIMEOUT_MS")) : 60000;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(g->stream);
        if (q == hipSuccess) return BP_OK;
        if (q != hipErrorNotReady) { rccl_poison(g); return gfail(g, BP_ERR_HIP, std::string("exchange stream: ") + hipGetErrorString(q)); }
        if ((++spins & 1023u) == 0) {
            if (a->CommGetAsyncError && g->comm) {
                int async = 0;
                if (a->CommGetAsyncError(g->comm, &async) == 0 && async != 0) {
                    const std::string why = std::string("RCCL reported an asynchronous error: ") + (a->GetErrorString ? a->GetErrorString(async) : "error");
                    rccl_poison(g);
                    return gfail(g, BP_ERR_COMM, why);
                }
            }
            const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
            if (ms > timeout_ms) {
                rccl_poison(g);
                return gfail(g, BP_ERR_COMM, "an exchange did not complete within BP_SHARDED_TIMEOUT_MS: a peer rank has failed or left the collective sequence");
            }
        }
    }
}

// test switch: the part of an exchange that stays on the rank also goes through ncclSend / ncclRecv (a one-rank group
// then exercises the whole RCCL binding)
bool self_through_rccl() { return getenv("BP_SHARDED_SELF_SEND") != nullptr; }  // read per call: tests flip it
// per-peer message size cap (RCCL of PyTorch-ROCm 2.10 halves messages > 1 GiB); BP_SHARDED_CHUNK_BYTES: test hook (the multi-round cut on small scenes)
static size_t chunk_bytes() {
    static const size_t v = getenv("BP_SHARDED_CHUNK_BYTES") ? std::max<size_t>(64, (size_t)strtoull(getenv("BP_SHARDED_CHUNK_BYTES"), nullptr, 0)) : (size_t)(256u << 20);
    return v;
}
#define CHUNK_BYTES (chunk_bytes())

#define RCCL_TRY(g, call)                                                                                   \
    do {                                                                                                    \
        int rc__ = (call);                                                                                  \
        if (rc__ != 0) {                                                                                    \
            const RcclApi* a__ = rccl_api();                                                                \
            const std::string why__ = std::string(#call) + ": " + (a__->GetErrorString ? a__->GetErrorString(rc__) : "rccl error"); \
            if (in_group__) (void)a__->GroupEnd();  /* never leave the communicator in group mode */        \
            rccl_poison(g);                                                                                 \
            return gfail((g), BP_ERR_COMM, why__);                                                          \
        }                                                                                                   \
    } while (0)
#define GHIP_TRY(g, call)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) return gfail((g), BP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// every rank's vector of n int64 -> all[world * n] (rank-major)
int grp_allgather_i64(bp_group* g, int rank, const int64_t* mine, size_t n, int64_t* all) {
    if (g->kind == GRP_LOCAL) {
        if (g->world == 1) { std::memcpy(all, mine, n * sizeof(int64_t)); return BP_OK; }
        g->host_slot[rank].assign(mine, mine + n);
        if (!local_barrier(g)) return gfail(g, BP_ERR_COMM, "local group broken");
        for (int r = 0; r < g->world; ++r) std::memcpy(all + (size_t)r * n, g->host_slot[r].data(), n * sizeof(int64_t));
        if (!local_barrier(g)) return gfail(g, BP_ERR_COMM, "local group broken");
        return BP_OK;
    }
    const RcclApi* a = rccl_api();
    if (g->broken) return gfail(g, BP_ERR_COMM, "the RCCL group was aborted by an earlier failure");
    bool in_group__ = false;
    const size_t bytes = n * sizeof(int64_t), total = bytes * (size_t)(g->world + 1);
    if (total > g->small_cap) {
        if (g->d_small) (void)hipFree(g->d_small);
        if (g->h_small) (void)hipHostFree(g->h_small);
        g->d_small = nullptr; g->h_small = nullptr; g->small_cap = 0;
        GHIP_TRY(g, hipMalloc(&g->d_small, total * 2));
        GHIP_TRY(g, hipHostMalloc(&g->h_small, total * 2, hipHostMallocDefault));
        g->small_cap = total * 2;
    }
    char* d_send = (char*)g->d_small;
    char* d_recv = d_send + bytes;
    std::memcpy(g->h_small, mine, bytes);
    GHIP_TRY(g, hipMemcpyAsync(d_send, g->h_small, bytes, hipMemcpyHostToDevice, g->stream));
    RCCL_TRY(g, a->AllGather(d_send, d_recv, n, RCCL_INT64, g->comm, g->stream));
    GHIP_TRY(g, hipMemcpyAsync((char*)g->h_small + bytes, d_recv, bytes * (size_t)g->world, hipMemcpyDeviceToHost, g->stream));
    { const int wrc = grp_wait(g); if (wrc != BP_OK) return wrc; }
    std::memcpy(all, (char*)g->h_small + bytes, bytes * (size_t)g->world);
    return BP_OK;
}

// all-to-all-v of a device buffer that is already grouped by destination: send_counts[d] elements go to rank d,
// recv_counts[s] arrive from rank s (rank order in `recv`); max_msg = the largest single message of ANY rank (elements)
// wait = false (RCCL only): enqueued on the group's stream, the caller orders its consumers behind that stream
int grp_alltoallv(bp_group* g, int rank, const void* send, const uint64_t* send_counts, const uint64_t* recv_counts, size_t elem,
                  void* recv, uint64_t max_msg, hipStream_t local_stream, bool wait = true) {
    const int W = g->world;
    std::vector<uint64_t> soff(W + 1, 0), roff(W + 1, 0);
    for (int r = 0; r < W; ++r) { soff[r + 1] = soff[r] + send_counts[r]; roff[r + 1] = roff[r] + recv_counts[r]; }
    if (g->kind == GRP_LOCAL) {
        g->pub[rank].ptr = send;
        g->pub[rank].offs = soff;
        if (W > 1 && !local_barrier(g)) return gfail(g, BP_ERR_COMM, "local group broken");
        for (int s = 0; s < W; ++s) {
            const bp_group::Pub& p = g->pub[s];
            const uint64_t cnt = p.offs[rank + 1] - p.offs[rank];
            if (cnt != recv_counts[s]) { local_break(g); return gfail(g, BP_ERR_COMM, "all-to-all: count matrix disagrees with the sender"); }
            if (cnt) {
                hipError_t e = hipMemcpyAsync((char*)recv + roff[s] * elem, (const char*)p.ptr + p.offs[rank] * elem, cnt * elem, hipMemcpyDefault, local_stream);
                if (e != hipSuccess) { local_break(g); return gfail(g, BP_ERR_HIP, std::string("all-to-all copy: ") + hipGetErrorString(e)); }
            }
        }
        if (hipStreamSynchronize(local_stream) != hipSuccess) { local_break(g); return gfail(g, BP_ERR_HIP, "all-to-all: stream sync failed"); }
        if (W > 1 && !local_barrier(g)) return gfail(g, BP_ERR_COMM, "local group broken");  // senders may reuse their buffers
        return BP_OK;
    }
    const RcclApi* a = rccl_api();
    if (g->broken) return gfail(g, BP_ERR_COMM, "the RCCL group was aborted by an earlier failure");
    bool in_group__ = false;
    const uint64_t ch = std::max<uint64_t>(1, CHUNK_BYTES / elem);
    const uint64_t rounds = std::max<uint64_t>(1, (max_msg + ch - 1) / ch);
    // the part that stays on this rank is a device copy (about three times the rate of a send to oneself)
    if (send_counts[rank] && !self_through_rccl())
        GHIP_TRY(g, hipMemcpyAsync((char*)recv + roff[rank] * elem, (const char*)send + soff[rank] * elem, send_counts[rank] * elem,
                                   hipMemcpyDeviceToDevice, g->stream));
    for (uint64_t rd = 0; rd < rounds; ++rd) {
        RCCL_TRY(g, a->GroupStart());
        in_group__ = true;
        for (int p = 0; p < W; ++p) {
            if (p == rank && !self_through_rccl()) continue;
            const uint64_t s0 = std::min(send_counts[p], rd * ch), s1 = std::min(send_counts[p], (rd + 1) * ch);
            const uint64_t r0 = std::min(recv_counts[p], rd * ch), r1 = std::min(recv_counts[p], (rd + 1) * ch);
            if (s1 > s0) RCCL_TRY(g, a->Send((const char*)send + (soff[p] + s0) * elem, (s1 - s0) * elem, RCCL_INT8, p, g->comm, g->stream));
            if (r1 > r0) RCCL_TRY(g, a->Recv((char*)recv + (roff[p] + r0) * elem, (r1 - r0) * elem, RCCL_INT8, p, g->comm, g->stream));
        }
        in_group__ = false;
        RCCL_TRY(g, a->GroupEnd());
    }
    if (wait) return grp_wait(g);
    return BP_OK;
}

// all-gather of per-rank device arrays of different lengths (sizes[r] elements), concatenated in rank order
int grp_allgatherv(bp_group* g, int rank, const void* send, const uint64_t* sizes, size_t elem, void* recv, hipStream_t local_stream) {
    const int W = g->world;
    std::vector<uint64_t> sc(W), rc(W);
    uint64_t mx = 0;
    for (int r = 0; r < W; ++r) { rc[r] = sizes[r]; mx = std::max(mx, sizes[r]); }
    if (g->kind == GRP_LOCAL) {
        // every rank sends its whole array to every rank: publish with all offsets at 0 .. n
        g->pub[rank].ptr = send;
        g->pub[rank].n = sizes[rank];
        if (W > 1 && !local_barrier(g)) return gfail(g, BP_ERR_COMM, "local group broken");
        uint64_t off = 0;
        for (int s = 0; s < W; ++s) {
            if (g->pub[s].n != sizes[s]) { local_break(g); return gfail(g, BP_ERR_COMM, "all-gather: sizes disagree with the sender"); }
            if (sizes[s]) {
                hipError_t e = hipMemcpyAsync((char*)recv + off * elem, g->pub[s].ptr, sizes[s] * elem, hipMemcpyDefault, local_stream);
                if (e != hipSuccess) { local_break(g); return gfail(g, BP_ERR_HIP, std::string("all-gather copy: ") + hipGetErrorString(e)); }
            }
            off += sizes[s];
        }
        if (hipStreamSynchronize(local_stream) != hipSuccess) { local_break(g); return gfail(g, BP_ERR_HIP, "all-gather: stream sync failed"); }
        if (W > 1 && !local_barrier(g)) return gfail(g, BP_ERR_COMM, "local group broken");
        return BP_OK;
    }
    // RCCL: as a grouped exchange in which every rank sends the same array to every peer (the arrays are small:
    // boundary keys), cut like any other message
    const RcclApi* a = rccl_api();
    if (g->broken) return gfail(g, BP_ERR_COMM, "the RCCL group was aborted by an earlier failure");
    bool in_group__ = false;
    std::vector<uint64_t> roff(W + 1, 0);
    for (int r = 0; r < W; ++r) roff[r + 1] = roff[r] + sizes[r];
    const uint64_t ch = std::max<uint64_t>(1, CHUNK_BYTES / elem);
    const uint64_t rounds = std::max<uint64_t>(1, (mx + ch - 1) / ch);
    const uint64_t mine = sizes[rank];
    for (uint64_t rd = 0; rd < rounds; ++rd) {
        RCCL_TRY(g, a->GroupStart());
        in_group__ = true;
        for (int p = 0; p < W; ++p) {
            const uint64_t s0 = std::min(mine, rd * ch), s1 = std::min(mine, (rd + 1) * ch);
            const uint64_t r0 = std::min(sizes[p], rd * ch), r1 = std::min(sizes[p], (rd + 1) * ch);
            if (s1 > s0) RCCL_TRY(g, a->Send((const char*)send + s0 * elem, (s1 - s0) * elem, RCCL_INT8, p, g->comm, g->stream));
            if (r1 > r0) RCCL_TRY(g, a->Recv((char*)recv + (roff[p] + r0) * elem, (r1 - r0) * elem, RCCL_INT8, p, g->comm, g->stream));
        }
        in_group__ = false;
        RCCL_TRY(g, a->GroupEnd());
    }
    return grp_wait(g);
}

// Several exchanges in ONE round of grouped sends / receives (one launch latency instead of one per array): the entry
// keys and ids to their owners plus the boundary keys and ids to everyone.  Per transfer and peer: element offsets and
// counts on both sides.  Both ends post the transfers of a peer pair in the same order, which is how RCCL matches them.
struct Xfer {
    const void* send = nullptr;
    void* recv = nullptr;
    size_t elem = 0;
    std::vector<uint64_t> s_off, s_cnt, r_off, r_cnt;  // [world]
};
Xfer xfer_alltoallv(int W, const void* send, const uint64_t* send_counts, const uint64_t* recv_counts, size_t elem, void* recv) {
    Xfer x;
    x.send = send; x.recv = recv; x.elem = elem;
    x.s_off.assign(W, 0); x.s_cnt.assign(W, 0); x.r_off.assign(W, 0); x.r_cnt.assign(W, 0);
    uint64_t so = 0, ro = 0;
    for (int r = 0; r < W; ++r) {
        x.s_off[r] = so; x.s_cnt[r] = send_counts[r]; so += send_counts[r];
        x.r_off[r] = ro; x.r_cnt[r] = recv_counts[r]; ro += recv_counts[r];
    }
    return x;
}
Xfer xfer_allgatherv(int W, int rank, const void* send, const uint64_t* sizes, size_t elem, void* recv) {
    Xfer x;
    x.send = send; x.recv = recv; x.elem = elem;
    x.s_off.assign(W, 0); x.s_cnt.assign(W, sizes[rank]); x.r_off.assign(W, 0); x.r_cnt.assign(W, 0);
    uint64_t ro = 0;
    for (int r = 0; r < W; ++r) { x.r_off[r] = ro; x.r_cnt[r] = sizes[r]; ro += sizes[r]; }
    return x;
}
// max_msg_bytes: the largest single message of ANY rank in ANY of the transfers (every rank passes the same value)
// wait = false: only enqueued on the group's stream; the caller orders its consumers behind that stream
int grp_exchange_rccl(bp_group* g, const std::vector<Xfer>& xs, uint64_t max_msg_bytes, bool wait = true) {
    const RcclApi* a = rccl_api();
    if (g->broken) return gfail(g, BP_ERR_COMM, "the RCCL group was aborted by an earlier failure");
    bool in_group__ = false;
    const int W = g->world;
    const uint64_t rounds = std::max<uint64_t>(1, (max_msg_bytes + CHUNK_BYTES - 1) / CHUNK_BYTES);
    // what stays on this rank is a device copy (about three times the rate of a send to oneself)
    for (const Xfer& x : xs)
        if (x.s_cnt[g->rank] && !self_through_rccl())
            GHIP_TRY(g, hipMemcpyAsync((char*)x.recv + x.r_off[g->rank] * x.elem, (const char*)x.send + x.s_off[g->rank] * x.elem,
                                       x.s_cnt[g->rank] * x.elem, hipMemcpyDeviceToDevice, g->stream));
    for (uint64_t rd = 0; rd < rounds; ++rd) {
        RCCL_TRY(g, a->GroupStart());
        in_group__ = true;
        for (int p = 0; p < W; ++p) {
            if (p == g->rank && !self_through_rccl()) continue;
            for (const Xfer& x : xs) {
                const uint64_t ch = std::max<uint64_t>(1, CHUNK_BYTES / x.elem);
                const uint64_t s0 = std::min(x.s_cnt[p], rd * ch), s1 = std::min(x.s_cnt[p], (rd + 1) * ch);
                const uint64_t r0 = std::min(x.r_cnt[p], rd * ch), r1 = std::min(x.r_cnt[p], (rd + 1) * ch);
                if (s1 > s0) RCCL_TRY(g, a->Send((const char*)x.send + (x.s_off[p] + s0) * x.elem, (s1 - s0) * x.elem, RCCL_INT8, p, g->comm, g->stream));
                if (r1 > r0) RCCL_TRY(g, a->Recv((char*)x.recv + (x.r_off[p] + r0) * x.elem, (r1 - r0) * x.elem, RCCL_INT8, p, g->comm, g->stream));
            }
        }
        in_group__ = false;
        RCCL_TRY(g, a->GroupEnd());
    }
    if (wait) return grp_wait(g);
    return BP_OK;
}

// Contiguous, balanced assignment of Morton cells to ranks (dist.py choose_owner_lut): cell c -> owner.  Deterministic
// in its inputs, so every rank computes the same table from the gathered histograms.
void choose_owner_lut(const std::vector<uint64_t>& hist_fine, int n_ranks, std::vector<uint8_t>& lut) {
    const size_t nc = hist_fine.size();
    lut.assign(nc, 0);
    double total = 0;
    for (uint64_t v : hist_fine) total += (double)v;
    if (total <= 0) return;
    double cum = 0;
    int prev = 0;
    for (size_t c = 0; c < nc; ++c) {
        cum += (double)hist_fine[c];
        const double mid = cum - (double)hist_fine[c] * 0.5;   // entries before the middle of cell c
        int owner = (int)(mid * n_ranks / total);
        owner = std::min(owner, n_ranks - 1);
        owner = std::max(owner, prev);                           // contiguous ranges
        prev = owner;
        lut[c] = (uint8_t)owner;
    }
}

uint64_t varying_bits(uint64_t lo, uint64_t hi) {  // mask of the bits that can differ among integers in [lo, hi]
    if (hi <= lo) return 0;
    const uint64_t x = lo ^ hi;
    const int bits = 64 - __builtin_clzll(x);
    return bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
}

}  // namespace

struct bp_sharded {
    hipEvent_t chain_ev = nullptr;  // orders the group's stream behind the local layer's (entry exchange after the partition pass)
    bp_group* group = nullptr;
    int rank = 0, world = 1, kind = BP_INDEX64_3D, device = 0;
    uint32_t split_depth = 3;
    uint64_t id_count = 0;
    bp_layer_t* local = nullptr;
    bp_layer_t* shard = nullptr;
    hipStream_t stream = nullptr;       // the rank's own exchange stream (LOCAL copies)
    void* d_recv_pairs = nullptr;       // packed pairs received for this rank's first-id range
    size_t recv_cap = 0;
    const uint32_t* d_out = nullptr;
    size_t n_out = 0;
    bp_sharded_info info{};
    std::string err;
};

namespace {
int sfail(bp_sharded* s, int code, const std::string& what) {
    if (s) s->err = what;
    g_group_error = what;
    return code;
}
// this rank gives up in the middle of the collective sequence: its peers must not wait for it for ever.  LOCAL: the
// group's barrier is broken (every waiting rank returns BP_ERR_COMM); RCCL: the communicator is aborted and the peers'
// bounded waits (grp_wait) end.  Failures that every rank sees alike (the gathered status words) return without this.
void leave_collective(bp_sharded* s) {
    if (s->world <= 1) return;
    if (s->group->kind == GRP_LOCAL) local_break(s->group);
    else rccl_poison(s->group);
}
#define L_TRY(s, layer, call)                                                                       \
    do {                                                                                            \
        int rc__ = (call);                                                                          \
        if (rc__ != BP_OK) {                                                                        \
            leave_collective((s));                                                                  \
            return sfail((s), rc__, std::string(#call) + ": " + bp_layer_last_error(layer));        \
        }                                                                                           \
    } while (0)
#define G_TRY(s, call)                                                                  \
    do {                                                                                \
        ++(s)->info.round_trips_exchanges;                                              \
        int rc__ = (call);                                                              \
        if (rc__ != BP_OK) { leave_collective((s)); return sfail((s), rc__, (s)->group->err); } \
    } while (0)
}  // namespace

extern "C" {

const char* bp_group_last_error(const bp_group_t* g) { return g ? g->err.c_str() : g_group_error.c_str(); }
const char* bp_sharded_last_error(const bp_sharded_t* s) { return s ? s->err.c_str() : g_group_error.c_str(); }

int bp_group_unique_id(void* out) {
    if (!out) return BP_ERR_INVALID_ARG;
    const RcclApi* a = rccl_api();
    if (!rccl_ok(a)) return gfail(nullptr, BP_ERR_COMM, "RCCL is not available: " + a->why);
    rccl_uid id;
    const int rc = a->GetUniqueId(&id);
    if (rc != 0) return gfail(nullptr, BP_ERR_COMM, "ncclGetUniqueId failed");
    std::memcpy(out, &id, BP_UNIQUE_ID_BYTES);
    return BP_OK;
}

int bp_group_create_local(bp_group_t** out, int world) {
    if (!out || world < 1 || world > BP_SHARDED_MAX_RANKS)
        return gfail(nullptr, BP_ERR_INVALID_ARG, "bp_group_create_local: world must be 1..16 (BP_SHARDED_MAX_RANKS)");
    bp_group* g = new (std::nothrow) bp_group();
    if (!g) return BP_ERR_OOM;
    g->kind = GRP_LOCAL;
    g->world = world;
    g->host_slot.resize(world);
    g->pub.resize(world);
    *out = g;
    return BP_OK;
}

int bp_group_create_rccl(bp_group_t** out, int rank, int world, const void* uid, int device) {
    if (!out || !uid || world < 1 || world > BP_SHARDED_MAX_RANKS || rank < 0 || rank >= world)
        return gfail(nullptr, BP_ERR_INVALID_ARG, "bp_group_create_rccl: bad rank / world (1..16, BP_SHARDED_MAX_RANKS) / id");
    const RcclApi* a = rccl_api();
    if (!rccl_ok(a)) return gfail(nullptr, BP_ERR_COMM, "RCCL is not available: " + a->why);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return gfail(nullptr, BP_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= count) return gfail(nullptr, BP_ERR_INVALID_ARG, "bp_group_create_rccl: device ordinal out of range");
    bp_group* g = new (std::nothrow) bp_group();
    if (!g) return BP_ERR_OOM;
    g->kind = GRP_RCCL;
    g->world = world;
    g->rank = rank;
    g->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete g; return gfail(nullptr, BP_ERR_NO_DEVICE, "hipSetDevice failed"); }
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) { delete g; return gfail(nullptr, BP_ERR_HIP, "hipStreamCreate failed"); }
    rccl_uid id;
    std::memcpy(&id, uid, BP_UNIQUE_ID_BYTES);
    const int rc = a->CommInitRank(&g->comm, world, id, rank);
    if (rc != 0) {
        const std::string why = std::string("ncclCommInitRank: ") + (a->GetErrorString ? a->GetErrorString(rc) : "error");
        (void)hipStreamDestroy(g->stream);
        delete g;
        return gfail(nullptr, BP_ERR_COMM, why);
    }
    *out = g;
    return BP_OK;
}

int bp_group_destroy(bp_group_t* g) {
    if (!g) return BP_OK;
    if (g->kind == GRP_RCCL) {
        (void)hipSetDevice(g->device);
        if (g->stream) (void)hipStreamSynchronize(g->stream);
        if (g->comm) (void)rccl_api()->CommDestroy(g->comm);
        if (g->d_small) (void)hipFree(g->d_small);
        if (g->h_small) (void)hipHostFree(g->h_small);
        if (g->stream) (void)hipStreamDestroy(g->stream);
    }
    delete g;
    return BP_OK;
}

int bp_sharded_create(bp_sharded_t** out, bp_group_t* group, int rank, int index_kind, uint32_t min_depth, uint32_t split_depth,
                      int device, uint64_t id_count) {
    if (!out || !group || rank < 0 || rank >= group->world || index_kind < 0 || index_kind > 2)
        return gfail(group, BP_ERR_INVALID_ARG, "bp_sharded_create: bad argument");
    if (group->kind == GRP_RCCL && rank != group->rank) return gfail(group, BP_ERR_INVALID_ARG, "bp_sharded_create: rank differs from the group's");
    if (group->world > BP_SHARDED_MAX_RANKS) return gfail(group, BP_ERR_INVALID_ARG, "bp_sharded_create: more than BP_SHARDED_MAX_RANKS (16) ranks");
    const uint32_t dim = index_kind == BP_INDEX64_3D ? 3u : 2u;
    if (split_depth == 0) split_depth = dim == 3 ? 3u : 4u;
    if (dim * split_depth > 12) return gfail(group, BP_ERR_INVALID_ARG, "bp_sharded_create: split_depth too deep (at most 4096 cells)");
    bp_sharded* s = new (std::nothrow) bp_sharded();
    if (!s) return BP_ERR_OOM;
    s->group = group; s->rank = rank; s->world = group->world; s->kind = index_kind; s->device = device;
    s->split_depth = split_depth; s->id_count = id_count;
    int rc = bp_layer_create(&s->local, index_kind, min_depth, 0, 0, device);
    if (rc == BP_OK) rc = bp_layer_create(&s->shard, index_kind, min_depth, 0, 0, device);
    if (rc == BP_OK) rc = bp_layer_device(s->local, &s->device);
    if (rc == BP_OK && (hipSetDevice(s->device) != hipSuccess || hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess)) rc = BP_ERR_HIP;
    if (rc == BP_OK) rc = bp_layer_set_cell_histogram_depth(s->local, split_depth);  // extend counts the splitter histogram itself
    if (rc != BP_OK) {
        g_group_error = "bp_sharded_create: could not create the rank's layers";
        bp_sharded_destroy(s);
        return rc;
    }
    *out = s;
    return BP_OK;
}

int bp_sharded_destroy(bp_sharded_t* s) {
    if (!s) return BP_OK;
    if (s->local) bp_layer_destroy(s->local);
    if (s->shard) bp_layer_destroy(s->shard);
    (void)hipSetDevice(s->device);
    if (s->d_recv_pairs) (void)hipFree(s->d_recv_pairs);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    if (s->chain_ev) (void)hipEventDestroy(s->chain_ev);
    delete s;
    return BP_OK;
}

int bp_sharded_clear(bp_sharded_t* s) {
    if (!s) return BP_ERR_INVALID_ARG;
    return bp_layer_clear(s->local);
}

int bp_sharded_extend(bp_sharded_t* s, const float* sys_min, const float* sys_max, const float* bounds, const uint32_t* ids, size_t n) {
    if (!s) return BP_ERR_INVALID_ARG;
    const int rc = bp_layer_extend(s->local, sys_min, sys_max, bounds, ids, n);
    if (rc != BP_OK) return sfail(s, rc, std::string("extend: ") + bp_layer_last_error(s->local));
    return BP_OK;
}

int bp_sharded_shard_layer(bp_sharded_t* s, bp_layer_t** layer) {
    if (!s || !layer) return BP_ERR_INVALID_ARG;
    *layer = s->shard;
    return BP_OK;
}

int bp_sharded_last_info(const bp_sharded_t* s, bp_sharded_info* out) {
    if (!s || !out) return BP_ERR_INVALID_ARG;
    *out = s->info;
    return BP_OK;
}

int bp_sharded_copy_pairs(bp_sharded_t* s, uint32_t* dst, size_t n_pairs) {
    if (!s || (n_pairs && !dst)) return BP_ERR_INVALID_ARG;
    if (n_pairs > s->n_out) return sfail(s, BP_ERR_INVALID_ARG, "bp_sharded_copy_pairs: more pairs than the last scan returned");
    if (n_pairs == 0) return BP_OK;
    (void)hipSetDevice(s->device);
    if (hipMemcpy(dst, s->d_out, n_pairs * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return sfail(s, BP_ERR_HIP, "hipMemcpy D2H failed");
    return BP_OK;
}

// The pipeline (numbering as in broadphase_rs_amd/dist.py):
//   2. balanced contiguous Morton ranges from the gathered cell histograms (+ every rank's facts in the same exchange)
//   3. fine entries -> owner of their cell (all-to-all), boundary keys -> everyone (all-gather)
//   4. local sort + scan of the shard (coarse-coarse pairs on rank 0 only)
//   5. raw pairs -> owner of their first id (all-to-all), sort (+ dedup) per id range
int bp_sharded_par_scan(bp_sharded_t* s, int pair_order, const uint32_t** d_pairs, size_t* n_pairs) {
    if (!s || !d_pairs || !n_pairs) return BP_ERR_INVALID_ARG;
    bp_group* g = s->group;
    const int G = s->world, rank = s->rank;
    const uint32_t d_s = s->split_depth;
    const uint32_t dim = s->kind == BP_INDEX64_3D ? 3u : 2u;
    const size_t n_cells = (size_t)1 << (dim * d_s);
    if (hipSetDevice(s->device) != hipSuccess) return sfail(s, BP_ERR_HIP, "hipSetDevice failed");
    s->info = bp_sharded_info{};
    s->d_out = nullptr; s->n_out = 0;
    typedef std::chrono::steady_clock clk;
    const clk::time_point t_begin = clk::now();
    clk::time_point t_prev = t_begin;
    auto lap = [&](int stage) {   // every stage ends with this rank's device work complete (the layer calls are synchronous)
        const clk::time_point now = clk::now();
        s->info.stage_ms[stage] += std::chrono::duration<float, std::milli>(now - t_prev).count();
        t_prev = now;
        if (getenv("BP_SHARDED_DEBUG")) {
            bp_stats a, b;
            (void)bp_layer_stats(s->local, &a); (void)bp_layer_stats(s->shard, &b);
            fprintf(stderr, "stage %d: syncs local %llu shard %llu exchanges %u\n", stage, (unsigned long long)a.host_syncs,
                    (unsigned long long)b.host_syncs, s->info.round_trips_exchanges);
        }
    };
    bp_stats st0_local, st0_shard;
    (void)bp_layer_stats(s->local, &st0_local);
    (void)bp_layer_stats(s->shard, &st0_shard);
    auto finish = [&]() {
        s->info.stage_ms[7] = std::chrono::duration<float, std::milli>(clk::now() - t_begin).count();
        bp_stats a, b;
        if (bp_layer_stats(s->local, &a) == BP_OK && bp_layer_stats(s->shard, &b) == BP_OK)
            s->info.round_trips_layers = (uint32_t)((a.host_syncs - st0_local.host_syncs) + (b.host_syncs - st0_shard.host_syncs));
    };

    // ---- 2. cell histogram + facts of every rank in ONE small all-gather ------------------------------------------------
    // A rank that fails BEFORE the first exchange still takes part in it: its status word rides with the facts, every rank
    // reads every rank's status, and all of them return together -- nobody is left waiting in a collective.
    std::vector<uint64_t> hist(n_cells + 1, 0);
    bp_facts lf;
    std::memset(&lf, 0, sizeof lf);
    int st_rc = bp_layer_cell_histogram(s->local, d_s, hist.data(), n_cells);
    std::string st_why = st_rc == BP_OK ? std::string() : std::string("bp_layer_cell_histogram: ") + bp_layer_last_error(s->local);
    if (st_rc == BP_OK) {
        st_rc = bp_layer_get_facts(s->local, &lf);
        if (st_rc != BP_OK) st_why = std::string("bp_layer_get_facts: ") + bp_layer_last_error(s->local);
    }
    if (st_rc != BP_OK) std::fill(hist.begin(), hist.end(), 0ull);
    uint64_t m_local = 0;
    for (uint64_t v : hist) m_local += v;
    enum { F_KOR, F_KAND, F_IOR, F_IAND, F_IMIN, F_IMAX, F_DM, F_FLAGS, F_M, F_STATUS, F_IDCOUNT, F_N };
    const size_t vlen = n_cells + 1 + F_N;
    std::vector<int64_t> mine(vlen), all(vlen * (size_t)G);
    for (size_t c = 0; c <= n_cells; ++c) mine[c] = (int64_t)hist[c];
    int64_t* f = mine.data() + n_cells + 1;
    f[F_KOR] = (int64_t)lf.key_or; f[F_KAND] = (int64_t)lf.key_and; f[F_IOR] = lf.id_or; f[F_IAND] = lf.id_and;
    f[F_IMIN] = lf.id_min; f[F_IMAX] = lf.id_max; f[F_DM] = lf.depth_mask; f[F_FLAGS] = lf.flags; f[F_M] = (int64_t)m_local;
    f[F_STATUS] = st_rc; f[F_IDCOUNT] = (int64_t)s->id_count;
    if (st_rc != BP_OK) { f[F_IMIN] = 0xFFFFFFFFll; f[F_IMAX] = 0; }  // no ids to speak of
    G_TRY(s, grp_allgather_i64(g, rank, mine.data(), vlen, all.data()));
    {   // agreement on what every rank brought: all ranks evaluate the same gathered words and take the same exit
        auto factq = [&](int r, int k) { return all[(size_t)r * vlen + n_cells + 1 + k]; };
        for (int r = 0; r < G; ++r)
            if (factq(r, F_STATUS) != BP_OK) {
                if (r == rank) return sfail(s, st_rc, st_why);
                return sfail(s, BP_ERR_COMM, "rank " + std::to_string(r) + " failed before the exchange (status " + std::to_string((long long)factq(r, F_STATUS)) + ")");
            }
        for (int r = 0; r < G; ++r)
            if ((uint64_t)factq(r, F_IDCOUNT) != s->id_count)
                return sfail(s, BP_ERR_INVALID_ARG, "bp_sharded_create: id_count differs between the ranks");
        if (s->id_count)  // the caller vouched for ids in [0, id_count): a larger one would be mis-sorted silently by the owners' digit plans
            for (int r = 0; r < G; ++r)
                if (factq(r, F_IMIN) <= factq(r, F_IMAX) && (uint64_t)factq(r, F_IMAX) >= s->id_count)
                    return sfail(s, BP_ERR_INVALID_ARG, "an id handed to rank " + std::to_string(r) + " is not below id_count (" +
                                                            std::to_string((long long)factq(r, F_IMAX)) + " >= " + std::to_string((unsigned long long)s->id_count) + ")");
    }
    std::vector<uint64_t> hist_all(n_cells, 0);
    for (int r = 0; r < G; ++r)
        for (size_t c = 0; c < n_cells; ++c) hist_all[c] += (uint64_t)all[(size_t)r * vlen + c];
    std::vector<uint8_t> lut;
    choose_owner_lut(hist_all, G, lut);
    lap(0);
    auto fact = [&](int r, int k) { return all[(size_t)r * vlen + n_cells + 1 + k]; };
    // packing (box codes inside the ids) must be decided before the partition, by every rank alike
    bool any_nonempty = false, pack = true, all_known = true;
    for (int r = 0; r < G; ++r) {
        if (fact(r, F_M) <= 0) continue;   // a rank without entries takes no part in the AND / min / max reductions
        any_nonempty = true;
        if (!(fact(r, F_FLAGS) & BP_FACT_IDS_CARRY_BOX)) pack = false;
        if (!(fact(r, F_FLAGS) & BP_FACT_BOX_ALL_KNOWN)) all_known = false;
    }
    if (!any_nonempty) pack = false;
    all_known = all_known && pack;

    // ---- 3. entries -> owners (fine), boundary keys -> everyone (coarse) ---------------------------------------------------
    std::vector<uint64_t> counts(G + 1, 0);
    const uint64_t* d_keys = nullptr;
    const uint32_t* d_ids = nullptr;
    // RCCL: the partition pass, the exchange and the shard's kernels are chained by events (partition on the local layer's
    // stream -> exchange on the group's stream -> shard layer's stream); the host waits for none of them
    const bool chained = g->kind == GRP_RCCL && !getenv("BP_SHARDED_SYNC_EXCHANGE");
    L_TRY(s, s->local, bp_layer_partition_entries_packed_device(s->local, d_s, lut.data(), n_cells, (uint32_t)G,
                                                                (pack ? BP_PARTITION_PACK_BOX : 0) | (chained ? BP_PARTITION_ENQUEUE_ONLY : 0),
                                                                counts.data(), &d_keys, &d_ids));
    if (chained) {
        void* local_stream = nullptr;
        L_TRY(s, s->local, bp_layer_stream(s->local, &local_stream));
        if (!s->chain_ev && hipEventCreateWithFlags(&s->chain_ev, hipEventDisableTiming) != hipSuccess) {
            leave_collective(s);
            return sfail(s, BP_ERR_HIP, "hipEventCreate failed");
        }
        if (hipEventRecord(s->chain_ev, (hipStream_t)local_stream) != hipSuccess || hipStreamWaitEvent(g->stream, s->chain_ev, 0) != hipSuccess) {
            leave_collective(s);
            return sfail(s, BP_ERR_HIP, "could not order the exchange behind the partition pass");
        }
    }
    lap(1);
    // cm[src][dst] entries go from src to dst, cm[src][G] = src's boundary keys: every rank derives the whole matrix from
    // the gathered histograms and the table (no second exchange of counts)
    std::vector<uint64_t> cm((size_t)(G + 1) * G, 0);
    for (int r = 0; r < G; ++r) {
        for (size_t c = 0; c < n_cells; ++c) cm[(size_t)r * (G + 1) + lut[c]] += (uint64_t)all[(size_t)r * vlen + c];
        cm[(size_t)r * (G + 1) + G] = (uint64_t)all[(size_t)r * vlen + n_cells];
    }
    for (int k = 0; k <= G; ++k)
        if (cm[(size_t)rank * (G + 1) + k] != counts[k]) {
            leave_collective(s);
            return sfail(s, BP_ERR_COMM, "partition counts disagree with the cell histogram");
        }
    std::vector<uint64_t> recv_counts(G), coarse_sizes(G), send_counts(G);
    uint64_t n_recv = 0, n_coarse = 0, n_fine = 0, max_msg = 0;
    for (int r = 0; r < G; ++r) {
        recv_counts[r] = cm[(size_t)r * (G + 1) + rank];
        coarse_sizes[r] = cm[(size_t)r * (G + 1) + G];
        send_counts[r] = counts[r];
        n_recv += recv_counts[r]; n_coarse += coarse_sizes[r]; n_fine += counts[r];
        for (int d = 0; d <= G; ++d) max_msg = std::max(max_msg, cm[(size_t)r * (G + 1) + d]);
    }
    uint64_t* sk = nullptr;
    uint32_t* si = nullptr;
    L_TRY(s, s->shard, bp_layer_begin_entries(s->shard, (size_t)(n_recv + n_coarse), &sk, &si));
    if (g->kind == GRP_RCCL) {
        std::vector<Xfer> xs;
        xs.push_back(xfer_alltoallv(G, d_keys, send_counts.data(), recv_counts.data(), sizeof(uint64_t), sk));
        xs.push_back(xfer_alltoallv(G, d_ids, send_counts.data(), recv_counts.data(), sizeof(uint32_t), si));
        if (n_coarse) {
            xs.push_back(xfer_allgatherv(G, rank, d_keys + n_fine, coarse_sizes.data(), sizeof(uint64_t), sk + n_recv));
            xs.push_back(xfer_allgatherv(G, rank, d_ids + n_fine, coarse_sizes.data(), sizeof(uint32_t), si + n_recv));
        }
        G_TRY(s, grp_exchange_rccl(g, xs, max_msg * sizeof(uint64_t), /*wait=*/!chained));
        if (chained) {  // the shard's kernels read what arrives; the local layer's next frame reuses what is being sent
            --s->info.round_trips_exchanges;  // nobody waited
            L_TRY(s, s->shard, bp_layer_wait_stream(s->shard, (void*)g->stream));
            L_TRY(s, s->local, bp_layer_wait_stream(s->local, (void*)g->stream));
        }
    } else {
        G_TRY(s, grp_alltoallv(g, rank, d_keys, send_counts.data(), recv_counts.data(), sizeof(uint64_t), sk, max_msg, s->stream));
        G_TRY(s, grp_alltoallv(g, rank, d_ids, send_counts.data(), recv_counts.data(), sizeof(uint32_t), si, max_msg, s->stream));
        if (n_coarse) {
            G_TRY(s, grp_allgatherv(g, rank, d_keys + n_fine, coarse_sizes.data(), sizeof(uint64_t), sk + n_recv, s->stream));
            G_TRY(s, grp_allgatherv(g, rank, d_ids + n_fine, coarse_sizes.data(), sizeof(uint32_t), si + n_recv, s->stream));
        }
    }
    lap(2);
    // facts of the union.  Chunks arrive in rank order and every rank's entries are in its own append order, so the order /
    // distinctness facts hold iff they hold locally everywhere and the ranks' id ranges ascend with the rank
    bp_facts uf;
    std::memset(&uf, 0, sizeof uf);
    uf.key_and = ~0ull; uf.id_and = 0xFFFFFFFFu; uf.id_min = 0xFFFFFFFFu;
    bool ascending = true, order_all = true, distinct_all = true;
    int64_t prev_max = -1;
    uint64_t max_id_seen = 0;
    bool any_ids = false;
    for (int r = 0; r < G; ++r) {
        if (fact(r, F_IMIN) <= fact(r, F_IMAX)) { any_ids = true; max_id_seen = std::max<uint64_t>(max_id_seen, (uint64_t)fact(r, F_IMAX)); }
        if (fact(r, F_M) <= 0) continue;
        uf.key_or |= (uint64_t)fact(r, F_KOR); uf.key_and &= (uint64_t)fact(r, F_KAND);
        uf.id_or |= (uint32_t)fact(r, F_IOR); uf.id_and &= (uint32_t)fact(r, F_IAND);
        uf.id_min = std::min(uf.id_min, (uint32_t)fact(r, F_IMIN)); uf.id_max = std::max(uf.id_max, (uint32_t)fact(r, F_IMAX));
        uf.depth_mask |= (uint32_t)fact(r, F_DM);
        if (prev_max >= 0 && !(prev_max < fact(r, F_IMIN))) ascending = false;
        prev_max = fact(r, F_IMAX);
        if (!(fact(r, F_FLAGS) & BP_FACT_ORDER_OK)) order_all = false;
        if (!(fact(r, F_FLAGS) & BP_FACT_IDS_DISTINCT)) distinct_all = false;
    }
    if (!any_nonempty) { uf.key_and = 0; uf.id_and = 0; uf.id_min = 0; }
    uf.flags = ((ascending && order_all) ? BP_FACT_ORDER_OK : 0u) | ((ascending && distinct_all) ? BP_FACT_IDS_DISTINCT : 0u) |
               (pack ? BP_FACT_IDS_CARRY_BOX : 0u) | (all_known ? BP_FACT_BOX_ALL_KNOWN : 0u);
    L_TRY(s, s->shard, bp_layer_commit_entries(s->shard, (size_t)(n_recv + n_coarse), &uf));
    lap(3);
    s->info.entries_local = m_local;
    s->info.entries_shard = n_recv + n_coarse;
    s->info.coarse = n_coarse;
    s->info.box_codes = pack ? 1u : 0u;

    // ---- 4. + 5. ----------------------------------------------------------------------------------------------------------
    const uint64_t max_id = s->id_count ? s->id_count - 1 : (any_ids ? max_id_seen : 0);
    const uint64_t span = (max_id + (uint64_t)G) / (uint64_t)G;
    std::vector<uint32_t> bounds(std::max(1, G - 1), 0xFFFFFFFFu);   // G - 1 bounds are used (never an empty vector: its data() is null)
    const uint32_t n_bounds = (uint32_t)(G - 1);
    for (int r = 0; r + 1 < G; ++r) bounds[r] = (uint32_t)std::min<uint64_t>(span * (uint64_t)(r + 1), 0xFFFFFFFFull);
    // pairs between two boundary entries (replicated on every rank) are emitted by the rank that owns the deeper entry's id
    // -- the first id of the pair, so on the GLOBAL route they are already where they belong -- instead of by rank 0 alone:
    // in a heavy-tailed scene (configs[4]) they are a large share of the pairs
    const uint32_t own_lo = rank == 0 ? 0u : bounds[rank - 1];
    const uint32_t own_hi = rank == G - 1 ? 0xFFFFFFFFu : (bounds[rank] == 0 ? 0u : bounds[rank] - 1u);
    const bool own_empty = rank != G - 1 && (bounds[rank] == 0 || bounds[rank] <= own_lo);
    static const bool coarse_rank0 = getenv("BP_SHARDED_COARSE_RANK0") != nullptr;  // A/B: the round-2 behaviour
    const int emit_coarse = (G == 1 || coarse_rank0) ? (rank == 0 ? 1 : 0) : 2;
    if (emit_coarse == 2) L_TRY(s, s->shard, bp_layer_set_coarse_emit_range(s->shard, own_empty ? 1u : own_lo, own_empty ? 0u : own_hi));
    // LOCAL order: possible while every shard suppresses its duplicates at the source -- box codes known for every entry and
    // ids provably distinct, facts every rank holds alike, so the route is decided without another exchange
    static const bool no_source_dedup = getenv("BP_NO_SOURCE_DEDUP") != nullptr;  // the layers' measurement switch
    if (pair_order == BP_PAIRS_LOCAL && pack && all_known && (uf.flags & BP_FACT_IDS_DISTINCT) && !no_source_dedup) {
        const uint32_t* dp = nullptr;
        size_t np = 0;
        int unique = 0;
        L_TRY(s, s->shard, bp_layer_par_scan_shard_device(s->shard, d_s, emit_coarse, &dp, &np, &unique));
        lap(4);
        if (!unique && n_recv + n_coarse != 0)  // cannot happen with these facts; the other ranks have already taken this route
            return sfail(s, BP_ERR_KERNEL, "the shard scan did not suppress its duplicates although every box code is known");  // (no exchange follows on this route)
        s->d_out = dp; s->n_out = np;
        s->info.raw_pairs = np; s->info.pairs = np; s->info.id_lo = 0; s->info.id_hi = (uint32_t)max_id; s->info.pair_order = BP_PAIRS_LOCAL;
        *d_pairs = dp; *n_pairs = np;
        finish();
        return BP_OK;
    }
    std::vector<uint64_t> pcounts(G, 0);
    const uint64_t* d_grouped = nullptr;
    size_t n_raw = 0;
    int unique = 0;
    L_TRY(s, s->shard, bp_layer_scan_shard_partition_device(s->shard, d_s, emit_coarse, bounds.data(), n_bounds, pcounts.data(), &d_grouped, &n_raw, &unique));
    lap(4);
    // the final step may skip dedup only if EVERY rank suppressed its duplicates at the source: the flag rides with the counts
    std::vector<int64_t> pmine(G + 1), pm((size_t)(G + 1) * G);
    for (int k = 0; k < G; ++k) pmine[k] = (int64_t)pcounts[k];
    pmine[G] = (unique && pack && all_known) ? 0 : 1;
    G_TRY(s, grp_allgather_i64(g, rank, pmine.data(), (size_t)G + 1, pm.data()));
    std::vector<uint64_t> precv(G);
    uint64_t n_in = 0, pmax = 0;
    bool all_unique = true;
    for (int r = 0; r < G; ++r) {
        precv[r] = (uint64_t)pm[(size_t)r * (G + 1) + rank];
        n_in += precv[r];
        all_unique = all_unique && pm[(size_t)r * (G + 1) + G] == 0;
        for (int d = 0; d < G; ++d) pmax = std::max(pmax, (uint64_t)pm[(size_t)r * (G + 1) + d]);
    }
    if (n_in * sizeof(uint64_t) > s->recv_cap) {
        if (s->d_recv_pairs) (void)hipFree(s->d_recv_pairs);
        s->d_recv_pairs = nullptr; s->recv_cap = 0;
        const size_t want = (size_t)(n_in + n_in / 8 + 1024) * sizeof(uint64_t);
        if (hipMalloc(&s->d_recv_pairs, want) != hipSuccess) {
            leave_collective(s);
            return sfail(s, BP_ERR_OOM, "hipMalloc of the pair receive buffer failed");
        }
        s->recv_cap = want;
    }
    // (the bucketing pass has been waited for -- its sizes were needed for the all-gather above -- so the group's stream may start
    // at once; the shard's sort is ordered behind the exchange by an event)
    G_TRY(s, grp_alltoallv(g, rank, d_grouped, pcounts.data(), precv.data(), sizeof(uint64_t), s->d_recv_pairs, pmax, s->stream, /*wait=*/!chained));
    if (chained) {
        --s->info.round_trips_exchanges;
        L_TRY(s, s->shard, bp_layer_wait_stream(s->shard, (void*)g->stream));
    }
    lap(5);
    const uint64_t lo = rank == 0 ? 0 : bounds[rank - 1];
    const uint64_t hi_raw = rank == G - 1 ? max_id : (uint64_t)bounds[rank] - 1;
    const uint64_t hi = std::max(hi_raw, lo);
    const uint64_t varying = (varying_bits(lo, hi) << 32) | varying_bits(0, max_id);
    const uint32_t* dp = nullptr;
    size_t np = 0;
    if (all_unique)
        L_TRY(s, s->shard, bp_layer_sort_packed_range_device(s->shard, (uint64_t*)s->d_recv_pairs, (size_t)n_in, varying, (uint32_t)lo, (uint32_t)hi, &dp, &np));
    else
        L_TRY(s, s->shard, bp_layer_sort_unique_packed_device(s->shard, (uint64_t*)s->d_recv_pairs, (size_t)n_in, varying, &dp, &np));
    lap(6);
    finish();
    s->d_out = dp; s->n_out = np;
    s->info.raw_pairs = n_raw; s->info.raw_pairs_recv = n_in; s->info.pairs = np;
    s->info.id_lo = (uint32_t)lo; s->info.id_hi = (uint32_t)hi_raw; s->info.pair_order = BP_PAIRS_GLOBAL;
    *d_pairs = dp; *n_pairs = np;
    return BP_OK;
}

}  // extern "C"
