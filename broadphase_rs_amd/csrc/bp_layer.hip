// bp_layer.hip -- the C ABI of include/bp_layer.h over the gfx950 kernels (host code only).
//
// One bp_layer owns: one HIP stream, the device-resident tree (SoA ping-pong buffers), all scan
// temporaries, and pinned host mirrors for the host-visible views.  There is no CPU fallback:
// every entry point fails with BP_ERR_NO_DEVICE / BP_ERR_HIP if the device path is unusable.
#include "bp_layer.h"
#include "bp_kernels.h"

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace bp;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;  // bytes
};
struct HostBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct ProfRec {
    int cls;
    hipEvent_t a, b;
};

}  // namespace

struct bp_layer {
    int kind = BP_INDEX64_3D;
    int device = 0;
    uint32_t min_depth = 0;
    hipStream_t stream = nullptr;
    hipEvent_t fence_ev = nullptr;   // bp_layer_wait_stream: orders this layer's stream after a producer's
    hipEvent_t early_ev = nullptr;   // extend on predicted sizes: the scalars' copy, waited for while encode_emit already runs
    // results the host waits for in the per-frame path arrive in pinned host memory, written by the kernels themselves (struct
    // Published): [0, PUB_FLAG_OFF) the data, then the flag word; pub_seq numbers the launches
    void* pub = nullptr;
    uint32_t pub_seq = 0;
    size_t pairs_want_slots = 0;       // raw-pair buffer size the cell route of the pair sort asked for (scan_fused_device)

    // persistent state: tree (src/layer.rs:49-50)
    DevBuf keys[2], ids[2];
    DevBuf box[2];               // per-entry BOX_* code, aligned with keys/ids while box_valid
    bool box_valid = true;       // box[cur] describes the entries in their current order
    bool box_all_known = true;   // no BOX_UNKNOWN code present
    bool ids_packed = false;     // ids[cur] still carry box codes in bits 27..31 (bp_layer_commit_entries)
    // compact records (TreeCompact, bp_kernels.h; between a predicted extend and the sort that follows it): keys[cur] holds ONE
    // composite word per entry -- the index bits that vary, the id, the box code -- and ids[cur] / box[cur] nothing; the cell route
    // of the tree sort moves these 8-byte records (ensure_plain expands them for every other consumer)
    bool tree_compact = false;
    TreeCompact cfmt{};
    // after a compact sort: ids[cur] / box[cur] have NOT been written -- the scan reads the {id, context} words, and whoever wants
    // the arrays gets them derived from those words first (ensure_ids; every code known, idctx current)
    bool ids_lazy = false;
    int cur = 0;
    size_t m = 0;
    bool sorted = true;
    // static-layer reuse: entries [run_split, m) were merged in from a SORTED layer and nothing was appended
    // since, so the next sort only has to sort [0, run_split) and merge the two runs
    bool tail_run = false;
    size_t run_split = 0;
    // multi-GPU: the last cell histogram, reused by the entry partition while the tree is unchanged
    std::vector<unsigned long long> cell_hist_cache;
    uint32_t cell_hist_depth = 0;
    bool cell_hist_valid = false;
    uint32_t cell_fuse_depth = 0;       // > 0: extend counts the cell histogram at this depth while it emits
    DevBuf cell_hist_dev;               // its HIST_COPIES copies on the device ...
    bool cell_dev_valid = false;        // ... describe the current tree (exactly one extend since clear)

    // facts about the tree that steer kernel selection (never change results)
    bool have_ids = false;      // anything appended since the last clear
    bool order_ok = true;       // equal keys are already id-ascending in current order
    bool ids_distinct = true;   // no id occurs on two objects (skip rule cannot fire)
    uint32_t max_id = 0, min_id = 0xFFFFFFFFu;   // id range of everything appended since the last clear
    uint64_t key_or = 0, key_and = ~0ull;
    uint32_t id_or = 0, id_and = 0xFFFFFFFFu;
    uint32_t depth_mask = 0;

    // temporaries
    DevBuf in_bounds, in_ids;           // staging for host inputs
    DevBuf counts, block_sums, aux, skip, cell_sum;
    DevBuf pairs[2];                    // packed raw pairs (ping-pong for the pair sort)
    DevBuf out_pairs;                   // final interleaved (id, id_) list
    DevBuf radix_ws;
    DevBuf d_sc;                        // DevScalars
    HostBuf h_sc, h_pairs, h_keys, h_ids, h_raw, h_cells;
    size_t n_raw = 0, n_pairs = 0;
    bool raw_is_unique = false;         // the last raw emission already suppressed every duplicate
    DevBuf idctx, big_list;             // expanding scan: per-entry id << 32 | suppression context; long-range work list
    DevBuf dcnt;                        // per-entry depth | range length << 5, left by the compact tree finish (dcnt_epoch: for which tree)
    uint64_t dcnt_epoch = 0;
    size_t big_cap = 0;                 // work-list capacity in use
    DevBuf qtab;                        // region queries: slice bounds of the top levels of the sorted tree ...
    uint64_t qtab_epoch = 0;            // ... computed for this epoch (0: never)
    uint32_t qtab_levels = 0;
    DevBuf blockfill, shards;           // ... records per output block; SX_SHARDS block cursors + SX_SHARDS pair counts
    HostBuf h_shards;
    bool raw_holes = false;             // pairs[0] is block-structured (blockfill, raw_extent slots), not dense
    // what the last expanding scan says about the scene's long ranges (launch_scan_expand's hint): 1 it listed some (mixed-scale:
    // six waves per SIMD), -1 it listed none (the work-list kernel is not launched), 0 nothing known yet (a new layer, forgotten hints)
    int scan_listed = 0;
    DevBuf tree_hist;                   // digit counts of the tree's keys, accumulated by encode for tree_hist_plan ...
    RadixPlan tree_hist_plan{};
    bool tree_hist_valid = false;       // ... while the tree is exactly what that one extend produced (m == tree_hist_m)
    size_t tree_hist_m = 0;
    DevBuf pair_hist;                   // digit counts of the raw stream, accumulated by the scan for pair_hist_plan
    RadixPlan pair_hist_plan{};
    bool pair_hist_valid = false;
    bool owner_hist_valid = false;      // pair_hist row 0 holds the per-owner pair counts of the last shard scan
    size_t raw_extent = 0;
    uint64_t epoch = 1;                 // bumped whenever the sorted tree's content may have changed
    uint64_t ctx_epoch = 0;             // epoch idctx was computed for (0: never), with ...
    uint32_t ctx_thr = 0;               // ... entries of depth < ctx_thr inactive and ...
    uint32_t ctx_lo = 1, ctx_hi = 0;    // ... except those whose id lies in [ctx_lo, ctx_hi] (lo > hi: none) ...
    // shard scans with emit_coarse == 2: coarse entries act as the deeper side iff their id lies in this range
    uint32_t coarse_lo = 1, coarse_hi = 0;
    bool ctx_box = false;               // ... the box codes in use

    bp_stats stats{};
    bool prof_on = false;
    uint32_t grp_off = 0;               // sorts left for which the grouped pair finish stays off (after a refusal)
    uint32_t grp_off_dups = 0;          // the same for the grouped finish of pairs that may repeat
    int grp_next_dups = 0;              // that finish ran last frame on groups of 2^this ids: the scan counts the digits of its plan
    // Pair sort enqueued AHEAD of the scan's sizes (scan_fused_device): what the last frame's sort looked like.  While a frame
    // matches it -- same plan, same id range, about as many entries -- the next scan is followed by the sort and the grouped
    // finish without a host round trip in between: their grids and buffers are sized from these numbers plus a margin and the
    // kernels take the true sizes from device memory (launch_scan_finalize); a frame that outgrows the margin leaves the
    // enqueued kernels idle and runs again on the blocking route.
    struct AheadHint {
        bool valid = false;
        size_t n_raw = 0, extent = 0, m = 0;
        uint32_t n_groups = 0, first_lo = 0, first_hi = 0;
        unsigned long long varying = 0;
        RadixPlan plan{};
    } ahead;
    DevBuf cell_ws;                     // the cell route of the pair sort (CellRoute): cell_base[65537], line / fragment cursors[65536] each
    uint32_t cell_off = 0;              // sorts ahead for which the cell route stays off (after a cell outgrew its range)
    uint32_t cell_tag = 0;              // launch tag of the cell walks' status words (CellRoute::tag)
    bool ahead_try = false;             // scan_device -> scan_fused_device: this frame qualifies
    bool ahead_done = false;            // scan_fused_device -> scan_device: pairs are sorted and finished already
    bool grp_next = true;               // the last distinct-pair sort used the grouped finish (plan hint for the next scan; a new layer expects it: a wrong guess only makes that sort count its own digits)
    // grouped finish of the tree sort: chosen for a sort when the previous sort's group sizes (grp_max of the scalars) allow it
    bool tgrp_ok = false, tgrp_known = false;
    uint32_t tgrp_probe = 0;
    DevBuf tgrp_bounds;
    void* radix_ws_seen = nullptr;      // the radix workspace allocation whose status words carry launch tags (radix_ws_tagged)
    size_t radix_ws_seen_cap = 0;
    uint32_t radix_epoch = 0;
    DevBuf grp_bounds;                  // group start offsets of the grouped finish
    // extend without a host round trip between its kernels: what the previous extend into the empty layer produced
    // predicts what the next one needs
    struct { bool valid = false; size_t n = 0, entries = 0; bool pack = false; uint64_t key_or = 0, key_and = 0; uint32_t depth_mask = 0, id_min = 0, id_max = 0;
             bool ids_distinct = false; } enc_hint;
    // deferred check of the grouped tree sort: the sort is enqueued, its error word and group statistics are read at the
    // NEXT host round trip (the scan's) instead of one of their own
    // copied: the sort's scalars are on their way to h_sort_sc (a deferred check leaves that to the fetch it rides on: sort_sc_fetch)
    struct { bool on = false; int src = 0; bool ride = false; bool cells = false; bool copied = false; bool compact = false; } sort_pend;
    // cell route of the grouped tree sort (CellRoute with a payload): laid out from the group bounds the previous sort of the same
    // plan left in tgrp_bounds; its gapped result goes to tcell_keys / tcell_ids, the finish kernel writes the dense tree
    struct { bool valid = false; unsigned long long g0 = 0; uint32_t n_groups = 0; int gshift = 0; size_t m = 0; } tcell;
    DevBuf tcell_keys, tcell_ids, tcell_ws;
    uint32_t tcell_off = 0, tcell_tag = 0;
    // a partition pass that was only enqueued (BP_PARTITION_ENQUEUE_ONLY): its error word, copied behind the pass into
    // h_part_err, is looked at by the layer's next host round trip (sync_stream)
    bool part_pend = false;
    DevBuf d_part_err;
    HostBuf h_part_err;
    DevBuf d_sort_sc;                   // DevScalars of the tree sort's kernels (the scan re-initialises d_sc)
    HostBuf h_sort_sc;
    int filt_shape = FILT_AABB;         // row shape of the built-in narrow phase during scan_filtered_*
    uint32_t prof_mask = 0xFFFFFFFFu;   // kernel classes that get event pairs while prof_on
    uint32_t prof_every = 1, prof_frame = 0;  // event pairs only in every prof_every-th frame (frames are counted by bp_layer_clear)
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> prof_pool;
    int prof_open = -1;
    hipEvent_t prof_open_ev = nullptr;
    double prof_ms[BP_K_COUNT] = {0};
    uint64_t prof_n[BP_K_COUNT] = {0};

    std::string err;
};

namespace {

thread_local std::string g_last_global_error;

int fail(bp_layer* L, int code, const char* what, hipError_t e = hipSuccess) {
    char buf[512];
    if (e != hipSuccess) {
        snprintf(buf, sizeof buf, "%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
        (void)hipGetLastError();  // reported here: must not resurface in the next call's launch check
    } else
        snprintf(buf, sizeof buf, "%s", what);
    if (L) L->err = buf;
    g_last_global_error = buf;
    return code;
}

#define HIP_TRY(L, call)                                                    \
    do {                                                                    \
        hipError_t e__ = (call);                                            \
        if (e__ != hipSuccess) return fail((L), e__ == hipErrorOutOfMemory ? BP_ERR_OOM : BP_ERR_HIP, #call, e__); \
    } while (0)

#define BP_TRY(call)               \
    do {                           \
        int rc__ = (call);         \
        if (rc__ != BP_OK) return rc__; \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() {
        int now = -1;
        if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev);
    }
};

// grow-only device buffer; preserve keeps the first keep_bytes bytes
int ensure_dev(bp_layer* L, DevBuf& b, size_t bytes, size_t keep_bytes = 0) {
    if (bytes <= b.cap) return BP_OK;
    // (1/16 of head room on top of the request: a scene whose sizes drift by a fraction of a percent from frame to frame -- a frame
    // loop that moves its objects -- must not outgrow an exactly-sized first allocation one frame later: hipFree + hipMalloc of a
    // gigabyte is 0.25 ms, seen as one slow frame in bench.py's moving record)
    static const bool exact = getenv("BP_ALLOC_EXACT") != nullptr;  // A/B switch: no head room
    size_t want = std::max(exact ? bytes : bytes + bytes / 16, b.cap + b.cap / 2);
    want = (want + 255) / 256 * 256;
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess) {
        // retry with the exact size before giving up
        want = (bytes + 255) / 256 * 256;
        e = hipMalloc(&q, want);
        if (e != hipSuccess) return fail(L, BP_ERR_OOM, "hipMalloc", e);
    }
    // a side buffer (box codes) can be shorter than the tree it belongs to while its contents are
    // not in use: only what the old block actually holds can be carried over
    keep_bytes = std::min(keep_bytes, b.cap);
    if (keep_bytes && b.p) {
        e = hipMemcpyAsync(q, b.p, keep_bytes, hipMemcpyDeviceToDevice, L->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(L->stream);
        if (e != hipSuccess) { (void)hipFree(q); return fail(L, BP_ERR_HIP, "grow copy", e); }
    } else if (b.p) {
        e = hipStreamSynchronize(L->stream);  // nothing in flight may still use the old block
        if (e != hipSuccess) { (void)hipFree(q); return fail(L, BP_ERR_HIP, "hipStreamSynchronize", e); }
    }
    if (b.p) (void)hipFree(b.p);
    b.p = q;
    b.cap = want;
    return BP_OK;
}

int ensure_host(bp_layer* L, HostBuf& b, size_t bytes) {
    if (bytes <= b.cap) return BP_OK;
    size_t want = std::max(bytes, b.cap + b.cap / 2);
    void* q = nullptr;
    hipError_t e = hipHostMalloc(&q, want, hipHostMallocDefault);
    if (e != hipSuccess) return fail(L, BP_ERR_OOM, "hipHostMalloc", e);
    if (b.p) (void)hipHostFree(b.p);
    b.p = q;
    b.cap = want;
    return BP_OK;
}

void free_dev(DevBuf& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
void free_host(HostBuf& b) { if (b.p) (void)hipHostFree(b.p); b.p = nullptr; b.cap = 0; }

// ---- profiling: HIP events on the layer's own stream ---------------------------------------------
hipEvent_t prof_event(bp_layer* L) {
    if (!L->prof_pool.empty()) {
        hipEvent_t e = L->prof_pool.back();
        L->prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
// cls >= 0 opens a timed region of that class, cls < 0 closes the open one
void prof_mark(void* ctx, int cls) {
    bp_layer* L = (bp_layer*)ctx;
    if (!L->prof_on) return;
    if (cls >= 0) {
        if (!((L->prof_mask >> cls) & 1u)) return;  // class not selected: no events, no stream bubbles
        if (L->prof_every > 1 && (L->prof_frame % L->prof_every) != 0) return;  // not a sampled frame
        L->prof_open = cls;
        L->prof_open_ev = prof_event(L);
        if (L->prof_open_ev) (void)hipEventRecord(L->prof_open_ev, L->stream);
    } else if (L->prof_open >= 0) {
        hipEvent_t b = prof_event(L);
        if (b && L->prof_open_ev) {
            (void)hipEventRecord(b, L->stream);
            L->prof_pending.push_back({L->prof_open, L->prof_open_ev, b});
        }
        L->prof_open = -1;
        L->prof_open_ev = nullptr;
    }
}
void prof_collect(bp_layer* L) {  // call after the stream has been synchronised
    for (auto& r : L->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            L->prof_ms[r.cls] += ms;
            L->prof_n[r.cls] += 1;
        }
        L->prof_pool.push_back(r.a);
        L->prof_pool.push_back(r.b);
    }
    L->prof_pending.clear();
}
struct ProfScope {
    bp_layer* L;
    ProfScope(bp_layer* l, int cls) : L(l) { prof_mark(L, cls); }
    ~ProfScope() { prof_mark(L, -1); }
};

int after_sync(bp_layer* L);
int sync_stream(bp_layer* L) {
    ++L->stats.host_syncs;
    HIP_TRY(L, hipStreamSynchronize(L->stream));
    return after_sync(L);
}
// what follows every completed wait for the stream (also a wait on a published flag behind everything enqueued)
int after_sync(bp_layer* L) {
    prof_collect(L);
    if (L->part_pend) {  // an enqueue-only partition pass has completed by now: its error word
        L->part_pend = false;
        const unsigned err = *(const unsigned*)L->h_part_err.p;
        if (err & KERR_LOOKBACK_TIMEOUT) return fail(L, BP_ERR_KERNEL, "radix look-back timed out in the entry partition pass (its buckets were incomplete)");
    }
    return BP_OK;
}

// ---- results published by kernels into pinned host memory (struct Published) -------------------------------
constexpr size_t PUB_FLAG_OFF = 4096, PUB_BYTES = PUB_FLAG_OFF + 64, PUB_SORT_OFF = 3072;  // (a second block: the tree sort's scalars)
int pub_next(bp_layer* L, Published* out) {
    if (!L->pub) {
        void* q = nullptr;
        hipError_t e = hipHostMalloc(&q, PUB_BYTES, hipHostMallocMapped | hipHostMallocCoherent);
        if (e != hipSuccess) return fail(L, BP_ERR_OOM, "hipHostMalloc", e);
        std::memset(q, 0, PUB_BYTES);
        L->pub = q;
        L->pub_seq = 0;
    }
    if (++L->pub_seq == 0) L->pub_seq = 1;  // (0 is the flag's initial value)
    void* dev = nullptr;
    HIP_TRY(L, hipHostGetDevicePointer(&dev, L->pub, 0));
    out->dst = dev;
    out->flag = (unsigned int*)((char*)dev + PUB_FLAG_OFF);
    out->seq = L->pub_seq;
    return BP_OK;
}
// spins until the launch `seq` has published (a host round trip like any other: counted); the stream is looked at now and then, so
// that a failed launch or a device error ends the wait
int pub_wait(bp_layer* L, uint32_t seq) {
    ++L->stats.host_syncs;
    const unsigned int* flag = (const unsigned int*)((const char*)L->pub + PUB_FLAG_OFF);
    for (unsigned long long spins = 1;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return BP_OK;
        if ((spins & 0x3FFFull) == 0) {
            const hipError_t q = hipStreamQuery(L->stream);
            if (q == hipSuccess) {  // everything has run: the flag is there, or the publishing launch never happened
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return BP_OK;
                return fail(L, BP_ERR_KERNEL, "a kernel's published results did not arrive");
            }
            if (q != hipErrorNotReady) return fail(L, BP_ERR_HIP, "hipStreamQuery", q);
        }
        __builtin_ia32_pause();
    }
}

// ---- device scalars ---------------------------------------------------------------------------------
DevScalars neutral_scalars() {  // neutral elements for every reduction
    DevScalars n;
    std::memset(&n, 0, sizeof n);
    n.key_and = ~0ull;
    n.id_and = 0xFFFFFFFFu;
    n.id_min = 0xFFFFFFFFu;
    return n;
}
// a tree sort whose check was deferred: its scalars join the D2H traffic of the round trip that settles it
int sort_sc_fetch(bp_layer* L) {
    if (L->sort_pend.on && !L->sort_pend.copied) {
        HIP_TRY(L, hipMemcpyAsync(L->h_sort_sc.p, L->d_sort_sc.p, sizeof(DevScalars), hipMemcpyDeviceToHost, L->stream));
        L->sort_pend.copied = true;
    }
    return BP_OK;
}
int sc_reset(bp_layer* L) {
    DevScalars* h = (DevScalars*)L->h_sc.p;
    *h = neutral_scalars();
    HIP_TRY(L, hipMemcpyAsync(L->d_sc.p, h, sizeof(DevScalars), hipMemcpyHostToDevice, L->stream));
    return BP_OK;
}
// the scalars' device / host buffers carry the expanding scan's shard cursors behind them, so that ONE copy brings both back
// (and, behind those, the device-side sizes of a scan whose sort was enqueued ahead: SIZES_WORDS words)
constexpr size_t SC_SHARDS_OFF = 256, SC_SIZES_OFF = SC_SHARDS_OFF + 3 * SX_SHARDS * sizeof(uint64_t),
                 SC_BLOCK_BYTES = SC_SIZES_OFF + SIZES_WORDS * sizeof(uint32_t);
static_assert(sizeof(DevScalars) <= SC_SHARDS_OFF, "scalars block layout");
uint32_t* sc_sizes_dev(bp_layer* L) { return (uint32_t*)((char*)L->d_sc.p + SC_SIZES_OFF); }
const uint32_t* sc_sizes_host(bp_layer* L) { return (const uint32_t*)((const char*)L->h_sc.p + SC_SIZES_OFF); }
uint64_t* sc_shards_dev(bp_layer* L) { return (uint64_t*)((char*)L->d_sc.p + SC_SHARDS_OFF); }
const uint64_t* sc_shards_host(bp_layer* L) { return (const uint64_t*)((const char*)L->h_sc.p + SC_SHARDS_OFF); }
int sc_fetch(bp_layer* L, DevScalars* out, bool with_shards = false) {  // D2H + sync
    BP_TRY(sort_sc_fetch(L));
    HIP_TRY(L, hipMemcpyAsync(L->h_sc.p, L->d_sc.p, with_shards ? SC_SIZES_OFF : sizeof(DevScalars), hipMemcpyDeviceToHost, L->stream));
    BP_TRY(sync_stream(L));
    *out = *(DevScalars*)L->h_sc.p;
    if (out->err & KERR_LOOKBACK_TIMEOUT) return fail(L, BP_ERR_KERNEL, "radix look-back timed out");
    if (out->err & KERR_COUNT_OVERFLOW)
        return fail(L, BP_ERR_TOO_LARGE, "an object expands to more than 2^20 cells (min_depth too high for its size)");
    return BP_OK;
}

// the radix workspace with launch-tagged status words (radix_sort's epoch_io): zero-filled once per allocation
int radix_ws_tagged(bp_layer* L, size_t bytes, uint32_t** epoch) {
    BP_TRY(ensure_dev(L, L->radix_ws, bytes));
    if (L->radix_ws.p != L->radix_ws_seen || L->radix_ws.cap != L->radix_ws_seen_cap) {
        HIP_TRY(L, hipMemsetAsync(L->radix_ws.p, 0, L->radix_ws.cap, L->stream));
        L->radix_ws_seen = L->radix_ws.p;
        L->radix_ws_seen_cap = L->radix_ws.cap;
        // test switch: start the launch counter just below the point where the 30-bit tags wrap (tests/test_gpu_parity.py)
        static const char* start = getenv("BP_RADIX_EPOCH_START");
        L->radix_epoch = start ? (uint32_t)strtoul(start, nullptr, 0) : 0u;
    }
    *epoch = &L->radix_epoch;
    return BP_OK;
}

bool is_device_ptr(const void* p) {
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof a);
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain host memory is reported as an error by some runtimes
        return false;
    }
    return a.type == hipMemoryTypeDevice;
}

void reset_tree_facts(bp_layer* L) {
    L->have_ids = false;
    L->order_ok = true;
    L->ids_distinct = true;
    L->max_id = 0;
    L->min_id = 0xFFFFFFFFu;
    L->key_or = 0; L->key_and = ~0ull;
    L->id_or = 0; L->id_and = 0xFFFFFFFFu;
    L->depth_mask = 0;
    L->box_valid = true;
    L->box_all_known = true;
    L->ids_packed = false;
    L->tree_compact = false;
    L->ids_lazy = false;
}

uint64_t* cur_keys(bp_layer* L) { return (uint64_t*)L->keys[L->cur].p; }
uint32_t* cur_ids(bp_layer* L) { return (uint32_t*)L->ids[L->cur].p; }

int ensure_tree_capacity(bp_layer* L, size_t entries) {
    BP_TRY(ensure_dev(L, L->keys[L->cur], entries * sizeof(uint64_t), L->m * sizeof(uint64_t)));
    BP_TRY(ensure_dev(L, L->ids[L->cur], entries * sizeof(uint32_t), L->m * sizeof(uint32_t)));
    BP_TRY(ensure_dev(L, L->box[L->cur], entries, L->m));
    return BP_OK;
}

int settle_sort(bp_layer* L, bool already_synced = false, bool* redone = nullptr);
// compact records -> index values in keys[buf], ids in ids[buf], box bytes in box[buf] (the form a non-packing extend leaves)
int expand_compact(bp_layer* L, int buf, size_t m) {
    BP_TRY(ensure_dev(L, L->ids[buf], std::max<size_t>(m, 1) * sizeof(uint32_t)));
    BP_TRY(ensure_dev(L, L->box[buf], std::max<size_t>(m, 1)));
    HIP_TRY(L, launch_tree_expand(L->kind, (uint64_t*)L->keys[buf].p, (uint32_t*)L->ids[buf].p, (uint8_t*)L->box[buf].p, m, L->cfmt, L->stream));
    ++L->stats.tree_expands;
    return BP_OK;
}
// ids / box bytes a compact sort left unwritten: from the context words (which that sort wrote for exactly this order)
int ensure_ids(bp_layer* L) {
    if (!L->ids_lazy) return BP_OK;
    if (L->ctx_epoch != L->epoch) return fail(L, BP_ERR_KERNEL, "internal: the context words a lazy id array depends on are stale");
    BP_TRY(ensure_dev(L, L->ids[L->cur], std::max<size_t>(L->m, 1) * sizeof(uint32_t)));
    BP_TRY(ensure_dev(L, L->box[L->cur], std::max<size_t>(L->m, 1)));
    HIP_TRY(L, launch_ids_from_ctx((const uint64_t*)L->idctx.p, (uint32_t*)L->ids[L->cur].p, (uint8_t*)L->box[L->cur].p, L->m, L->stream));
    L->ids_lazy = false;
    return BP_OK;
}
// a tree that an extend left as compact records is expanded before anything but the compact sort reads it
int ensure_plain(bp_layer* L) {
    if (!L->tree_compact) return BP_OK;
    BP_TRY(expand_compact(L, L->cur, L->m));
    L->tree_compact = false;
    L->ids_packed = false;
    L->box_valid = true;
    L->tree_hist_valid = false;  // (the digit counts were the composite's)
    return BP_OK;
}
// ids committed with their box codes still packed are split before anything but the sort reads them; a tree sort whose
// check was deferred is settled first (every accessor of the tree comes through here)
int ensure_unpacked(bp_layer* L) {
    if (L->sort_pend.on) BP_TRY(settle_sort(L));
    BP_TRY(ensure_plain(L));
    BP_TRY(ensure_ids(L));
    if (!L->ids_packed) return BP_OK;
    if (L->m) {
        BP_TRY(ensure_dev(L, L->box[L->cur], L->m));
        HIP_TRY(L, launch_split_box(cur_ids(L), (uint8_t*)L->box[L->cur].p, L->m, L->stream));
        BP_TRY(sync_stream(L));
    }
    L->ids_packed = false;
    L->box_valid = true;
    return BP_OK;
}

bool tree_partial_plan(size_t m, unsigned long long varying, RadixPlan* plan, int* seg_shift_out);
// will the next tree sort take the grouped plan where it applies?  Yes when the last sort's group sizes allowed it -- and, nothing
// being known yet, on a layer's first sort (sort_tree: optimistic; BP_NO_OPTIMISTIC_FIRST_SORT switches that off)
static bool tgrp_expected(const bp_layer* L);
struct TreeGroupPlan { RadixPlan plan; int gshift; uint32_t n_groups; unsigned long long g0; };
bool tree_group_plan(size_t m, unsigned long long key_and, unsigned long long key_or, TreeGroupPlan* out);
// ---- compact route of the per-frame tree sort (TreeCompact, bp_kernels.h) ------------------------------------------------------
// would the grouped tree sort of m entries under plan gp take the cell route (what sort_tree checks before its radix passes)?
// before_sort: asked ahead of that sort (its back-off counter has not been stepped yet)
bool tree_cells_possible(const bp_layer* L, size_t m, const TreeGroupPlan& gp, bool before_sort) {
    static const bool no_tcells = getenv("BP_NO_TREE_CELL_SORT") != nullptr;  // A/B switch
    const uint32_t cell0 = (uint32_t)(gp.g0 & 0xFFFFu);
    const bool off_ok = before_sort ? L->tcell_off <= 1 : L->tcell_off == 0;
    return !no_tcells && off_ok && L->tcell.valid && L->tcell.g0 == gp.g0 && L->tcell.n_groups == gp.n_groups && L->tcell.gshift == gp.gshift &&
           gp.plan.n_passes == 2 && (size_t)cell0 + gp.n_groups <= CELL_COUNT && radix_claim_applies(m) && m <= L->tcell.m + L->tcell.m / 16 &&
           m + m / 16 >= L->tcell.m;
}
// the grouped plan gp (shifts in the index value) restated for composite words of format fmt: the groups are the same -- the
// group shift must lie inside the Morton bits the format keeps
struct CompactPlan { TreeCompact fmt; RadixPlan plan; int comp_gshift; };
bool compact_plan(const TreeGroupPlan& gp, const TreeCompact& fmt, CompactPlan* out) {
    if (gp.gshift < fmt.pshift || gp.plan.n_passes != 2) return false;
    const int cg = gp.gshift - fmt.pshift + fmt.tbits + fmt.low;
    if (!tgroup_compact_applies(cg) || cg + 8 >= 64) return false;
    out->fmt = fmt;
    out->comp_gshift = cg;
    out->plan.n_passes = 2;
    for (int i = 0; i < RS_MAX_PASSES; ++i) { out->plan.shift[i] = 0; out->plan.from_val[i] = 0; }
    out->plan.shift[0] = cg; out->plan.shift[1] = cg + 8;
    return true;
}
// the format an extend chooses from what the PREVIOUS frame looked like (depths present, id bits), with a level of head room
// on either side where 64 bits allow it -- a frame that leaves the format all the same runs again on the plain path
bool compact_choose(int kind, uint32_t depth_mask, uint32_t id_min, uint32_t id_max, const TreeGroupPlan& gp, CompactPlan* out) {
    static const bool disabled = getenv("BP_NO_COMPACT_TREE") != nullptr;  // A/B switch
    if (disabled || depth_mask == 0 || id_max < id_min) return false;
    const KindInfo ki = kind_info(kind);
    const uint32_t dlo = (uint32_t)__builtin_ctz(depth_mask), dhi = 31u - (uint32_t)__builtin_clz(depth_mask);
    // (the id field holds id - id_min: ids of any magnitude, as long as their range leaves room -- also ids >= 2^27, which cannot
    // carry their box codes through the (index, id) sort and used to cost the scan its duplicate suppression)
    const int idbits = std::max(1, 32 - __builtin_clz((id_max - id_min) | 1u));
    if (idbits > 27) return false;
    // (head room on the deep side only: a level below the shallowest depth would lengthen the digit tail -- a few instructions per
    // entry in the encode and the finish -- and never costs fewer bits)
    const uint32_t hi_m = std::min(dhi + 1u, ki.axis_bits);
    const uint32_t tries[2][2] = {{dlo, hi_m}, {dlo, dhi}};
    for (const auto& t : tries) {
        TreeCompact f;
        if (tree_compact_format(kind, t[0], t[1], idbits, &f)) {
            f.id_base = id_min;
            if (compact_plan(gp, f, out)) return true;
        }
    }
    return false;
}


// ---- extend -------------------------------------------------------------------------------------------
// Extend into the EMPTY layer (the per-frame pattern) when an earlier extend left a prediction: count, block sums and
// emit are enqueued back to back and the host reads the outcome ONCE (the plain path: a round trip after the count
// pass for the entry total, the id facts and the digit plan, and one after the emit pass).  Predicted: entry capacity
// (guarded on the device: encode_emit writes nothing beyond it), whether the box codes may ride in the ids, the digit
// plan of the coming tree sort.  *done = 0: not applicable or a prediction failed -- nothing of the layer's state was
// touched and the caller runs the plain path.
int extend_predicted(bp_layer* L, const SysBounds& sb, const float* d_bounds, const uint32_t* d_ids, size_t n, int* done) {
    *done = 0;
    const auto& h = L->enc_hint;
    // entries scale with the object count; 3 % + a little head room, and never less than what is allocated anyway
    const double scale = h.n ? (double)n / (double)h.n : 1.0;
    const size_t pred_m = n == h.n ? h.entries : (size_t)((double)h.entries * scale);
    if (pred_m == 0) return BP_OK;
    const size_t want = pred_m + pred_m / 32 + 8192;
    BP_TRY(ensure_tree_capacity(L, want));
    const bool pack = h.pack;
    size_t capacity = std::min(L->keys[L->cur].cap / sizeof(uint64_t), L->ids[L->cur].cap / sizeof(uint32_t));
    if (!pack) capacity = std::min(capacity, L->box[L->cur].cap);
    const size_t nb = enc_blocks(n);
    BP_TRY(ensure_dev(L, L->counts, n * ENC_REC_BYTES));
    BP_TRY(ensure_dev(L, L->block_sums, nb * sizeof(unsigned long long)));
    BP_TRY(ensure_dev(L, L->aux, nb * sizeof(BlockAux)));
    // the digit plan the coming tree sort will most likely take (it verifies: a different plan only costs its own histogram)
    RadixPlan hist_plan;
    int seg_shift_unused = 0;
    bool count_digits = tree_partial_plan(pred_m, h.key_or ^ h.key_and, &hist_plan, &seg_shift_unused) && hist_plan.n_passes <= SX_HIST_MAX;
    const KindInfo kic = kind_info(L->kind);
    const bool count_cells = L->cell_fuse_depth > 0 && kic.dim * L->cell_fuse_depth <= 9;
    // compact records: when the coming sort will most likely take the cell route and the previous frame's depths and ids leave
    // room in 64 bits, the emit pass writes one composite word per entry and counts the composite's digits
    bool compact = false;
    CompactPlan cp{};
    {
        TreeGroupPlan gp;
        if (L->tgrp_ok && tree_group_plan(pred_m, h.key_and, h.key_or, &gp) && gp.plan.n_passes <= SX_HIST_MAX) {
            hist_plan = gp.plan;
            count_digits = true;
            if ((pack || h.ids_distinct) && !count_cells && tree_cells_possible(L, pred_m, gp, /*before_sort=*/true) &&
                compact_choose(L->kind, h.depth_mask, h.id_min, h.id_max, gp, &cp)) {
                compact = true;
                hist_plan = cp.plan;
            }
        }
    }
    const size_t cell_bytes = count_cells ? (size_t)HIST_COPIES * (((size_t)1 << (kic.dim * L->cell_fuse_depth)) + 1) * sizeof(unsigned long long) : 0;
    // the histograms encode_emit adds into and the scalars are reset by scan_block_sums (between count and emit): no fills, no H2D copy
    if (count_cells) BP_TRY(ensure_dev(L, L->cell_hist_dev, cell_bytes));
    if (count_digits) BP_TRY(ensure_dev(L, L->tree_hist, (size_t)HIST_COPIES * SX_HIST_MAX * RS_BINS * sizeof(unsigned long long)));
    L->tree_hist_valid = false; L->cell_hist_valid = false; L->cell_dev_valid = false;
    {
        ProfScope ps(L, BP_K_ENCODE);
        // (the histograms the emit pass adds into are zero-filled by the count pass, the scalars reset by scan_block_sums: no fills, no H2D copy)
        HIP_TRY(L, launch_encode_count(L->kind, sb, d_bounds, d_ids, n, L->min_depth, 0, 0u, L->counts.p,
                                       (unsigned long long*)L->block_sums.p, (BlockAux*)L->aux.p, (DevScalars*)L->d_sc.p, L->stream,
                                       count_digits ? (unsigned long long*)L->tree_hist.p : nullptr,
                                       (size_t)HIST_COPIES * SX_HIST_MAX * RS_BINS,
                                       count_cells ? (unsigned long long*)L->cell_hist_dev.p : nullptr, cell_bytes / sizeof(unsigned long long)));
    }
    static const bool no_early = getenv("BP_NO_EARLY_FETCH") != nullptr;  // A/B switches
    static const bool no_publish = getenv("BP_NO_PUBLISH") != nullptr;
    const bool early = !no_early && !count_cells;
    Published early_pub{};
    if (early && !no_publish) BP_TRY(pub_next(L, &early_pub));
    {
        ProfScope ps(L, BP_K_MISC);
        const DevScalars init = neutral_scalars();
        HIP_TRY(L, launch_scan_block_sums((unsigned long long*)L->block_sums.p, nb, (const BlockAux*)L->aux.p, (DevScalars*)L->d_sc.p, L->stream,
                                          &init, nullptr, 0, nullptr, 0, early_pub.dst ? &early_pub : nullptr));
    }
    // The entry total and the facts are final once the block sums are scanned: their copy is enqueued BEFORE encode_emit and
    // the host waits for that copy alone (an event), while encode_emit is already running behind it -- the round trip costs
    // the GPU no idle time, and the sort's kernels queue up behind the emit pass.  (The emit pass's own capacity guard is
    // evaluated on the host from the same total.)  Better still, scan_block_sums PUBLISHES them (struct Published): no copy sits
    // between the two kernels (a D2H copy + event cost the stream 6 - 13 us there) and the host polls memory.
    if (early && !early_pub.dst) {
        if (!L->early_ev && hipEventCreateWithFlags(&L->early_ev, hipEventDisableTiming) != hipSuccess) return fail(L, BP_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(L, hipMemcpyAsync(L->h_sc.p, L->d_sc.p, sizeof(DevScalars), hipMemcpyDeviceToHost, L->stream));
        HIP_TRY(L, hipEventRecord(L->early_ev, L->stream));
    }
    {
        ProfScope ps(L, BP_K_ENCODE);
        HIP_TRY(L, launch_encode_emit(L->kind, sb, d_bounds, d_ids, n, L->min_depth, L->counts.p,
                                      (const unsigned long long*)L->block_sums.p, cur_keys(L), cur_ids(L),
                                      pack ? nullptr : (uint8_t*)L->box[L->cur].p, L->stream, count_digits ? &hist_plan : nullptr,
                                      count_digits ? (unsigned long long*)L->tree_hist.p : nullptr, pack ? 1 : 0,
                                      count_cells ? (unsigned long long*)L->cell_hist_dev.p : nullptr, L->cell_fuse_depth,
                                      (DevScalars*)L->d_sc.p, capacity, compact ? &cp.fmt : nullptr));
    }
    // the cell histogram counted on the way (multi-GPU splitter): summed on the device and brought back by THIS round trip,
    // so that bp_layer_cell_histogram costs none of its own
    const uint32_t n_cellbins = count_cells ? (1u << (kic.dim * L->cell_fuse_depth)) + 1u : 0u;
    if (count_cells) {
        BP_TRY(ensure_dev(L, L->cell_sum, n_cellbins * sizeof(unsigned long long)));
        BP_TRY(ensure_host(L, L->h_cells, n_cellbins * sizeof(unsigned long long)));
        HIP_TRY(L, launch_sum_copies((const unsigned long long*)L->cell_hist_dev.p, HIST_COPIES, n_cellbins, (unsigned long long*)L->cell_sum.p, L->stream));
        HIP_TRY(L, hipMemcpyAsync(L->h_cells.p, L->cell_sum.p, n_cellbins * sizeof(unsigned long long), hipMemcpyDeviceToHost, L->stream));
    }
    DevScalars sc;
    if (early) {
        if (early_pub.dst) {
            BP_TRY(pub_wait(L, early_pub.seq));  // a host round trip all the same (the GPU stays busy through it)
            sc = *(const DevScalars*)L->pub;
        } else {
            ++L->stats.host_syncs;
            HIP_TRY(L, hipEventSynchronize(L->early_ev));
            sc = *(const DevScalars*)L->h_sc.p;
        }
        if (sc.err & KERR_COUNT_OVERFLOW)
            return fail(L, BP_ERR_TOO_LARGE, "an object expands to more than 2^20 cells (min_depth too high for its size)");
        if ((size_t)sc.total > capacity) sc.err |= KERR_ENCODE_OVERFLOW;  // what the emit pass's guard concludes from the same total
    } else {
        BP_TRY(sc_fetch(L, &sc));
    }
    const size_t add = (size_t)sc.total;
    const bool can_pack = !(sc.flags & EXT_IDS_REPEAT) && (sc.id_or >> BOX_ID_BITS) == 0;
    // (compact records: every depth and id of THIS frame must lie inside the format chosen from the last one)
    const bool compact_fits = !compact || add == 0 ||
                              (sc.depth_mask != 0 && (uint32_t)__builtin_ctz(sc.depth_mask) >= cp.fmt.dmin &&
                               31u - (uint32_t)__builtin_clz(sc.depth_mask) <= cp.fmt.dmax && !(sc.flags & EXT_IDS_REPEAT) &&
                               sc.id_min >= cp.fmt.id_base && ((sc.id_max - cp.fmt.id_base) >> cp.fmt.idbits) == 0);
    if ((sc.err & KERR_ENCODE_OVERFLOW) || (pack && !can_pack) || !compact_fits) {
        // the prediction failed: remember the truth for the next frame and redo this one on the plain path
        L->enc_hint.entries = add; L->enc_hint.n = n; L->enc_hint.pack = can_pack;
        L->enc_hint.depth_mask = sc.depth_mask; L->enc_hint.id_min = sc.id_min; L->enc_hint.id_max = sc.id_max;
        L->enc_hint.ids_distinct = !(sc.flags & EXT_IDS_REPEAT);
        ++L->stats.extend_reruns;
        return BP_OK;
    }
    L->stats.objects_invalid += sc.n_invalid;
    L->stats.objects_valid += n - sc.n_invalid;
    if (sc.flags & EXT_IDS_DECREASE) L->order_ok = false;
    if (sc.flags & EXT_IDS_REPEAT) L->ids_distinct = false;
    if (sc.flags & EXT_BOX_UNKNOWN) L->box_all_known = false;
    L->have_ids = true;
    L->max_id = std::max(L->max_id, sc.id_max);
    L->min_id = std::min(L->min_id, sc.id_min);
    *done = 1;
    L->enc_hint.valid = true; L->enc_hint.n = n; L->enc_hint.entries = add;
    static const bool no_emit_pack = getenv("BP_NO_EMIT_PACK") != nullptr;
    L->enc_hint.pack = can_pack && !no_emit_pack;
    L->enc_hint.key_or = sc.key_or; L->enc_hint.key_and = sc.key_and;
    L->enc_hint.depth_mask = sc.depth_mask; L->enc_hint.id_min = sc.id_min; L->enc_hint.id_max = sc.id_max;
    L->enc_hint.ids_distinct = !(sc.flags & EXT_IDS_REPEAT);
    if (add == 0) return BP_OK;
    if (count_cells) {
        L->cell_dev_valid = true;
        const unsigned long long* hc = (const unsigned long long*)L->h_cells.p;
        L->cell_hist_cache.assign(hc, hc + n_cellbins);
        L->cell_hist_depth = L->cell_fuse_depth;
        L->cell_hist_valid = true;
    }
    if (compact) { L->tree_compact = true; L->cfmt = cp.fmt; L->box_valid = false; }
    else if (pack) { L->ids_packed = true; L->box_valid = false; }
    L->tail_run = false;
    if (count_digits) { L->tree_hist_valid = true; L->tree_hist_plan = hist_plan; L->tree_hist_m = add; }
    L->m += add;
    L->sorted = false;
    L->key_or |= sc.key_or; L->key_and &= sc.key_and;
    L->id_or |= sc.id_or; L->id_and &= sc.id_and;
    L->depth_mask |= sc.depth_mask;
    L->stats.entries = L->m;
    return BP_OK;
}

int extend_device(bp_layer* L, const float* sys_min, const float* sys_max, const float* d_bounds,
                  const uint32_t* d_ids, size_t n) {
    if (n == 0) return BP_OK;
    if (n > 0x7FFFFFFFull) return fail(L, BP_ERR_TOO_LARGE, "extend: more than 2^31-1 objects in one call");
    const KindInfo ki = kind_info(L->kind);
    SysBounds sb;
    for (uint32_t a = 0; a < 3; ++a) {
        sb.mn[a] = a < ki.dim ? sys_min[a] : 0.f;
        sb.mx[a] = a < ki.dim ? sys_max[a] : 0.f;
    }
    if (L->m) BP_TRY(ensure_unpacked(L));  // appending to a tree whose ids still carry their box codes
    static const bool no_predicted = getenv("BP_NO_PREDICTED_EXTEND") != nullptr;  // A/B switch
    const bool was_empty = L->m == 0 && !L->have_ids;
    if (!no_predicted && was_empty && L->enc_hint.valid && L->box_valid && L->ids_distinct) {
        int done = 0;
        BP_TRY(extend_predicted(L, sb, d_bounds, d_ids, n, &done));
        if (done) return BP_OK;
    }
    const size_t nb = enc_blocks(n);
    BP_TRY(ensure_dev(L, L->counts, n * ENC_REC_BYTES));
    BP_TRY(ensure_dev(L, L->block_sums, nb * sizeof(unsigned long long)));
    BP_TRY(ensure_dev(L, L->aux, nb * sizeof(BlockAux)));
    BP_TRY(sc_reset(L));
    {
        ProfScope ps(L, BP_K_ENCODE);
        HIP_TRY(L, launch_encode_count(L->kind, sb, d_bounds, d_ids, n, L->min_depth, L->have_ids ? 1 : 0, L->max_id,
                                       L->counts.p, (unsigned long long*)L->block_sums.p,
                                       (BlockAux*)L->aux.p, (DevScalars*)L->d_sc.p, L->stream));
    }
    {
        ProfScope ps(L, BP_K_MISC);
        HIP_TRY(L, launch_scan_block_sums((unsigned long long*)L->block_sums.p, nb, (const BlockAux*)L->aux.p,
                                          (DevScalars*)L->d_sc.p, L->stream));
    }
    DevScalars sc;
    BP_TRY(sc_fetch(L, &sc));
    const size_t add = (size_t)sc.total;
    L->stats.objects_invalid += sc.n_invalid;
    L->stats.objects_valid += n - sc.n_invalid;
    // ordering facts are judged on every id handed in, valid or not (conservative)
    if (sc.flags & EXT_IDS_DECREASE) L->order_ok = false;
    if (sc.flags & EXT_IDS_REPEAT) L->ids_distinct = false;
    if (sc.flags & EXT_BOX_UNKNOWN) L->box_all_known = false;
    L->have_ids = true;
    L->max_id = std::max(L->max_id, sc.id_max);
    L->min_id = std::min(L->min_id, sc.id_min);
    if (add == 0) return BP_OK;
    BP_TRY(ensure_tree_capacity(L, L->m + add));
    // first extend into an empty tree (the per-frame pattern): the plan of the coming tree sort is
    // already known, so encode counts its digit histograms while the keys are in registers
    RadixPlan hist_plan;
    int seg_shift_unused = 0;
    bool count_digits = L->m == 0 && tree_partial_plan(add, sc.key_or ^ sc.key_and, &hist_plan, &seg_shift_unused) &&
                        hist_plan.n_passes <= SX_HIST_MAX;
    {   // the coming sort takes the grouped plan when the previous sort's group sizes allowed it: count THOSE digits
        TreeGroupPlan gp;
        if (L->m == 0 && tgrp_expected(L) && tree_group_plan(add, sc.key_and, sc.key_or, &gp) && gp.plan.n_passes <= SX_HIST_MAX) {
            hist_plan = gp.plan;
            count_digits = true;
        }
    }
    L->tree_hist_valid = false; L->cell_hist_valid = false; L->cell_dev_valid = false;
    // same situation: the codes can ride in the ids from the start (ids distinct and < 2^27), so the tree
    // sort's first pass has no box bytes to gather and none are written here
    static const bool no_emit_pack = getenv("BP_NO_EMIT_PACK") != nullptr;  // A/B switch for measurements
    const bool pack = !no_emit_pack && L->m == 0 && L->box_valid && L->ids_distinct && (sc.id_or >> BOX_ID_BITS) == 0;
    const KindInfo kic = kind_info(L->kind);
    const bool count_cells = L->m == 0 && L->cell_fuse_depth > 0 && kic.dim * L->cell_fuse_depth <= 9;
    const size_t cell_bytes = count_cells ? (size_t)HIST_COPIES * (((size_t)1 << (kic.dim * L->cell_fuse_depth)) + 1) * sizeof(unsigned long long) : 0;
    if (count_cells) {
        BP_TRY(ensure_dev(L, L->cell_hist_dev, cell_bytes));
        HIP_TRY(L, hipMemsetAsync(L->cell_hist_dev.p, 0, cell_bytes, L->stream));
    }
    if (count_digits) {
        BP_TRY(ensure_dev(L, L->tree_hist, (size_t)HIST_COPIES * SX_HIST_MAX * RS_BINS * sizeof(unsigned long long)));
        HIP_TRY(L, hipMemsetAsync(L->tree_hist.p, 0, (size_t)HIST_COPIES * SX_HIST_MAX * RS_BINS * sizeof(unsigned long long), L->stream));
    }
    {
        ProfScope ps(L, BP_K_ENCODE);
        HIP_TRY(L, launch_encode_emit(L->kind, sb, d_bounds, d_ids, n, L->min_depth, L->counts.p,
                                      (const unsigned long long*)L->block_sums.p, cur_keys(L) + L->m, cur_ids(L) + L->m,
                                      pack ? nullptr : (uint8_t*)L->box[L->cur].p + L->m, L->stream,
                                      count_digits ? &hist_plan : nullptr,
                                      count_digits ? (unsigned long long*)L->tree_hist.p : nullptr, pack ? 1 : 0,
                                      count_cells ? (unsigned long long*)L->cell_hist_dev.p : nullptr, L->cell_fuse_depth));
    }
    if (count_cells) L->cell_dev_valid = true;
    if (pack) { L->ids_packed = true; L->box_valid = false; }  // until the sort (or ensure_unpacked) splits them off
    L->tail_run = false;
    if (count_digits) { L->tree_hist_valid = true; L->tree_hist_plan = hist_plan; L->tree_hist_m = add; }
    BP_TRY(sync_stream(L));
    if (was_empty) {  // what this extend produced predicts the next frame's (one-pass extend)
        L->enc_hint.valid = true; L->enc_hint.n = n; L->enc_hint.entries = add; L->enc_hint.pack = pack;
        L->enc_hint.key_or = sc.key_or; L->enc_hint.key_and = sc.key_and;
        L->enc_hint.depth_mask = sc.depth_mask; L->enc_hint.id_min = sc.id_min; L->enc_hint.id_max = sc.id_max;
        L->enc_hint.ids_distinct = !(sc.flags & EXT_IDS_REPEAT);
    }
    L->m += add;
    L->sorted = false;  // src/layer.rs:119 (at least one valid object was appended)
    L->key_or |= sc.key_or; L->key_and &= sc.key_and;
    L->id_or |= sc.id_or; L->id_and &= sc.id_and;
    L->depth_mask |= sc.depth_mask;
    L->stats.entries = L->m;
    return BP_OK;
}

// fold facts about raw entries appended from outside (merge / set_tree / append_entries)
int absorb_raw_entries(bp_layer* L, const uint64_t* d_keys, const uint32_t* d_ids, size_t m) {
    if (m == 0) return BP_OK;
    BP_TRY(sc_reset(L));
    {
        ProfScope ps(L, BP_K_MISC);
        HIP_TRY(L, launch_tree_reduce(L->kind, d_keys, d_ids, m, (DevScalars*)L->d_sc.p, L->stream));
    }
    DevScalars sc;
    BP_TRY(sc_fetch(L, &sc));
    L->key_or |= sc.key_or; L->key_and &= sc.key_and;
    L->id_or |= sc.id_or; L->id_and &= sc.id_and;
    L->depth_mask |= sc.depth_mask;
    // nothing is known about the order or uniqueness of foreign ids: take the general paths
    L->order_ok = false;
    L->ids_distinct = false;
    L->have_ids = true;
    L->max_id = 0xFFFFFFFFu;
    L->min_id = 0;
    return BP_OK;
}

// ---- sort -----------------------------------------------------------------------------------------------
bool box_can_ride(const bp_layer* L) {  // the box codes travel in the id payload's bits 27..31
    return (L->box_valid || L->ids_packed || L->tree_compact) && L->order_ok && L->ids_distinct && (L->id_or >> BOX_ID_BITS) == 0;
}

// full-width LSD sort (every varying key bit, ids first when the order fact is missing)
int sort_tree_full(bp_layer* L, bool payload_already_packed) {
    const RadixPlan plan = make_radix_plan(L->key_or ^ L->key_and, L->order_ok ? 0 : 1, L->id_or ^ L->id_and);
    L->stats.tree_sort_passes += (uint32_t)plan.n_passes;
    // the box codes ride in the id payload of the KEY passes; without one (all index values equal) they cannot
    const bool has_key_pass = (L->key_or ^ L->key_and) != 0;
    if (payload_already_packed && (plan.n_passes == 0 || !has_key_pass)) {
        L->ids_packed = true;
        BP_TRY(ensure_unpacked(L));  // split the codes off first
        payload_already_packed = false;
    }
    if (plan.n_passes == 0) return BP_OK;
    const int other = L->cur ^ 1;
    BP_TRY(ensure_dev(L, L->keys[other], L->m * sizeof(uint64_t)));
    BP_TRY(ensure_dev(L, L->ids[other], L->m * sizeof(uint32_t)));
    const bool ride = has_key_pass && (payload_already_packed ||
                                       (L->box_valid && L->ids_distinct && (L->id_or >> BOX_ID_BITS) == 0));
    if (ride && !payload_already_packed && plan.from_val[0]) {
        // id passes come first and permute the records: the codes must be inside the ids before that
        HIP_TRY(L, launch_pack_box(cur_ids(L), (const uint8_t*)L->box[L->cur].p, L->m, L->stream));
        payload_already_packed = true;
    }
    const int dest = (plan.n_passes & 1) ? other : L->cur;
    if (ride) BP_TRY(ensure_dev(L, L->box[dest], L->m, dest == L->cur ? L->m : 0));
    BP_TRY(ensure_dev(L, L->radix_ws, radix_workspace_bytes(L->m, plan.n_passes)));
    BP_TRY(sc_reset(L));
    int res = 0;
    uint64_t* k0 = (uint64_t*)L->keys[L->cur].p; uint64_t* k1 = (uint64_t*)L->keys[other].p;
    uint32_t* v0 = (uint32_t*)L->ids[L->cur].p;  uint32_t* v1 = (uint32_t*)L->ids[other].p;
    HIP_TRY(L, radix_sort(k0, k1, v0, v1, L->m, plan, L->radix_ws.p, L->radix_ws.cap, (DevScalars*)L->d_sc.p, &res,
                          L->stream, prof_mark, L, BP_K_TREE_HIST, BP_K_TREE_SORT,
                          (ride && !payload_already_packed) ? (const uint8_t*)L->box[L->cur].p : nullptr,
                          ride ? (uint8_t*)L->box[dest].p : nullptr));
    DevScalars sc;
    BP_TRY(sc_fetch(L, &sc));
    if (res) L->cur = other;
    L->box_valid = ride;  // otherwise the codes stayed behind in append order
    return BP_OK;
}

// Partial tree sort: radix passes over the top key bits only, enough that the groups of equal prefix
// hold ~4 entries on average; the groups are then rank-sorted by (key, id) in LDS.  Returns false
// when the full-width sort should run instead.
bool tree_partial_plan(size_t m, unsigned long long varying, RadixPlan* plan, int* seg_shift_out) {
    static const bool no_finish = getenv("BP_NO_TREE_FINISH") != nullptr;  // A/B switch for measurements
    const int top = varying ? 64 - __builtin_clzll(varying) : 0;
    int want_bits = 8;
    while (want_bits < 56 && ((unsigned long long)m >> (want_bits + 2)) != 0) want_bits += 8;
    const int n_part = want_bits / 8;
    const int seg_shift = top - 8 * n_part;
    if (no_finish || m < 4096 || seg_shift <= 0 || (varying & ((1ull << seg_shift) - 1ull)) == 0) return false;
    plan->n_passes = n_part;
    for (int i = 0; i < RS_MAX_PASSES; ++i) { plan->shift[i] = 0; plan->from_val[i] = 0; }
    for (int i = 0; i < n_part; ++i) plan->shift[i] = seg_shift + 8 * i;
    *seg_shift_out = seg_shift;
    return true;
}
// Grouped tree sort: radix-sort only the top G key bits, and let launch_tgroup_finish order the groups in LDS -- one pass fewer than the classic partial sort.
bool tree_group_plan(size_t m, unsigned long long key_and, unsigned long long key_or, TreeGroupPlan* out) {
    static const bool disabled = getenv("BP_NO_TREE_GROUPS") != nullptr;  // A/B switch
    const unsigned long long varying = key_or ^ key_and;
    if (disabled || m < 65536 || m >= 0xFFFFFFFFull || varying == 0) return false;
    RadixPlan classic;
    int seg_shift = 0;
    if (!tree_partial_plan(m, varying, &classic, &seg_shift)) return false;
    const int top = 64 - __builtin_clzll(varying);
    // the fewest top bits G whose groups of equal prefix average at most 60 % of the LDS capacity (the largest groups
    // that still leave room for the scene's unevenness); worth it only if that needs fewer passes than the classic plan
    int G = 1;
    while (G < 56 && (double)m / (double)(1ull << G) > 0.6 * tgroup_capacity()) ++G;
    const int gshift = top - G;
    const int passes = (G + 7) / 8;
    if (G < 1 || gshift < tgroup_bin_bits() || passes >= classic.n_passes) return false;
    const unsigned long long g0 = key_and >> gshift, g1 = key_or >> gshift;  // key_and <= every key <= key_or
    if (g1 - g0 + 1 > (unsigned long long)m / 16 + 1) return false;
    out->plan.n_passes = passes;
    for (int i = 0; i < RS_MAX_PASSES; ++i) { out->plan.shift[i] = 0; out->plan.from_val[i] = 0; }
    for (int i = 0; i < passes; ++i) out->plan.shift[i] = gshift + 8 * i;
    out->gshift = gshift;
    out->g0 = g0;
    out->n_groups = (uint32_t)(g1 - g0 + 1);
    return true;
}
bool same_plan(const RadixPlan& a, const RadixPlan& b) {
    if (a.n_passes != b.n_passes) return false;
    for (int i = 0; i < a.n_passes; ++i)
        if (a.shift[i] != b.shift[i] || a.from_val[i] != b.from_val[i]) return false;
    return true;
}

static bool tgrp_expected(const bp_layer* L) {
    static const bool no_optimistic = getenv("BP_NO_OPTIMISTIC_FIRST_SORT") != nullptr;  // A/B switch
    return L->tgrp_ok || (!L->tgrp_known && !no_optimistic);
}
int sort_tree(bp_layer* L, bool defer_check = false);
int sort_tree_full(bp_layer* L, bool payload_already_packed);
// tree = [head | sorted tail run]: sort the head alone, then one merge pass.  Returns 1 when it did the job.
int sort_tree_by_merge(bp_layer* L, int* handled) {
    static const bool disabled = getenv("BP_NO_MERGE_SORT") != nullptr;  // A/B switch
    *handled = 0;
    const size_t m = L->m, n_head = L->run_split, n_tail = m - n_head;
    L->tail_run = false;
    if (disabled || n_head > n_tail || m >= 0xFFFFFFFFull) return BP_OK;  // a large head: one full sort is cheaper
    BP_TRY(ensure_unpacked(L));
    const int tail_buf = L->cur;
    if (n_head > 0) {
        // sort the head in place of the whole tree (the tail stays where it is, beyond the first n_head slots);
        // the tree-wide facts are a superset of the head's, so every kernel choice stays valid
        L->m = n_head;
        const int rc = sort_tree(L);
        L->m = m;
        if (rc != BP_OK) return rc;
        L->sorted = false;
        if (L->cur != tail_buf) {  // the sorted head ended in the other buffer: bring it next to the tail
            HIP_TRY(L, hipMemcpyAsync(L->keys[tail_buf].p, L->keys[L->cur].p, n_head * sizeof(uint64_t), hipMemcpyDeviceToDevice, L->stream));
            HIP_TRY(L, hipMemcpyAsync(L->ids[tail_buf].p, L->ids[L->cur].p, n_head * sizeof(uint32_t), hipMemcpyDeviceToDevice, L->stream));
            if (L->box_valid)
                HIP_TRY(L, hipMemcpyAsync(L->box[tail_buf].p, L->box[L->cur].p, n_head, hipMemcpyDeviceToDevice, L->stream));
            L->cur = tail_buf;
        }
    }
    const int src = L->cur, dst = src ^ 1;
    BP_TRY(ensure_dev(L, L->keys[dst], m * sizeof(uint64_t)));
    BP_TRY(ensure_dev(L, L->ids[dst], m * sizeof(uint32_t)));
    if (L->box_valid) BP_TRY(ensure_dev(L, L->box[dst], m));
    BP_TRY(ensure_dev(L, L->skip, merge_runs_split_words(m) * sizeof(uint32_t)));  // (scratch: the tiles' merge-path splits)
    {
        ProfScope ps(L, BP_K_TREE_SORT);
        HIP_TRY(L, launch_merge_runs((const uint64_t*)L->keys[src].p, (const uint32_t*)L->ids[src].p,
                                     L->box_valid ? (const uint8_t*)L->box[src].p : nullptr, n_head,
                                     (const uint64_t*)L->keys[src].p + n_head, (const uint32_t*)L->ids[src].p + n_head,
                                     L->box_valid ? (const uint8_t*)L->box[src].p + n_head : nullptr, n_tail,
                                     (uint64_t*)L->keys[dst].p, (uint32_t*)L->ids[dst].p,
                                     L->box_valid ? (uint8_t*)L->box[dst].p : nullptr, L->stream, (uint32_t*)L->skip.p));
    }
    BP_TRY(sync_stream(L));
    L->cur = dst;
    ++L->epoch;  // the head's own sort may have left per-entry context words for ITS order: the merged tree is a new one
    L->sorted = true;
    L->order_ok = true;
    ++L->stats.tree_sort_passes;
    *handled = 1;
    return BP_OK;
}

// The grouped tree sort was enqueued without waiting for it (sort_tree with defer_check): read its error word and group
// statistics now.  already_synced: the caller has just synchronised the stream.  A group that did not fit (the scene
// changed under the plan) is repaired here by the full-width sort of the partially sorted data; *redone tells the caller
// that whatever it enqueued behind the sort read a tree that was not yet in order and must be run again.
int settle_sort(bp_layer* L, bool already_synced, bool* redone) {
    if (redone) *redone = false;
    if (!L->sort_pend.on) return BP_OK;
    if (!already_synced) {
        BP_TRY(sort_sc_fetch(L));
        BP_TRY(sync_stream(L));
    }
    if (!L->sort_pend.copied) return fail(L, BP_ERR_KERNEL, "internal: a deferred sort check was settled without its scalars");
    L->sort_pend.on = false;
    const DevScalars sc = *(const DevScalars*)L->h_sort_sc.p;
    if (sc.err & KERR_LOOKBACK_TIMEOUT) return fail(L, BP_ERR_KERNEL, "radix look-back timed out");
    const bool tgrp_before = L->tgrp_ok;
    L->tgrp_known = true;
    L->tgrp_ok = sc.grp_max <= (uint32_t)tgroup_capacity();  // for the next sort (stays on up to the capacity)
    if ((sc.err & KERR_SEGMENT_TOO_LONG) || L->tgrp_ok != tgrp_before) ++L->stats.tree_plan_flips;
    if (sc.err & KERR_SEGMENT_TOO_LONG) {
        ++L->stats.sort_repairs;
        // (a cell-route attempt that failed with groups that fit: only the cells were laid out too small -- the grouped plan stays)
        if (!(L->sort_pend.cells && sc.grp_max <= (uint32_t)tgroup_capacity())) L->tgrp_ok = false;
        if (L->sort_pend.cells) L->tcell_off = 16;
        L->tcell.valid = false;
        L->cur = L->sort_pend.src;
        // (a compact sort: the first pass's dense output is composite words -- the repair sorts (index, id) records)
        bool packed_payload = L->sort_pend.ride;
        L->ids_lazy = false;           // (the repair writes real arrays)
        if (L->sort_pend.compact) {
            BP_TRY(expand_compact(L, L->cur, L->m));
            L->box_valid = true;       // (expanded: plain ids and box bytes, in the first pass's order)
            packed_payload = false;
        }
        L->ctx_epoch = 0;              // the context words were written for an order that is not the final one
        ++L->epoch;
        // the first radix pass placed records of equal digit in no particular order: equal keys are no longer in id order,
        // so the full-width sort must order by id as well
        L->order_ok = false;
        BP_TRY(sort_tree_full(L, packed_payload));
        L->ids_packed = false;
        BP_TRY(ensure_unpacked(L));
        L->sorted = true;
        L->order_ok = true;
        if (redone) *redone = true;
    }
    return BP_OK;
}

// the cell route of a grouped tree sort: its gapped output buffers, the cells' workspace and the launch tag of the cell walks;
// the cells are laid out (inside the sort's first kernel) from the group bounds the previous sort left in tgrp_bounds
int tcell_setup(bp_layer* L, const TreeGroupPlan& gp, bool with_vals, CellRoute* out) {
    CellRoute& cr = *out;
    const size_t slots = cell_layout_capacity(L->m);
    BP_TRY(ensure_dev(L, L->tcell_keys, slots * sizeof(uint64_t)));
    if (with_vals) BP_TRY(ensure_dev(L, L->tcell_ids, slots * sizeof(uint32_t)));
    const bool fresh = L->tcell_ws.p == nullptr;
    BP_TRY(ensure_dev(L, L->tcell_ws, (7 * (size_t)CELL_COUNT + 16 + 128 + 16) * sizeof(uint32_t)));
    uint32_t* const cw = (uint32_t*)L->tcell_ws.p;
    cr.src = (uint4*)cw;
    cr.cell_base = cw + 4 * (size_t)CELL_COUNT; cr.line_cur = cw + 5 * (size_t)CELL_COUNT + 16; cr.frag_cur = cw + 6 * (size_t)CELL_COUNT + 16;
    cr.status = (unsigned long long*)(cw + 7 * (size_t)CELL_COUNT + 16);
    cr.miss = cw + 7 * (size_t)CELL_COUNT + 16 + 128;
    cr.out_keys = (uint64_t*)L->tcell_keys.p; cr.out_vals = with_vals ? (uint32_t*)L->tcell_ids.p : nullptr;
    if (fresh) { HIP_TRY(L, hipMemsetAsync(cr.status, 0, 64 * sizeof(unsigned long long), L->stream)); L->tcell_tag = 0; }
    L->tcell_tag += 2;
    if (L->tcell_tag >= (1u << 23)) {
        HIP_TRY(L, hipMemsetAsync(cr.status, 0, 64 * sizeof(unsigned long long), L->stream));
        L->tcell_tag = 2;
    }
    cr.tag = L->tcell_tag;
    // (the layout runs inside the sort's first kernel)
    cr.layout_prev_bounds = (const uint32_t*)L->tgrp_bounds.p; cr.layout_g0 = (uint32_t)(gp.g0 & 0xFFFFu); cr.layout_groups = gp.n_groups;
    cr.layout_limit = slots;
    return BP_OK;
}

int sort_tree(bp_layer* L, bool defer_check) {
    if (L->sort_pend.on) {
        if (defer_check && L->sorted) return BP_OK;  // still pending: this caller settles it at its own host round trip
        BP_TRY(settle_sort(L));
    }
    if (L->sorted) return ensure_unpacked(L);  // src/layer.rs:148, 161
    ++L->epoch;
    // compact records (what the frame's extend wrote, expecting this sort to take the cell route): sorted as they are if that still
    // holds, expanded for every other plan
    bool compact_go = false;
    CompactPlan ccp{};
    if (L->tree_compact) {
        TreeGroupPlan gpc;
        compact_go = L->m > 1 && !L->tail_run && L->tgrp_ok && tree_group_plan(L->m, L->key_and, L->key_or, &gpc) &&
                     tree_cells_possible(L, L->m, gpc, /*before_sort=*/true) && compact_plan(gpc, L->cfmt, &ccp) && L->tree_hist_valid &&
                     L->tree_hist_m == L->m && same_plan(ccp.plan, L->tree_hist_plan);
        if (!compact_go) BP_TRY(ensure_plain(L));
    }
    if (L->tail_run && L->m > 1) {
        int handled = 0;
        BP_TRY(sort_tree_by_merge(L, &handled));
        if (handled) return BP_OK;
    }
    L->tail_run = false;
    L->stats.tree_sort_passes = 0;
    if (L->m > 1) {
        const unsigned long long varying = L->key_or ^ L->key_and;
        RadixPlan plan;
        int seg_shift = 0;
        const bool partial = tree_partial_plan(L->m, varying, &plan, &seg_shift);
        bool done = false;
        TreeGroupPlan gp;
        const bool group_able = tree_group_plan(L->m, L->key_and, L->key_or, &gp);
        // A layer's FIRST sort (nothing known about its group sizes yet) takes the grouped plan optimistically: the plan sizes its
        // groups for an average of 60 % of the finish's capacity, which a scene without heavy clustering fits -- one pass fewer and
        // the context words from the finish instead of three passes + segment finish + entry_ctx (a fresh layer's frame: 2.9 -> 2.6 ms
        // at 10 M objects).  A scene that does not fit pays ONE repair (the deferred check's full-width sort, bp_stats.sort_repairs)
        // and the layer remembers (tgrp_known); BP_NO_OPTIMISTIC_FIRST_SORT: the probing first sort of rounds 1 - 4
        const bool tgrp_go = group_able && tgrp_expected(L);
        if (group_able && L->tgrp_ok && compact_go) {
            // the cell route on compact records: key-only claiming passes over composite words (16 bytes per record and pass instead
            // of 24), the finish expands them into index values, ids, box bytes and context words
            L->stats.tree_sort_passes = 2;
            const int other = L->cur ^ 1, dst = L->cur;
            // the id and box arrays are written only on demand (ensure_ids) when every code is known: the scan reads the context words
            static const bool no_lazy = getenv("BP_NO_LAZY_IDS") != nullptr;  // A/B switch
            const bool lazy_ids = !no_lazy && L->box_all_known;
            BP_TRY(ensure_dev(L, L->keys[other], L->m * sizeof(uint64_t)));
            if (!lazy_ids) {
                BP_TRY(ensure_dev(L, L->ids[dst], L->m * sizeof(uint32_t)));
                BP_TRY(ensure_dev(L, L->box[dst], L->m));
            }
            // (readable a window beyond the tree: the scan's chunk loads are unconditional when it has the range lengths)
            BP_TRY(ensure_dev(L, L->idctx, (L->m + SCAN_PAD_ENTRIES) * sizeof(uint64_t)));
            BP_TRY(ensure_dev(L, L->dcnt, (L->m + SCAN_PAD_ENTRIES) * sizeof(uint16_t)));
            uint32_t* ws_epoch = nullptr;
            BP_TRY(radix_ws_tagged(L, radix_workspace_bytes(L->m, 2), &ws_epoch));
            BP_TRY(ensure_dev(L, L->tgrp_bounds, ((size_t)gp.n_groups + 1) * sizeof(uint32_t)));
            BP_TRY(ensure_dev(L, L->d_sort_sc, sizeof(DevScalars)));
            BP_TRY(ensure_host(L, L->h_sort_sc, sizeof(DevScalars)));
            const DevScalars ssc_init = neutral_scalars();
            DevScalars* const d_ssc = (DevScalars*)L->d_sort_sc.p;
            L->tree_hist_valid = false; L->cell_hist_valid = false; L->cell_dev_valid = false;
            if (L->tcell_off) --L->tcell_off;
            const uint32_t cell0 = (uint32_t)(gp.g0 & 0xFFFFu);
            CellRoute cr{};
            BP_TRY(tcell_setup(L, gp, /*with_vals=*/false, &cr));
            int res = 0;
            HIP_TRY(L, radix_sort((uint64_t*)L->keys[L->cur].p, (uint64_t*)L->keys[other].p, nullptr, nullptr, L->m, ccp.plan, L->radix_ws.p,
                                  L->radix_ws.cap, d_ssc, &res, L->stream, prof_mark, L, BP_K_TREE_HIST, BP_K_TREE_SORT, nullptr, nullptr, nullptr, 0,
                                  (const unsigned long long*)L->tree_hist.p, HIST_COPIES, /*unstable_first=*/1, ws_epoch, &ssc_init, nullptr, nullptr, &cr));
            {
                ProfScope ps(L, BP_K_TREE_FINISH);
                HIP_TRY(L, launch_cell_count(cr, cell0, gp.n_groups, (uint32_t*)L->tgrp_bounds.p, (uint32_t)tgroup_capacity(), group_stats_words(d_ssc),
                                             nullptr, L->stream, &d_ssc->err, (uint32_t)KERR_SEGMENT_TOO_LONG, (uint32_t)L->m));
                HIP_TRY(L, launch_tgroup_finish_compact((const uint64_t*)L->tcell_keys.p, gp.n_groups, ccp.comp_gshift, L->cfmt, (uint64_t*)L->keys[dst].p,
                                                        lazy_ids ? nullptr : (uint32_t*)L->ids[dst].p, lazy_ids ? nullptr : (uint8_t*)L->box[dst].p, d_ssc,
                                                        L->stream, L->kind, (uint64_t*)L->idctx.p, (uint16_t*)L->dcnt.p, group_stats_words(d_ssc),
                                                        (uint32_t)(L->m / 4), cr));
                ++L->stats.tree_sorts_in_cells;
                ++L->stats.tree_sorts_compact;
            }
            L->tcell.valid = true; L->tcell.g0 = gp.g0; L->tcell.n_groups = gp.n_groups; L->tcell.gshift = gp.gshift; L->tcell.m = L->m;
            // optimistic state: the sort succeeded (what a failed check does about it: settle_sort)
            L->cur = dst;
            L->tree_compact = false;
            L->ids_lazy = lazy_ids;
            L->box_valid = true;
            L->ctx_epoch = L->epoch; L->ctx_thr = 0; L->ctx_box = true; L->ctx_lo = 1; L->ctx_hi = 0;
            L->dcnt_epoch = L->epoch;
            L->ids_packed = false;
            L->sort_pend.on = true; L->sort_pend.src = other; L->sort_pend.ride = true; L->sort_pend.cells = true; L->sort_pend.copied = false;
            L->sort_pend.compact = true;
            static const bool no_defer_c = getenv("BP_NO_DEFERRED_SORT_CHECK") != nullptr;  // A/B switch
            if (defer_check && !no_defer_c) {
                L->sorted = true;
                L->order_ok = true;
                return BP_OK;
            }
            BP_TRY(settle_sort(L));
            done = true;
        } else if (group_able && tgrp_go) {  // one pass fewer; the groups of equal prefix are ordered in LDS
            L->stats.tree_sort_passes = (uint32_t)gp.plan.n_passes;
            const int other = L->cur ^ 1;
            BP_TRY(ensure_dev(L, L->keys[other], L->m * sizeof(uint64_t)));
            BP_TRY(ensure_dev(L, L->ids[other], L->m * sizeof(uint32_t)));
            const bool packed_in = L->ids_packed;
            const bool ride = packed_in || (L->box_valid && L->ids_distinct && (L->id_or >> BOX_ID_BITS) == 0);
            uint32_t* ws_epoch = nullptr;
            BP_TRY(radix_ws_tagged(L, radix_workspace_bytes(L->m, gp.plan.n_passes), &ws_epoch));
            BP_TRY(ensure_dev(L, L->tgrp_bounds, ((size_t)gp.n_groups + 1) * sizeof(uint32_t)));
            BP_TRY(ensure_dev(L, L->d_sort_sc, sizeof(DevScalars)));
            BP_TRY(ensure_host(L, L->h_sort_sc, sizeof(DevScalars)));
            // the sort's own scalars, reset by the sort's first kernel
            const DevScalars ssc_init = neutral_scalars();
            DevScalars* const d_ssc = (DevScalars*)L->d_sort_sc.p;
            int res = 0;
            uint64_t* kb[2] = {(uint64_t*)L->keys[L->cur].p, (uint64_t*)L->keys[other].p};
            uint32_t* vb[2] = {(uint32_t*)L->ids[L->cur].p, (uint32_t*)L->ids[other].p};
            const bool hist_ready = L->tree_hist_valid && L->tree_hist_m == L->m && same_plan(gp.plan, L->tree_hist_plan);
            L->tree_hist_valid = false; L->cell_hist_valid = false; L->cell_dev_valid = false;
            // The cell route (CellRoute, bp_kernels.h): claiming pass by the high digit, claiming pass into cells
            // (= the finish's groups) laid out from the bounds the previous sort left in tgrp_bounds, finish from the gapped
            // buffers -- no stable pass, no search for the bounds.  A cell that outgrows its range sets KERR_SEGMENT_TOO_LONG in
            // the sort's scalars: the deferred check repairs the tree from the first pass's (dense) output like any other failure.
            static const bool no_tcells = getenv("BP_NO_TREE_CELL_SORT") != nullptr;  // A/B switch
            const int distinct_bits = (ride && L->ids_distinct) ? std::max(1, 32 - __builtin_clz(L->id_or | 1u)) : 0;
            const uint32_t cell0 = (uint32_t)(gp.g0 & 0xFFFFu);
            if (L->tcell_off) --L->tcell_off;
            const bool tcell = !no_tcells && !L->tcell_off && L->tcell.valid && L->tcell.g0 == gp.g0 && L->tcell.n_groups == gp.n_groups &&
                               L->tcell.gshift == gp.gshift && gp.plan.n_passes == 2 && hist_ready && ride &&
                               tgroup_composite_applies(gp.gshift, distinct_bits) && (size_t)cell0 + gp.n_groups <= CELL_COUNT &&
                               radix_claim_applies(L->m) && L->m <= L->tcell.m + L->tcell.m / 16 && L->m + L->m / 16 >= L->tcell.m;
            CellRoute cr{};
            if (tcell) BP_TRY(tcell_setup(L, gp, /*with_vals=*/true, &cr));
            HIP_TRY(L, radix_sort(kb[0], kb[1], vb[0], vb[1], L->m, gp.plan, L->radix_ws.p, L->radix_ws.cap, d_ssc,
                                  &res, L->stream, prof_mark, L, BP_K_TREE_HIST, BP_K_TREE_SORT,
                                  (ride && !packed_in) ? (const uint8_t*)L->box[L->cur].p : nullptr, nullptr, nullptr, 0,
                                  hist_ready ? (const unsigned long long*)L->tree_hist.p : nullptr, HIST_COPIES,
                                  /*unstable_first=*/1, ws_epoch, &ssc_init, nullptr, nullptr,
                                  tcell ? &cr : nullptr));  // the finish kernel orders every group by (key, id, position)
            // cell route: the gapped result sits in tcell_keys / tcell_ids, the first pass's dense output in `other` (what a repair
            // starts from), and the finish writes the tree back into the current buffers
            const int src = tcell ? other : (res ? other : L->cur), dst = src ^ 1;
            if (ride) BP_TRY(ensure_dev(L, L->box[dst], L->m, dst == L->cur ? L->m : 0));
            // the finish has key and box code at hand: it also writes the scan's per-entry context words
            static const bool no_ctx_fuse = getenv("BP_NO_CTX_FUSE") != nullptr;  // A/B switch
            const bool with_ctx = ride && !no_ctx_fuse;
            if (with_ctx) BP_TRY(ensure_dev(L, L->idctx, L->m * sizeof(uint64_t)));
            {
                ProfScope ps(L, BP_K_TREE_FINISH);
                if (tcell)
                    HIP_TRY(L, launch_cell_count(cr, cell0, gp.n_groups, (uint32_t*)L->tgrp_bounds.p, (uint32_t)tgroup_capacity(), group_stats_words(d_ssc),
                                                 nullptr, L->stream, &d_ssc->err, (uint32_t)KERR_SEGMENT_TOO_LONG, (uint32_t)L->m));
                else
                    HIP_TRY(L, launch_tgroup_bounds((const uint64_t*)L->keys[src].p, L->m, gp.n_groups, gp.g0, gp.gshift,
                                                    (uint32_t*)L->tgrp_bounds.p, group_stats_words(d_ssc), L->stream));
                HIP_TRY(L, launch_tgroup_finish(tcell ? (const uint64_t*)L->tcell_keys.p : (const uint64_t*)L->keys[src].p,
                                                tcell ? (const uint32_t*)L->tcell_ids.p : (const uint32_t*)L->ids[src].p, gp.n_groups,
                                                (const uint32_t*)L->tgrp_bounds.p, gp.gshift, ride ? 1 : 0, (uint64_t*)L->keys[dst].p,
                                                (uint32_t*)L->ids[dst].p, ride ? (uint8_t*)L->box[dst].p : nullptr,
                                                d_ssc, L->stream, with_ctx ? L->kind : -1,
                                                with_ctx ? (uint64_t*)L->idctx.p : nullptr, group_stats_words(d_ssc),
                                                (uint32_t)(L->m / 4),
                                                // ids distinct (the codes ride in them): the composite-key kernel may run
                                                distinct_bits, tcell ? &cr : nullptr));
                if (tcell) ++L->stats.tree_sorts_in_cells;
            }
            // (optimistic, like the rest of the state below: tgrp_bounds will hold this sort's group bounds)
            L->tcell.valid = true; L->tcell.g0 = gp.g0; L->tcell.n_groups = gp.n_groups; L->tcell.gshift = gp.gshift; L->tcell.m = L->m;
            // optimistic state: the sort succeeded (what a failed check does about it: settle_sort)
            L->cur = dst;
            L->box_valid = ride;
            if (with_ctx) { L->ctx_epoch = L->epoch; L->ctx_thr = 0; L->ctx_box = true; L->ctx_lo = 1; L->ctx_hi = 0; }
            L->ids_packed = false;
            L->sort_pend.on = true; L->sort_pend.src = src; L->sort_pend.ride = ride; L->sort_pend.cells = tcell; L->sort_pend.copied = false;
            L->sort_pend.compact = false;
            static const bool no_defer = getenv("BP_NO_DEFERRED_SORT_CHECK") != nullptr;  // A/B switch
            if (defer_check && !no_defer) {  // no host round trip here: the check rides on the caller's next one
                L->sorted = true;
                L->order_ok = true;
                return BP_OK;
            }
            BP_TRY(settle_sort(L));
            done = true;
        }
        if (!done && partial) {
            const int n_part = plan.n_passes;
            L->stats.tree_sort_passes = (uint32_t)n_part;
            const int other = L->cur ^ 1;
            BP_TRY(ensure_dev(L, L->keys[other], L->m * sizeof(uint64_t)));
            BP_TRY(ensure_dev(L, L->ids[other], L->m * sizeof(uint32_t)));
            // the finish kernel orders equal keys by id itself, so the codes may ride without order_ok
            const bool packed_in = L->ids_packed;  // already packed by the sender (bp_layer_commit_entries)
            const bool ride = packed_in || (L->box_valid && L->ids_distinct && (L->id_or >> BOX_ID_BITS) == 0);
            BP_TRY(ensure_dev(L, L->radix_ws, radix_workspace_bytes(L->m, plan.n_passes)));
            BP_TRY(sc_reset(L));
            int res = 0;
            uint64_t* kb[2] = {(uint64_t*)L->keys[L->cur].p, (uint64_t*)L->keys[other].p};
            uint32_t* vb[2] = {(uint32_t*)L->ids[L->cur].p, (uint32_t*)L->ids[other].p};
            // first pass packs the box codes into the payload; the finish kernel splits them off again
            // digit counts left behind by encode_emit, if the tree is still exactly that extend's output
            const bool hist_ready = L->tree_hist_valid && L->tree_hist_m == L->m && same_plan(plan, L->tree_hist_plan);
            L->tree_hist_valid = false; L->cell_hist_valid = false; L->cell_dev_valid = false;
            HIP_TRY(L, radix_sort(kb[0], kb[1], vb[0], vb[1], L->m, plan, L->radix_ws.p, L->radix_ws.cap, (DevScalars*)L->d_sc.p,
                                  &res, L->stream, prof_mark, L, BP_K_TREE_HIST, BP_K_TREE_SORT,
                                  (ride && !packed_in) ? (const uint8_t*)L->box[L->cur].p : nullptr, nullptr, nullptr, 0,
                                  hist_ready ? (const unsigned long long*)L->tree_hist.p : nullptr, HIST_COPIES,
                                  /*unstable_first=*/1));  // the finish kernel orders every group by (key, id, position)
            const int src = res ? other : L->cur, dst = src ^ 1;
            if (ride) BP_TRY(ensure_dev(L, L->box[dst], L->m, dst == L->cur ? L->m : 0));
            {
                ProfScope ps(L, BP_K_TREE_FINISH);
                HIP_TRY(L, launch_tree_segment_finish((const uint64_t*)L->keys[src].p, (const uint32_t*)L->ids[src].p, L->m, seg_shift,
                                                      ride ? 1 : 0, (uint64_t*)L->keys[dst].p, (uint32_t*)L->ids[dst].p,
                                                      ride ? (uint8_t*)L->box[dst].p : nullptr, (DevScalars*)L->d_sc.p, L->stream));
            }
            // would the grouped finish suit this tree?  group sizes of the (now sorted) keys decide for the NEXT sort;
            // a scene that did not qualify is looked at again every 16th sort only
            const bool probe = group_able && (!L->tgrp_known || (++L->tgrp_probe & 15u) == 0);
            if (probe) {
                BP_TRY(ensure_dev(L, L->tgrp_bounds, ((size_t)gp.n_groups + 1) * sizeof(uint32_t)));
                ProfScope ps(L, BP_K_TREE_FINISH);
                HIP_TRY(L, launch_tgroup_bounds((const uint64_t*)L->keys[dst].p, L->m, gp.n_groups, gp.g0, gp.gshift,
                                                (uint32_t*)L->tgrp_bounds.p, group_stats_words((DevScalars*)L->d_sc.p), L->stream));
                // (exact group bounds of this tree under the grouped plan: what the cell route of the next sort lays its cells out by)
                L->tcell.valid = true; L->tcell.g0 = gp.g0; L->tcell.n_groups = gp.n_groups; L->tcell.gshift = gp.gshift; L->tcell.m = L->m;
            } else {
                L->tcell.valid = false;
            }
            HIP_TRY(L, hipMemcpyAsync(L->h_sc.p, L->d_sc.p, sizeof(DevScalars), hipMemcpyDeviceToHost, L->stream));
            BP_TRY(sync_stream(L));
            const DevScalars sc = *(DevScalars*)L->h_sc.p;
            if (sc.err & KERR_LOOKBACK_TIMEOUT) return fail(L, BP_ERR_KERNEL, "radix look-back timed out");
            if (probe && !(sc.err & KERR_SEGMENT_TOO_LONG)) {
                L->tgrp_known = true;
                // switched on only with a margin (80 % of the capacity): a scene hovering at the limit must not flip the
                // plan every frame -- each flip costs the digit counts encode_emit made for the other plan
                const bool tgrp_before = L->tgrp_ok;
                L->tgrp_ok = sc.grp_max <= (uint32_t)(tgroup_capacity() * 4 / 5);
                if (L->tgrp_ok != tgrp_before) ++L->stats.tree_plan_flips;
            }
            if (sc.err & KERR_SEGMENT_TOO_LONG) {
                // a group of equal key prefix is longer than the LDS window (heavily clustered input):
                // finish with the full-width sort, starting from the partially sorted (still packed) data
                L->cur = src;
                L->order_ok = false;  // the unstable first pass has shuffled equal keys: sort by id as well
                BP_TRY(sort_tree_full(L, ride));
            } else {
                L->cur = dst;
                L->box_valid = ride;
            }
            L->ids_packed = false;
            done = true;
        }
        if (!done) {
            BP_TRY(sort_tree_full(L, L->ids_packed));
            L->ids_packed = false;
        }
    }
    BP_TRY(ensure_unpacked(L));
    L->sorted = true;
    L->order_ok = true;  // a (key, id)-sorted array satisfies the stable-sort precondition again
    return BP_OK;
}

// ---- scan ------------------------------------------------------------------------------------------------
// produces the raw pair stream (packed id << 32 | id_) in L->pairs[0], in scan_impl emission order
int scan_raw_device(bp_layer* L, uint32_t coarse_depth, int emit_coarse, bool source_dedup = false) {
    BP_TRY(sort_tree(L));  // src/layer.rs:465, 499
    L->n_raw = 0;
    L->raw_holes = false;
    L->owner_hist_valid = false;
    if (L->m == 0) return BP_OK;
    if (L->m >= 0xFFFFFFFFull) return fail(L, BP_ERR_TOO_LARGE, "scan: more than 2^32-1 index entries");
    const size_t nb = scan_blocks(L->m);
    BP_TRY(ensure_dev(L, L->counts, L->m * sizeof(uint32_t)));
    BP_TRY(ensure_dev(L, L->block_sums, nb * sizeof(unsigned long long)));
    ScanParams p;
    p.keys = (const unsigned long long*)cur_keys(L);
    p.ids = cur_ids(L);
    p.m = L->m;
    p.depth_mask = L->depth_mask;
    p.coarse_depth = coarse_depth;
    p.emit_coarse = emit_coarse;
    p.coarse_id_lo = L->coarse_lo; p.coarse_id_hi = L->coarse_hi;
    p.general = L->ids_distinct ? 0 : 1;
    p.skip = nullptr;
    p.filt_bounds = nullptr;
    p.filt_rows = 0;
    p.filt_shape = FILT_AABB;
    p.box = (source_dedup && !p.general && L->box_valid) ? (const uint8_t*)L->box[L->cur].p : nullptr;
    L->raw_is_unique = p.box != nullptr && L->box_all_known;
    if (p.general) {
        BP_TRY(ensure_dev(L, L->skip, L->m));
        p.skip = (uint8_t*)L->skip.p;
    }
    BP_TRY(sc_reset(L));
    {
        ProfScope ps(L, BP_K_SCAN_COUNT);
        if (p.general) HIP_TRY(L, launch_scan_skipflags(L->kind, p, L->stream));
        HIP_TRY(L, launch_scan_count(L->kind, p, (uint32_t*)L->counts.p, (unsigned long long*)L->block_sums.p, L->stream));
    }
    {
        ProfScope ps(L, BP_K_MISC);
        HIP_TRY(L, launch_scan_block_sums((unsigned long long*)L->block_sums.p, nb, nullptr, (DevScalars*)L->d_sc.p, L->stream));
    }
    DevScalars sc;
    BP_TRY(sc_fetch(L, &sc));
    const size_t n_raw = (size_t)sc.total;
    L->n_raw = n_raw;
    L->stats.raw_pairs = n_raw;
    L->stats.raw_candidates = p.box ? 0 : n_raw;  // without suppression this IS the reference's raw stream
    if (n_raw == 0) return BP_OK;
    BP_TRY(ensure_dev(L, L->pairs[0], n_raw * sizeof(uint64_t)));
    {
        ProfScope ps(L, BP_K_SCAN_EMIT);
        HIP_TRY(L, launch_scan_emit(L->kind, p, (const uint32_t*)L->counts.p, (const unsigned long long*)L->block_sums.p,
                                    (uint64_t*)L->pairs[0].p, L->stream));
    }
    return BP_OK;
}

// sort + dedup of n packed pairs sitting in buf0 (scratch partner: L->pairs[1]) -> L->out_pairs
// (interleaved), L->n_pairs
int make_raw_dense(bp_layer* L);
bool segment_finish_enabled() {
    static const bool disabled = getenv("BP_NO_SEGMENT_FINISH") != nullptr;  // A/B switch
    return !disabled;
}
// the plan the sort of a pair stream will use (distinct pairs: first id only, then segment finish)
RadixPlan pair_sort_plan(unsigned long long varying, bool distinct, bool grouped = false, int dups_bits = 0) {
    if (!distinct && segment_finish_enabled()) {  // sort_dups_pairs_in: first id only, above the grouped finish's low bits
        const RadixPlan plan = make_radix_plan(varying & ~((1ull << (32 + dups_bits)) - 1ull), 0, 0);
        if (plan.n_passes > 0) return plan;
    }
    if (distinct && segment_finish_enabled()) {
        if (grouped) {  // the grouped finish takes the lowest 8 bits of the first id as well
            const RadixPlan plan = make_radix_plan(varying & 0xFFFFFF0000000000ull, 0, 0);
            if (plan.n_passes > 0) return plan;
        }
        const RadixPlan plan = make_radix_plan(varying & 0xFFFFFFFF00000000ull, 0, 0);
        if (plan.n_passes > 0) return plan;
    }
    return make_radix_plan(varying, 0, 0);
}
// Grouped finish (launch_group_finish) instead of the last first-id pass + segment_finish: worth it when a group of
// 256 consecutive first ids holds a few hundred to ~2400 pairs on average (the kernel takes 4096).  id_lo / id_hi bound the first ids (AND / OR of the
// layer's ids).  Returns the number of groups, 0 = use the classic path.
uint32_t group_finish_groups(size_t n, unsigned long long varying, uint32_t id_lo, uint32_t id_hi) {
    static const bool disabled = getenv("BP_NO_GROUP_FINISH") != nullptr;  // A/B switch
    if (disabled || !segment_finish_enabled() || n < 4096 || n >= 0xFFFFFFFFull || id_hi < id_lo) return 0;
    const uint32_t idv = (uint32_t)(varying >> 32);
    if ((idv >> 8) == 0 || (idv & 0xFFu) != 0xFFu) return 0;  // needs varying bits above the group digit and a full low digit
    const unsigned long long span = (unsigned long long)id_hi - id_lo + 1ull;
    const double per_group = (double)n * 256.0 / (double)span;
    if (per_group < 320.0 || per_group > 0.6 * group_finish_capacity()) return 0;
    return (id_hi >> 8) - (id_lo >> 8) + 1u;
}
// digit counts the scan left behind for exactly this plan (consumed by the first sort that asks)
constexpr size_t PAIR_HIST_WORDS = (size_t)HIST_COPIES * SX_HIST_MAX * RS_BINS;
const unsigned long long* take_ready_hist(bp_layer* L, const RadixPlan& plan, bool holes) {
    if (!L->pair_hist_valid) return nullptr;
    L->pair_hist_valid = false;
    if (!holes || plan.n_passes != L->pair_hist_plan.n_passes) return nullptr;
    for (int i = 0; i < plan.n_passes; ++i)
        if (plan.shift[i] != L->pair_hist_plan.shift[i] || plan.from_val[i] != L->pair_hist_plan.from_val[i]) return nullptr;
    return (const unsigned long long*)L->pair_hist.p;
}
// holes: buf0 == pairs[0] holds the block-structured stream (L->blockfill, L->raw_extent)
int sort_unique_pairs_in(bp_layer* L, uint64_t* buf0, size_t n, unsigned long long varying, bool assume_unique = false,
                         bool holes = false);

// n DISTINCT packed pairs in buf0: radix sort on the first id only, then rank-sort the small
// equal-first-id segments in LDS while writing the ABI layout (3 passes + 1 kernel instead of
// 6 passes + 1 kernel).  Falls back to the full-width sort when a segment is too long.
int sort_distinct_pairs_in(bp_layer* L, uint64_t* buf0, size_t n, unsigned long long varying, bool holes = false,
                           const uint32_t* first_range = nullptr) {
    const unsigned long long hi_var = varying & 0xFFFFFFFF00000000ull;
    // first_range: inclusive bounds of the first ids (the grouped finish needs them), nullptr when unknown
    // after a refusal (uneven groups) the grouped finish stays off for the next 16 distinct-pair sorts of this layer
    if (L->grp_off) --L->grp_off;
    const uint32_t n_groups = (first_range && !L->grp_off) ? group_finish_groups(n, varying, first_range[0], first_range[1]) : 0u;
    if (L->grp_next != (n_groups != 0)) ++L->stats.pair_plan_flips;
    L->grp_next = n_groups != 0;  // the next scan counts its digit histograms for the plan this frame used
    const RadixPlan plan = n_groups ? make_radix_plan(hi_var & 0xFFFFFF0000000000ull, 0, 0) : make_radix_plan(hi_var, 0, 0);
    if (!segment_finish_enabled() || n < 2 || plan.n_passes == 0) return sort_unique_pairs_in(L, buf0, n, varying, true, holes);
    L->stats.pair_sort_passes = (uint32_t)plan.n_passes;
    int res = 0;
    uint64_t* bufs[2] = {buf0, nullptr};
    // (buffers are sized with the margin the next frame's sort will ask for when it is enqueued ahead of its scan's sizes:
    // that frame then allocates nothing)
    const bool margin = holes && n_groups != 0;
    const size_t n_alloc = margin ? n + n / 8 + 4096 : n;
    BP_TRY(ensure_dev(L, L->pairs[1], n_alloc * sizeof(uint64_t)));
    bufs[1] = (uint64_t*)L->pairs[1].p;
    {
        const size_t extent = holes ? L->raw_extent : n;
        const size_t extent_alloc = margin ? std::max(extent + extent / 8 + (size_t)SX_SHARDS * SX_BLOCK, n_alloc) : extent;
        uint32_t* ws_epoch = nullptr;
        BP_TRY(radix_ws_tagged(L, radix_workspace_bytes(extent_alloc, plan.n_passes), &ws_epoch));
        const DevScalars sc_init = neutral_scalars();  // the scalars are reset by the sort's first kernel
        HIP_TRY(L, radix_sort(bufs[0], bufs[1], nullptr, nullptr, n, plan, L->radix_ws.p, L->radix_ws.cap,
                              (DevScalars*)L->d_sc.p, &res, L->stream, prof_mark, L, BP_K_PAIR_HIST, BP_K_PAIR_SORT, nullptr, nullptr,
                              holes ? (const uint32_t*)L->blockfill.p : nullptr, extent, take_ready_hist(L, plan, holes), HIST_COPIES,
                              /*unstable_first=*/1, ws_epoch, &sc_init));  // distinct pairs, finished by (first id, second id) in LDS
    }
    BP_TRY(ensure_dev(L, L->out_pairs, n_alloc * 2 * sizeof(uint32_t)));
    if (n_groups) {
        BP_TRY(ensure_dev(L, L->grp_bounds, ((size_t)n_groups + 1) * sizeof(uint32_t)));
        ProfScope ps(L, BP_K_DEDUP);
        HIP_TRY(L, launch_group_finish(bufs[res], n, n_groups, first_range[0] >> 8, 8, (uint32_t*)L->grp_bounds.p, (uint32_t*)L->out_pairs.p,
                                       (DevScalars*)L->d_sc.p, L->stream, group_stats_words((DevScalars*)L->d_sc.p),
                                       (uint32_t)(varying & 0xFFFFFFFFull)));
    } else {
        ProfScope ps(L, BP_K_DEDUP);
        HIP_TRY(L, launch_segment_finish(bufs[res], n, (uint32_t*)L->out_pairs.p, (DevScalars*)L->d_sc.p, L->stream));
    }
    HIP_TRY(L, hipMemcpyAsync(L->h_sc.p, L->d_sc.p, sizeof(DevScalars), hipMemcpyDeviceToHost, L->stream));
    BP_TRY(sync_stream(L));
    const DevScalars sc = *(DevScalars*)L->h_sc.p;
    if (sc.err & KERR_LOOKBACK_TIMEOUT) return fail(L, BP_ERR_KERNEL, "radix look-back timed out");
    if (sc.err & KERR_SEGMENT_TOO_LONG) {  // a group / segment did not fit its window: sort all 64 bits instead
        L->grp_next = false;
        L->ahead.valid = false;
        if (n_groups) { L->grp_off = 16; ++L->stats.pair_plan_flips; }
        return sort_unique_pairs_in(L, bufs[res], n, varying, true);
    }
    L->n_pairs = n;
    L->stats.pairs = n;
    // what this sort looked like: the next frame's may be enqueued ahead of its scan's sizes (scan_fused_device)
    L->ahead.valid = holes && n_groups != 0 && buf0 == (uint64_t*)L->pairs[0].p;
    if (L->ahead.valid) {
        L->ahead.n_raw = n; L->ahead.extent = L->raw_extent; L->ahead.m = L->m;
        L->ahead.n_groups = n_groups; L->ahead.first_lo = first_range[0]; L->ahead.first_hi = first_range[1];
        L->ahead.varying = varying; L->ahead.plan = plan;
        // (should the sort that goes out ahead take the cell route, the scan's buffer must hold the gapped layout: scan_fused_device)
        const size_t nb = n + n / 8 + 4096;
        if (plan.n_passes == 2 && !plan.from_val[0] && !plan.from_val[1] && plan.shift[0] == 40 && plan.shift[1] == 48 && (first_range[1] >> 24) == (first_range[0] >> 24) &&
            radix_claim_applies(std::max(L->raw_extent + L->raw_extent / 8, nb)))
            L->pairs_want_slots = std::max(L->pairs_want_slots, cell_layout_capacity(nb) + cell_layout_capacity(nb) / 16);
    }
    return BP_OK;
}

// n packed pairs that MAY REPEAT in buf0: radix sort on the first id only, rank-sort the equal-first-id segments
// in LDS (equal pairs by position), then the adjacent-unique pass -- 3 passes + finish instead of 6 passes.
// *done = 0: not applicable / a segment was too long (the caller runs the full-width sort).
int sort_dups_pairs_in(bp_layer* L, uint64_t* buf0, size_t n, unsigned long long varying, bool holes, int* done,
                       const uint32_t* first_range = nullptr) {
    *done = 0;
    // grouped route: radix passes above the low 7 (6) bits of the first id, one workgroup per group of ids orders the
    // group, drops the repeats and leaves its survivors compacted; a copy kernel closes the gaps between groups
    if (first_range && !L->grp_off_dups && n >= 4096 && n < 0xFFFFFFFFull && first_range[1] >= first_range[0] && segment_finish_enabled() &&
        !getenv("BP_NO_GROUP_FINISH")) {
        const uint32_t idv = (uint32_t)(varying >> 32);
        const unsigned long long span = (unsigned long long)first_range[1] - first_range[0] + 1ull;
        // groups of 2^gb consecutive first ids: the widest that leaves headroom below the kernel's capacity
        int gb = group_finish_dups_max_bits();
        // (headroom of 2.4x the mean: ids handed out along the scene's structure give uneven groups)
        const double mean_max = 0.42 * group_finish_dups_capacity();
        while (gb > 5 && (double)n * (double)(1u << gb) / (double)span > mean_max) --gb;
        const double per_group = (double)n * (double)(1u << gb) / (double)span;
        const uint32_t low = (1u << gb) - 1u;
        if ((idv >> gb) != 0 && (idv & low) == low && per_group >= 320.0 && per_group <= mean_max) {
            const uint32_t n_groups = (first_range[1] >> gb) - (first_range[0] >> gb) + 1u;
            const RadixPlan gplan = make_radix_plan(varying & ~((1ull << (32 + gb)) - 1ull), 0, 0);
            L->stats.pair_sort_passes = (uint32_t)gplan.n_passes;
            DevBuf& partner = (buf0 == (uint64_t*)L->pairs[1].p) ? L->pairs[0] : L->pairs[1];
            BP_TRY(ensure_dev(L, partner, n * sizeof(uint64_t)));
            uint64_t* bufs[2] = {buf0, (uint64_t*)partner.p};
            int res = 0;
            const size_t extent = holes ? L->raw_extent : n;
            uint32_t* ws_epoch = nullptr;
            BP_TRY(radix_ws_tagged(L, radix_workspace_bytes(extent, gplan.n_passes), &ws_epoch));
            const DevScalars sc_init = neutral_scalars();  // the scalars are reset by the sort's first kernel
            HIP_TRY(L, radix_sort(bufs[0], bufs[1], nullptr, nullptr, n, gplan, L->radix_ws.p, L->radix_ws.cap, (DevScalars*)L->d_sc.p,
                                  &res, L->stream, prof_mark, L, BP_K_PAIR_HIST, BP_K_PAIR_SORT, nullptr, nullptr,
                                  holes ? (const uint32_t*)L->blockfill.p : nullptr, extent, take_ready_hist(L, gplan, holes), HIST_COPIES,
                                  /*unstable_first=*/1, ws_epoch, &sc_init));  // equal pairs are interchangeable; the finish orders by (first, second)
            BP_TRY(ensure_dev(L, L->grp_bounds, ((size_t)n_groups + 1) * sizeof(uint32_t)));
            L->ahead.valid = false;  // (grp_bounds no longer holds the last distinct-pair sort's groups)
            BP_TRY(ensure_dev(L, L->block_sums, ((size_t)n_groups + 1) * sizeof(unsigned long long)));
            {
                ProfScope ps(L, BP_K_DEDUP);
                HIP_TRY(L, launch_group_finish_dups(bufs[res], n, n_groups, first_range[0] >> gb, gb, (uint32_t*)L->grp_bounds.p, bufs[res ^ 1],
                                                    (unsigned long long*)L->block_sums.p, (DevScalars*)L->d_sc.p, L->stream,
                                                    group_stats_words((DevScalars*)L->d_sc.p), first_range[0], first_range[1]));
            }
            {
                ProfScope ps(L, BP_K_MISC);
                HIP_TRY(L, launch_scan_block_sums((unsigned long long*)L->block_sums.p, (size_t)n_groups + 1, nullptr, (DevScalars*)L->d_sc.p,
                                                  L->stream));
            }
            DevScalars scg;
            BP_TRY(sc_fetch(L, &scg));
            if (!(scg.err & KERR_SEGMENT_TOO_LONG)) {
                const size_t npg = (size_t)scg.total;
                BP_TRY(ensure_dev(L, L->out_pairs, npg * 2 * sizeof(uint32_t)));
                {
                    ProfScope ps(L, BP_K_DEDUP);
                    HIP_TRY(L, launch_group_compact(bufs[res ^ 1], (const uint32_t*)L->grp_bounds.p, (const unsigned long long*)L->block_sums.p,
                                                    n_groups, (uint32_t*)L->out_pairs.p, L->stream));
                }
                BP_TRY(sync_stream(L));
                L->n_pairs = npg;
                L->stats.pairs = npg;
                L->grp_next_dups = gb;
                *done = 1;
                return BP_OK;
            }
            // uneven groups: the stream (grouped by the high digits, nothing lost) goes through the ungrouped route below;
            // the grouped one stays off for the next 16 sorts of this layer
            L->grp_off_dups = 16;
            L->grp_next_dups = 0;
            buf0 = bufs[res];
            holes = false;
        }
    } else if (L->grp_off_dups) {
        --L->grp_off_dups;
    }
    const RadixPlan plan = make_radix_plan(varying & 0xFFFFFFFF00000000ull, 0, 0);
    if (!segment_finish_enabled() || n < 2 || plan.n_passes == 0) return BP_OK;
    L->stats.pair_sort_passes = (uint32_t)plan.n_passes;
    DevBuf& partner = (buf0 == (uint64_t*)L->pairs[1].p) ? L->pairs[0] : L->pairs[1];
    BP_TRY(ensure_dev(L, partner, n * sizeof(uint64_t)));
    uint64_t* bufs[2] = {buf0, (uint64_t*)partner.p};
    int res = 0;
    BP_TRY(sc_reset(L));
    {
        const size_t extent = holes ? L->raw_extent : n;
        BP_TRY(ensure_dev(L, L->radix_ws, radix_workspace_bytes(extent, plan.n_passes)));
        HIP_TRY(L, radix_sort(bufs[0], bufs[1], nullptr, nullptr, n, plan, L->radix_ws.p, L->radix_ws.cap,
                              (DevScalars*)L->d_sc.p, &res, L->stream, prof_mark, L, BP_K_PAIR_HIST, BP_K_PAIR_SORT, nullptr, nullptr,
                              holes ? (const uint32_t*)L->blockfill.p : nullptr, extent, take_ready_hist(L, plan, holes), HIST_COPIES));
    }
    uint64_t* sorted = bufs[res ^ 1];  // the finish writes into the other ping-pong buffer (both hold >= n)
    {
        ProfScope ps(L, BP_K_DEDUP);
        HIP_TRY(L, launch_segment_finish_dups(bufs[res], n, sorted, (DevScalars*)L->d_sc.p, L->stream));
    }
    const size_t nb = uq_blocks(n);
    BP_TRY(ensure_dev(L, L->block_sums, nb * sizeof(unsigned long long)));
    {
        ProfScope ps(L, BP_K_DEDUP);
        HIP_TRY(L, launch_unique_count(sorted, n, (unsigned long long*)L->block_sums.p, L->stream));
    }
    {
        ProfScope ps(L, BP_K_MISC);
        HIP_TRY(L, launch_scan_block_sums((unsigned long long*)L->block_sums.p, nb, nullptr, (DevScalars*)L->d_sc.p, L->stream));
    }
    HIP_TRY(L, hipMemcpyAsync(L->h_sc.p, L->d_sc.p, sizeof(DevScalars), hipMemcpyDeviceToHost, L->stream));
    BP_TRY(sync_stream(L));
    const DevScalars sc = *(DevScalars*)L->h_sc.p;
    if (sc.err & KERR_LOOKBACK_TIMEOUT) return fail(L, BP_ERR_KERNEL, "radix look-back timed out");
    if (sc.err & KERR_SEGMENT_TOO_LONG) {  // some id owns a very long segment: sort all 64 bits instead, from the grouped data
        int rc = sort_unique_pairs_in(L, bufs[res], n, varying, false, false);
        if (rc == BP_OK) *done = 1;
        return rc;
    }
    const size_t np = (size_t)sc.total;
    BP_TRY(ensure_dev(L, L->out_pairs, np * 2 * sizeof(uint32_t)));
    {
        ProfScope ps(L, BP_K_DEDUP);
        HIP_TRY(L, launch_unique_emit(sorted, n, (const unsigned long long*)L->block_sums.p, (uint32_t*)L->out_pairs.p, L->stream));
    }
    BP_TRY(sync_stream(L));
    L->n_pairs = np;
    L->stats.pairs = np;
    *done = 1;
    return BP_OK;
}

int sort_unique_pairs_in(bp_layer* L, uint64_t* buf0, size_t n, unsigned long long varying, bool assume_unique, bool holes) {
    L->n_pairs = 0;
    L->stats.pairs = 0;
    if (n == 0) return BP_OK;
    const RadixPlan plan = make_radix_plan(varying, 0, 0);
    if (holes && !(plan.n_passes > 0 && n > 1)) {  // no pass to read around the holes
        BP_TRY(make_raw_dense(L));
        buf0 = (uint64_t*)L->pairs[0].p;
        holes = false;
    }
    L->stats.pair_sort_passes = (uint32_t)plan.n_passes;
    int res = 0;
    uint64_t* bufs[2] = {buf0, nullptr};
    if (plan.n_passes > 0 && n > 1) {
        DevBuf& partner = (buf0 == (uint64_t*)L->pairs[1].p) ? L->pairs[0] : L->pairs[1];  // never buf0 itself
        BP_TRY(ensure_dev(L, partner, n * sizeof(uint64_t)));
        bufs[1] = (uint64_t*)partner.p;
        const size_t extent = holes ? L->raw_extent : n;
        const size_t wsb = radix_workspace_bytes(extent, plan.n_passes);
        BP_TRY(ensure_dev(L, L->radix_ws, wsb));
        BP_TRY(sc_reset(L));
        HIP_TRY(L, radix_sort(bufs[0], bufs[1], nullptr, nullptr, n, plan, L->radix_ws.p, L->radix_ws.cap,
                              (DevScalars*)L->d_sc.p, &res, L->stream, prof_mark, L, BP_K_PAIR_HIST, BP_K_PAIR_SORT, nullptr, nullptr,
                              holes ? (const uint32_t*)L->blockfill.p : nullptr, extent, take_ready_hist(L, plan, holes), HIST_COPIES));
    } else {
        BP_TRY(sc_reset(L));
    }
    const uint64_t* sorted = bufs[res];
    if (assume_unique) {  // duplicates were suppressed at the source: only the ABI layout conversion is left
        DevScalars scu;
        BP_TRY(ensure_dev(L, L->out_pairs, n * 2 * sizeof(uint32_t)));
        {
            ProfScope ps(L, BP_K_DEDUP);
            HIP_TRY(L, launch_unpack_pairs(sorted, n, (uint32_t*)L->out_pairs.p, L->stream));
        }
        BP_TRY(sc_fetch(L, &scu));  // surfaces a look-back timeout of the pair sort
        L->n_pairs = n;
        L->stats.pairs = n;
        return BP_OK;
    }
    const size_t nb = uq_blocks(n);
    BP_TRY(ensure_dev(L, L->block_sums, nb * sizeof(unsigned long long)));
    {
        ProfScope ps(L, BP_K_DEDUP);
        HIP_TRY(L, launch_unique_count(sorted, n, (unsigned long long*)L->block_sums.p, L->stream));
    }
    {
        ProfScope ps(L, BP_K_MISC);
        HIP_TRY(L, launch_scan_block_sums((unsigned long long*)L->block_sums.p, nb, nullptr, (DevScalars*)L->d_sc.p, L->stream));
    }
    DevScalars sc;
    BP_TRY(sc_fetch(L, &sc));  // also surfaces a look-back timeout of the pair sort
    const size_t np = (size_t)sc.total;
    BP_TRY(ensure_dev(L, L->out_pairs, np * 2 * sizeof(uint32_t)));
    {
        ProfScope ps(L, BP_K_DEDUP);
        HIP_TRY(L, launch_unique_emit(sorted, n, (const unsigned long long*)L->block_sums.p, (uint32_t*)L->out_pairs.p, L->stream));
    }
    BP_TRY(sync_stream(L));
    L->n_pairs = np;
    L->stats.pairs = np;
    return BP_OK;
}

int sort_unique_pairs(bp_layer* L, size_t n, uint32_t id_varying, bool assume_unique = false, bool holes = false, bool own_ids = false) {
    const unsigned long long varying = ((unsigned long long)id_varying << 32) | id_varying;
    // own_ids: the first ids are ids of this layer, so its AND / OR bound them
    const uint32_t own_range[2] = {L->id_and, L->id_or};
    if (assume_unique) return sort_distinct_pairs_in(L, (uint64_t*)L->pairs[0].p, n, varying, holes, own_ids ? own_range : nullptr);
    int done = 0;
    BP_TRY(sort_dups_pairs_in(L, (uint64_t*)L->pairs[0].p, n, varying, holes, &done, own_ids ? own_range : nullptr));
    if (done) return BP_OK;
    return sort_unique_pairs_in(L, (uint64_t*)L->pairs[0].p, n, varying, false, holes);
}

// raw pairs in ARBITRARY order (they are sorted afterwards): one walk per entry, workgroup-staged
// allow_holes: the caller's next step reads around the holes of the expanding scan's block-structured
// output (sort_unique_pairs); otherwise pairs[0] is made dense before returning
int make_raw_dense(bp_layer* L);
// hist_plan: the radix plan of the pair sort that will follow (its digit histograms are then counted
// by the scan itself and the sort skips its histogram kernel)
int scan_fused_device(bp_layer* L, uint32_t coarse_depth, int emit_coarse, bool source_dedup, const float* filt_bounds = nullptr,
                      uint32_t filt_rows = 0, bool allow_holes = false, const RadixPlan* hist_plan = nullptr,
                      const uint32_t* owner_bounds = nullptr, int owner_n = -1) {
    BP_TRY(sort_tree(L, /*defer_check=*/true));  // settled at this function's own host round trip (after the scan kernel)
    L->n_raw = 0;
    L->raw_is_unique = false;
    L->raw_holes = false;
    L->pair_hist_valid = false;
    L->owner_hist_valid = false;
    if (L->m == 0) return settle_sort(L);
    if (L->m >= 0xFFFFFFFFull) return fail(L, BP_ERR_TOO_LARGE, "scan: more than 2^32-1 index entries");
    // the scan kernel ran behind a tree sort whose check was deferred and that check failed: the tree has been repaired,
    // run the scan again (it read entries that were not yet in order -- memory-safe, but its output is void)
    auto rerun = [&]() {
        return scan_fused_device(L, coarse_depth, emit_coarse, source_dedup, filt_bounds, filt_rows, allow_holes, hist_plan, owner_bounds, owner_n);
    };
    ScanParams p;
    p.keys = (const unsigned long long*)cur_keys(L);
    p.ids = cur_ids(L);
    p.m = L->m;
    p.depth_mask = L->depth_mask;
    p.coarse_depth = coarse_depth;
    p.emit_coarse = emit_coarse;
    p.coarse_id_lo = L->coarse_lo; p.coarse_id_hi = L->coarse_hi;
    p.general = 0;
    p.skip = nullptr;
    p.filt_bounds = filt_bounds;
    p.filt_rows = filt_rows;
    p.filt_shape = L->filt_shape;
    // (after a compact sort the box array may not have been written yet -- ensure_ids -- so "codes in use" is a flag, not the pointer)
    const bool have_box = source_dedup && L->box_valid;
    p.box = have_box ? (const uint8_t*)L->box[L->cur].p : nullptr;
    L->raw_is_unique = have_box && L->box_all_known;
    static const bool no_expand = getenv("BP_NO_EXPAND_SCAN") != nullptr;  // A/B switch for measurements
    static const bool no_range_counts = getenv("BP_NO_RANGE_COUNTS") != nullptr;  // A/B switch: the scan bisects the range ends in its key window
    bool expand = !no_expand;
    if (expand) {
        // per-entry {id, suppression context} words; recomputed only when the tree or the mode changed
        const uint32_t thr = emit_coarse == 1 ? 0u : coarse_depth;
        // emit_coarse == 2: coarse entries whose id lies in the layer's range stay active (an empty range otherwise)
        const uint32_t alo = (emit_coarse == 2 && thr) ? L->coarse_lo : 1u, ahi = (emit_coarse == 2 && thr) ? L->coarse_hi : 0u;
        const bool with_box = have_box;
        if (L->ctx_epoch != L->epoch || L->ctx_thr != thr || L->ctx_box != with_box || L->ctx_lo != alo || L->ctx_hi != ahi) {
            BP_TRY(ensure_ids(L));  // (the words are rebuilt from the id / box arrays: those must exist)
            p.ids = cur_ids(L);
            if (have_box) p.box = (const uint8_t*)L->box[L->cur].p;
            BP_TRY(ensure_dev(L, L->idctx, L->m * sizeof(uint64_t)));
            ProfScope ps(L, BP_K_SCAN_COUNT);
            HIP_TRY(L, launch_entry_ctx(L->kind, cur_keys(L), cur_ids(L), p.box, L->m, thr, (uint64_t*)L->idctx.p, L->stream, alo, ahi));
            L->ctx_epoch = L->epoch; L->ctx_thr = thr; L->ctx_box = with_box; L->ctx_lo = alo; L->ctx_hi = ahi;
        }
        size_t big_cap = std::max<size_t>(65536, L->m / 16);
        if (const char* e = getenv("BP_BIGLIST_CAP")) big_cap = std::max<size_t>(1, (size_t)atoll(e));  // test hook: force the overflow fallback
        L->big_cap = big_cap;
        BP_TRY(ensure_dev(L, L->big_list, big_cap * 2 * sizeof(uint64_t)));  // (two words per item)
    }
    size_t cap = std::max<size_t>(L->pairs[0].cap / sizeof(uint64_t), 6 * L->m);
    cap = std::max(cap, L->pairs_want_slots);  // (what the previous frame's cell route would have needed: below)
    for (int attempt = 0; attempt < 4; ++attempt) {
        BP_TRY(ensure_dev(L, L->pairs[0], cap * sizeof(uint64_t)));
        cap = L->pairs[0].cap / sizeof(uint64_t);
        if (!expand) BP_TRY(sc_reset(L));
        if (expand) {
            const size_t out_blocks = cap / SX_BLOCK;
            const size_t shard_bytes = 3 * SX_SHARDS * sizeof(uint64_t);
            BP_TRY(ensure_dev(L, L->blockfill, out_blocks * sizeof(uint32_t)));
            uint64_t* const d_shards = sc_shards_dev(L);  // behind the scalars: one copy fetches both
            const bool count_digits = allow_holes && hist_plan && scan_can_count(*hist_plan);
            const bool count_owners = allow_holes && !count_digits && owner_n >= 0 && owner_n <= 15;
            L->owner_hist_valid = false;
            if (count_digits || count_owners) BP_TRY(ensure_dev(L, L->pair_hist, PAIR_HIST_WORDS * sizeof(unsigned long long)));  // (the scan spreads its digit counts over HIST_COPIES copies)
            // scalars, fill counts, cursors and histograms are all reset by one small kernel
            static const bool no_prepare = getenv("BP_NO_SCAN_PREPARE") != nullptr;  // A/B switch for measurements
            if (no_prepare) {
                BP_TRY(sc_reset(L));
                HIP_TRY(L, hipMemsetAsync(L->blockfill.p, 0, out_blocks * sizeof(uint32_t), L->stream));
                HIP_TRY(L, hipMemsetAsync(d_shards, 0, shard_bytes, L->stream));
                if (count_digits || count_owners)
                    HIP_TRY(L, hipMemsetAsync(L->pair_hist.p, 0, PAIR_HIST_WORDS * sizeof(unsigned long long), L->stream));
            } else
            HIP_TRY(L, launch_scan_prepare((DevScalars*)L->d_sc.p, neutral_scalars(), (uint32_t*)L->blockfill.p, out_blocks,
                                           d_shards, 3 * SX_SHARDS,
                                           (count_digits || count_owners) ? (unsigned long long*)L->pair_hist.p : nullptr,
                                           count_digits ? (uint32_t)PAIR_HIST_WORDS : (uint32_t)(SX_HIST_MAX * RS_BINS), L->stream));
            // the sort that goes out behind this scan (below): decided HERE, because the scan counts only the digits that sort will
            // read -- on the cell route the high digit alone (the second pass is laid out from the previous frame's group sizes)
            const bool ahead_now = L->ahead_try && attempt == 0 && count_digits;
            const bp_layer::AheadHint h = L->ahead;
            size_t n_bound = 0, extent_bound = 0;
            bool cells_apply = false, cell_route = false;
            if (ahead_now) {
                n_bound = std::min<size_t>(h.n_raw + h.n_raw / 8 + 4096, out_blocks * SX_BLOCK);
                extent_bound = std::min<size_t>(std::max(h.extent + h.extent / 8 + (size_t)SX_SHARDS * SX_BLOCK, n_bound), out_blocks * SX_BLOCK);
                // The cell route (CellRoute, bp_kernels.h): two claiming passes -- high digit, then every high bin over its cells
                // (= the finish's groups), laid out from the group sizes the LAST frame's finish left in grp_bounds -- instead of
                // a claiming and a stable pass + the search for the group bounds.  A cell that outgrows its range is a miss.
                static const bool no_cells = getenv("BP_NO_CELL_SORT") != nullptr;  // A/B switch for measurements
                if (L->cell_off) --L->cell_off;
                cells_apply = !no_cells && !L->cell_off && h.plan.n_passes == 2 && !h.plan.from_val[0] && !h.plan.from_val[1] &&
                              h.plan.shift[0] == 40 && h.plan.shift[1] == 48 &&
                              // (a pair's cell is bits 8 .. 23 of its first id: unique per 256-id group as long as the ids' top byte does not vary --
                              // ids of any magnitude, round 5; rounds 3 - 4 asked for first ids below 2^24)
                              (h.first_hi >> 24) == (h.first_lo >> 24) && radix_claim_applies(extent_bound);
                cell_route = cells_apply && cell_layout_capacity(n_bound) <= out_blocks * SX_BLOCK;
            }
            static const bool all_rows = getenv("BP_SCAN_ALL_DIGITS") != nullptr;  // A/B switch: both digit rows whatever the route
            const uint32_t hist_rows = (cell_route && !all_rows) ? 2u : 0xFFFFFFFFu;  // (row 1 = the plan's second digit, shift 48)
            {
                ProfScope ps(L, BP_K_SCAN_EMIT);
                HIP_TRY(L, launch_scan_expand(L->kind, cur_keys(L), (const uint64_t*)L->idctx.p, L->m, filt_bounds, L->filt_shape,
                                              (uint64_t*)L->pairs[0].p, out_blocks, (uint32_t*)L->blockfill.p, d_shards,
                                              (uint64_t*)L->big_list.p,
                                              (uint32_t)std::min<size_t>(L->big_cap, 0xFFFFFFFFu),
                                              (DevScalars*)L->d_sc.p, L->stream, count_digits ? hist_plan : nullptr,
                                              (count_digits || count_owners) ? (unsigned long long*)L->pair_hist.p : nullptr,
                                              count_owners ? owner_bounds : nullptr, count_owners ? owner_n : -1, 0, L->scan_listed,
                                              // the range lengths the tree finish left for THIS tree (else the kernel searches the range ends)
                                              (!no_range_counts && L->dcnt_epoch == L->epoch) ? (const uint16_t*)L->dcnt.p : nullptr, hist_rows));
            }
            if (ahead_now) {
                // ---- the pair sort and its finish go out behind the scan, sized by the last frame's numbers; ONE round trip ----
                L->ahead_try = false;
                uint32_t* const d_sizes = sc_sizes_dev(L);
                // (the sort's first kernel derives the sizes from the shard words: launch_scan_finalize's job without the launch)
                const ScanFinalize fin{d_shards, out_blocks, n_bound, extent_bound, (const DevScalars*)L->d_sc.p, d_sizes};
                uint64_t* bufs[2] = {(uint64_t*)L->pairs[0].p, nullptr};
                BP_TRY(ensure_dev(L, L->pairs[1], n_bound * sizeof(uint64_t)));
                bufs[1] = (uint64_t*)L->pairs[1].p;
                BP_TRY(ensure_dev(L, L->grp_bounds, ((size_t)h.n_groups + 1) * sizeof(uint32_t)));
                // the gapped layout needs n + n / 4 + 80 slots per cell: more than the raw stream's buffer holds when the scene is small
                // (5.2 M slots for the 65536 cells alone).  The NEXT frame's buffer is sized for it (one reallocation, 42 MB at most
                // beyond what the pairs need) -- before, a small scene took this route or not depending on how its buffer had grown
                if (cells_apply && !cell_route) L->pairs_want_slots = cell_layout_capacity(n_bound) + cell_layout_capacity(n_bound) / 16;
                if (getenv("BP_DEBUG_AHEAD"))
                    fprintf(stderr, "sort ahead: cells %d (off %u, passes %d shifts %d %d, first_hi %x, claim %d, capacity %zu <= %zu) n_bound %zu extent_bound %zu\n",
                            (int)cell_route, L->cell_off, h.plan.n_passes, h.plan.shift[0], h.plan.shift[1], h.first_hi, (int)radix_claim_applies(extent_bound),
                            cell_layout_capacity(n_bound), out_blocks * SX_BLOCK, n_bound, extent_bound);
                CellRoute cr{};
                if (cell_route) {
                    const bool fresh = L->cell_ws.p == nullptr;
                    BP_TRY(ensure_dev(L, L->cell_ws, (7 * (size_t)CELL_COUNT + 16 + 128) * sizeof(uint32_t)));
                    uint32_t* const cw = (uint32_t*)L->cell_ws.p;
                    cr.status = (unsigned long long*)(cw + 7 * (size_t)CELL_COUNT + 16);
                    if (fresh) { HIP_TRY(L, hipMemsetAsync(cr.status, 0, 64 * sizeof(unsigned long long), L->stream)); L->cell_tag = 0; }
                    L->cell_tag += 2;
                    if (L->cell_tag >= (1u << 23)) {  // (24-bit tags: start over on clean words)
                        HIP_TRY(L, hipMemsetAsync(cr.status, 0, 64 * sizeof(unsigned long long), L->stream));
                        L->cell_tag = 2;
                    }
                    cr.tag = L->cell_tag;
                    cr.src = (uint4*)cw;  // (16-byte aligned: first)
                    cr.cell_base = cw + 4 * (size_t)CELL_COUNT; cr.line_cur = cw + 5 * (size_t)CELL_COUNT + 16; cr.frag_cur = cw + 6 * (size_t)CELL_COUNT + 16;
                    cr.miss = d_sizes + SIZES_CELL_MISS;
                    // (the layout runs inside the sort's first kernel)
                    cr.layout_prev_bounds = (const uint32_t*)L->grp_bounds.p; cr.layout_g0 = (h.first_lo >> 8) & 0xFFFFu; cr.layout_groups = h.n_groups;
                    cr.layout_limit = out_blocks * SX_BLOCK;
                }
                int res = 0;
                {
                    uint32_t* ws_epoch = nullptr;
                    BP_TRY(radix_ws_tagged(L, radix_workspace_bytes(extent_bound, h.plan.n_passes), &ws_epoch));
                    const DevScalars sc_init = neutral_scalars();
                    HIP_TRY(L, radix_sort(bufs[0], bufs[1], nullptr, nullptr, n_bound, h.plan, L->radix_ws.p, L->radix_ws.cap, (DevScalars*)L->d_sc.p,
                                          &res, L->stream, prof_mark, L, BP_K_PAIR_HIST, BP_K_PAIR_SORT, nullptr, nullptr,
                                          (const uint32_t*)L->blockfill.p, extent_bound, (const unsigned long long*)L->pair_hist.p, HIST_COPIES,
                                          /*unstable_first=*/1, ws_epoch, &sc_init, d_sizes, &fin, cell_route ? &cr : nullptr));
                }
                BP_TRY(ensure_dev(L, L->out_pairs, n_bound * 2 * sizeof(uint32_t)));
                {
                    ProfScope ps(L, BP_K_DEDUP);
                    HIP_TRY(L, launch_group_finish(bufs[res], n_bound, h.n_groups, h.first_lo >> 8, 8, (uint32_t*)L->grp_bounds.p, (uint32_t*)L->out_pairs.p,
                                                   (DevScalars*)L->d_sc.p, L->stream, group_stats_words((DevScalars*)L->d_sc.p),
                                                   (uint32_t)(h.varying & 0xFFFFFFFFull), d_sizes, cell_route ? &cr : nullptr));
                }
                static const bool no_publish = getenv("BP_NO_PUBLISH") != nullptr;  // A/B switch
                if (!no_publish) {
                    // the frame's one blocking round trip: the last kernel publishes the scalars block into pinned memory and the host
                    // polls for it (a D2H copy + hipStreamSynchronize: the copy's hand-over and the completion interrupt on top)
                    static_assert(SC_BLOCK_BYTES <= PUB_SORT_OFF && PUB_SORT_OFF + sizeof(DevScalars) <= PUB_FLAG_OFF && SC_BLOCK_BYTES % 4 == 0 &&
                                  sizeof(DevScalars) % 4 == 0, "published block layout");
                    Published fp{};
                    BP_TRY(pub_next(L, &fp));
                    // (the scalars of a tree sort whose check is still pending ride along)
                    const bool with_sort = L->sort_pend.on && !L->sort_pend.copied;
                    HIP_TRY(L, launch_publish(L->d_sc.p, SC_BLOCK_BYTES, fp, L->stream, with_sort ? L->d_sort_sc.p : nullptr, sizeof(DevScalars), PUB_SORT_OFF));
                    BP_TRY(pub_wait(L, fp.seq));
                    std::memcpy(L->h_sc.p, L->pub, SC_BLOCK_BYTES);
                    if (with_sort) {
                        std::memcpy(L->h_sort_sc.p, (const char*)L->pub + PUB_SORT_OFF, sizeof(DevScalars));
                        L->sort_pend.copied = true;
                    }
                    BP_TRY(after_sync(L));
                } else {
                    BP_TRY(sort_sc_fetch(L));
                    HIP_TRY(L, hipMemcpyAsync(L->h_sc.p, L->d_sc.p, SC_BLOCK_BYTES, hipMemcpyDeviceToHost, L->stream));
                    BP_TRY(sync_stream(L));
                }
                ++L->stats.sorts_ahead;
                if (cell_route) ++L->stats.sorts_in_cells;
                {
                    bool redone = false;
                    BP_TRY(settle_sort(L, /*already_synced=*/true, &redone));
                    if (redone) { L->ahead.valid = false; return rerun(); }
                }
                const DevScalars sc = *(const DevScalars*)L->h_sc.p;
                const uint32_t* sz = sc_sizes_host(L);
                L->scan_listed = sz[SIZES_BIG] != 0 ? 1 : -1;  // (the scan's own scalars were reset by the sort's first kernel)
                if (sc.err & KERR_LOOKBACK_TIMEOUT) return fail(L, BP_ERR_KERNEL, "radix look-back timed out");
                if (sz[SIZES_MISS] || (cell_route && (sz[SIZES_CELL_MISS] || (sc.err & KERR_SEGMENT_TOO_LONG)))) {
                    // the frame outgrew the margins (or overflowed the work list): everything behind the scan idled; or a cell
                    // outgrew its range / a group the finish's window (the gapped cells cannot be handed to the full-width sort)
                    ++L->stats.sorts_ahead_missed;
                    L->ahead.valid = false;
                    if (getenv("BP_DEBUG_AHEAD"))
                        fprintf(stderr, "sort ahead missed: sizes extent %u n %u miss %u cell_miss %u err %x (bounds n %zu extent %zu, hint n %zu) cells %d grp_max %u\n",
                                sz[SIZES_EXTENT], sz[SIZES_N], sz[SIZES_MISS], sz[SIZES_CELL_MISS], sc.err, n_bound, extent_bound, h.n_raw, (int)cell_route, sc.grp_max);
                    if (!sz[SIZES_MISS]) L->cell_off = 16;  // the cells were the problem: the next sorts ahead go the two-kernel-kinds way
                    return rerun();
                }
                const uint64_t* sh = sc_shards_host(L);
                uint64_t examined = 0;
                for (int i = 0; i < SX_SHARDS; ++i) examined += sh[2 * SX_SHARDS + i];
                const size_t n = sz[SIZES_N];
                L->n_raw = n;
                L->stats.raw_pairs = n;
                L->stats.raw_candidates = examined;
                L->raw_holes = false;  // (consumed)
                L->raw_extent = sz[SIZES_EXTENT];
                L->stats.pair_sort_passes = (uint32_t)h.plan.n_passes;
                L->ahead_done = true;
                if (sc.err & KERR_SEGMENT_TOO_LONG) {  // as on the blocking route: sort all 64 bits instead
                    L->grp_next = false;
                    L->ahead.valid = false;
                    L->grp_off = 16; ++L->stats.pair_plan_flips;
                    return sort_unique_pairs_in(L, bufs[res], n, h.varying, true);
                }
                L->n_pairs = n;
                L->stats.pairs = n;
                L->ahead.n_raw = n; L->ahead.extent = sz[SIZES_EXTENT]; L->ahead.m = L->m;
                return BP_OK;
            }
            L->ahead_try = false;
            DevScalars sc;
            BP_TRY(sc_fetch(L, &sc, /*with_shards=*/true));
            {
                bool redone = false;
                BP_TRY(settle_sort(L, /*already_synced=*/true, &redone));
                if (redone) return rerun();
            }
            const bool was_inline = L->scan_listed < 0;  // (this launch went without the work-list kernel)
            L->scan_listed = sc.big_count != 0 ? 1 : -1;
            if (sc.err & KERR_BIGLIST_OVERFLOW) {
                // without the work-list kernel: a range too long for its wave appeared -- once more, with the kernel;
                // with it: pathological nesting, the per-entry walk handles it
                if (!was_inline) expand = false;
                continue;
            }
            const uint64_t* sh = sc_shards_host(L);
            uint64_t kmax = 0, total = 0, examined = 0;
            for (int i = 0; i < SX_SHARDS; ++i) {
                kmax = std::max<uint64_t>(kmax, sh[i]);
                total += sh[SX_SHARDS + i];
                examined += sh[2 * SX_SHARDS + i];
            }
            const size_t extent_blocks = (size_t)kmax * SX_SHARDS;
            if (extent_blocks <= out_blocks) {
                L->n_raw = (size_t)total;
                L->stats.raw_pairs = L->n_raw;
                L->stats.raw_candidates = examined;
                L->raw_holes = true;
                L->raw_extent = extent_blocks * SX_BLOCK;
                if (count_digits) { L->pair_hist_plan = *hist_plan; L->pair_hist_valid = true; }
                if (count_owners) L->owner_hist_valid = true;
                if (!allow_holes) BP_TRY(make_raw_dense(L));
                return BP_OK;
            }
            cap = extent_blocks * SX_BLOCK + extent_blocks * SX_BLOCK / 16;  // the buffer was too small: grow and redo
            continue;
        }
        BP_TRY(ensure_ids(L));  // (the per-entry walk reads the id / box arrays)
        p.ids = cur_ids(L);
        if (have_box) p.box = (const uint8_t*)L->box[L->cur].p;
        {
            ProfScope ps(L, BP_K_SCAN_EMIT);
            HIP_TRY(L, launch_scan_fused(L->kind, p, (uint64_t*)L->pairs[0].p, cap, (DevScalars*)L->d_sc.p, L->stream));
        }
        DevScalars sc;
        BP_TRY(sc_fetch(L, &sc));
        {
            bool redone = false;
            BP_TRY(settle_sort(L, /*already_synced=*/true, &redone));
            if (redone) return rerun();
        }
        if ((size_t)sc.total <= cap) {
            L->n_raw = (size_t)sc.total;
            L->stats.raw_pairs = L->n_raw;
            L->stats.raw_candidates = have_box ? 0 : L->n_raw;
            return BP_OK;
        }
        cap = (size_t)sc.total + (size_t)sc.total / 16;  // the buffer was too small: grow and redo
    }
    return fail(L, BP_ERR_KERNEL, "fused scan: output size did not converge");
}

// ids that may repeat: the reference's raw stream AS A SET (every combo of two entries the skip rule keeps,
// src/layer.rs:563-570) from two expanding passes -- MARK the skipped entries, then EMIT -- instead of the three
// ancestor walks of the literal path.  Block-structured output in pairs[0]; *done = 0 when the work list
// overflowed (the caller falls back to the literal kernels).
int scan_general_expand_device(bp_layer* L, uint32_t coarse_depth, int emit_coarse, int* done) {
    *done = 0;
    BP_TRY(sort_tree(L));
    L->n_raw = 0;
    L->raw_is_unique = false;
    L->raw_holes = false;
    L->pair_hist_valid = false;
    L->owner_hist_valid = false;
    if (L->m == 0) { *done = 1; return BP_OK; }
    if (L->m >= 0xFFFFFFFFull) return fail(L, BP_ERR_TOO_LARGE, "scan: more than 2^32-1 index entries");
    const uint32_t thr = emit_coarse == 1 ? 0u : coarse_depth;
    BP_TRY(ensure_dev(L, L->idctx, L->m * sizeof(uint64_t)));
    L->ctx_epoch = 0;  // the words below carry skip marks, not suppression contexts
    {
        ProfScope ps(L, BP_K_SCAN_COUNT);
        HIP_TRY(L, launch_entry_ctx(L->kind, cur_keys(L), cur_ids(L), nullptr, L->m, thr, (uint64_t*)L->idctx.p, L->stream,
                                    (emit_coarse == 2 && thr) ? L->coarse_lo : 1u, (emit_coarse == 2 && thr) ? L->coarse_hi : 0u));
    }
    size_t big_cap = std::max<size_t>(65536, L->m / 16);
    if (const char* e = getenv("BP_BIGLIST_CAP")) big_cap = std::max<size_t>(1, (size_t)atoll(e));
    L->big_cap = big_cap;
    BP_TRY(ensure_dev(L, L->big_list, big_cap * 2 * sizeof(uint64_t)));  // (two words per item)
    const size_t shard_bytes = 3 * SX_SHARDS * sizeof(uint64_t);
    BP_TRY(ensure_dev(L, L->shards, shard_bytes));
    BP_TRY(ensure_host(L, L->h_shards, shard_bytes));
    size_t cap = std::max<size_t>(L->pairs[0].cap / sizeof(uint64_t), 6 * L->m);
    BP_TRY(ensure_dev(L, L->pairs[0], cap * sizeof(uint64_t)));
    BP_TRY(ensure_dev(L, L->blockfill, (L->pairs[0].cap / sizeof(uint64_t) / SX_BLOCK) * sizeof(uint32_t)));
    BP_TRY(sc_reset(L));
    {
        ProfScope ps(L, BP_K_SCAN_COUNT);
        HIP_TRY(L, launch_scan_expand(L->kind, cur_keys(L), (const uint64_t*)L->idctx.p, L->m, nullptr, FILT_AABB, (uint64_t*)L->pairs[0].p, 0,
                                      (uint32_t*)L->blockfill.p, (uint64_t*)L->shards.p, (uint64_t*)L->big_list.p,
                                      (uint32_t)std::min<size_t>(big_cap, 0xFFFFFFFFu), (DevScalars*)L->d_sc.p, L->stream,
                                      nullptr, nullptr, nullptr, -1, /*gen=*/1));
    }
    {
        DevScalars sc;
        BP_TRY(sc_fetch(L, &sc));
        if (sc.err & KERR_BIGLIST_OVERFLOW) return BP_OK;  // literal path instead
    }
    for (int attempt = 0; attempt < 4; ++attempt) {
        BP_TRY(ensure_dev(L, L->pairs[0], cap * sizeof(uint64_t)));
        cap = L->pairs[0].cap / sizeof(uint64_t);
        const size_t out_blocks = cap / SX_BLOCK;
        BP_TRY(ensure_dev(L, L->blockfill, out_blocks * sizeof(uint32_t)));
        BP_TRY(sc_reset(L));
        HIP_TRY(L, hipMemsetAsync(L->blockfill.p, 0, out_blocks * sizeof(uint32_t), L->stream));
        HIP_TRY(L, hipMemsetAsync(L->shards.p, 0, shard_bytes, L->stream));
        // the sort that follows groups by first id first (sort_dups_pairs_in): count those digits while emitting
        const uint32_t idv = L->id_or ^ L->id_and;
        // (grp_next_dups: the previous frame's sort took the grouped route, which stops above that many low bits)
        const RadixPlan hplan = make_radix_plan(((unsigned long long)idv << 32) & ~((1ull << (32 + L->grp_next_dups)) - 1ull), 0, 0);
        const bool count_digits = segment_finish_enabled() && hplan.n_passes > 0 && scan_can_count(hplan);
        if (count_digits) {
            BP_TRY(ensure_dev(L, L->pair_hist, PAIR_HIST_WORDS * sizeof(unsigned long long)));
            HIP_TRY(L, hipMemsetAsync(L->pair_hist.p, 0, PAIR_HIST_WORDS * sizeof(unsigned long long), L->stream));
        }
        {
            ProfScope ps(L, BP_K_SCAN_EMIT);
            HIP_TRY(L, launch_scan_expand(L->kind, cur_keys(L), (const uint64_t*)L->idctx.p, L->m, nullptr, FILT_AABB, (uint64_t*)L->pairs[0].p,
                                          out_blocks, (uint32_t*)L->blockfill.p, (uint64_t*)L->shards.p, (uint64_t*)L->big_list.p,
                                          (uint32_t)std::min<size_t>(big_cap, 0xFFFFFFFFu), (DevScalars*)L->d_sc.p, L->stream,
                                          count_digits ? &hplan : nullptr, count_digits ? (unsigned long long*)L->pair_hist.p : nullptr,
                                          nullptr, -1, /*gen=*/2));
        }
        HIP_TRY(L, hipMemcpyAsync(L->h_shards.p, L->shards.p, shard_bytes, hipMemcpyDeviceToHost, L->stream));
        DevScalars sc;
        BP_TRY(sc_fetch(L, &sc));
        if (sc.err & KERR_BIGLIST_OVERFLOW) return BP_OK;
        const uint64_t* sh = (const uint64_t*)L->h_shards.p;
        uint64_t kmax = 0, total = 0;
        for (int i = 0; i < SX_SHARDS; ++i) { kmax = std::max<uint64_t>(kmax, sh[i]); total += sh[SX_SHARDS + i]; }
        const size_t extent_blocks = (size_t)kmax * SX_SHARDS;
        if (extent_blocks <= out_blocks) {
            L->n_raw = (size_t)total;
            L->stats.raw_pairs = L->n_raw;
            L->stats.raw_candidates = L->n_raw;  // no suppression in this mode: the raw stream is the reference's
            if (count_digits) { L->pair_hist_plan = hplan; L->pair_hist_valid = true; }
            L->raw_holes = true;
            L->raw_extent = extent_blocks * SX_BLOCK;
            *done = 1;
            return BP_OK;
        }
        cap = extent_blocks * SX_BLOCK + extent_blocks * SX_BLOCK / 16;
    }
    return fail(L, BP_ERR_KERNEL, "general scan: output size did not converge");
}

// block-structured raw stream in pairs[0] -> dense (for consumers that cannot read around the holes)
int make_raw_dense(bp_layer* L) {
    if (!L->raw_holes) return BP_OK;
    L->raw_holes = false;
    if (L->n_raw == 0) return BP_OK;
    const size_t nb = L->raw_extent / SX_BLOCK;
    BP_TRY(ensure_dev(L, L->block_sums, nb * sizeof(unsigned long long)));
    BP_TRY(ensure_dev(L, L->pairs[1], L->n_raw * sizeof(uint64_t)));
    BP_TRY(sc_reset(L));
    ProfScope ps(L, BP_K_MISC);
    HIP_TRY(L, launch_blockfill_to_u64((const uint32_t*)L->blockfill.p, nb, (unsigned long long*)L->block_sums.p, L->stream));
    HIP_TRY(L, launch_scan_block_sums((unsigned long long*)L->block_sums.p, nb, nullptr, (DevScalars*)L->d_sc.p, L->stream));
    HIP_TRY(L, launch_compact_blocks((const uint64_t*)L->pairs[0].p, nb, (const uint32_t*)L->blockfill.p,
                                     (const unsigned long long*)L->block_sums.p, (uint64_t*)L->pairs[1].p, L->stream));
    std::swap(L->pairs[0], L->pairs[1]);
    return BP_OK;
}

int scan_device(bp_layer* L, uint32_t coarse_depth, int emit_coarse) {
    static const bool no_source_dedup = getenv("BP_NO_SOURCE_DEDUP") != nullptr;  // A/B switches for measurements
    static const bool no_fused = getenv("BP_NO_FUSED_SCAN") != nullptr;
    if (L->ids_distinct && !no_fused) {
        BP_TRY(sort_tree(L, /*defer_check=*/true));  // settles box_valid (optimistically: scan_fused_device checks), which decides whether the raw stream will be duplicate-free
        const uint32_t idv = L->id_or ^ L->id_and;
        const RadixPlan plan = pair_sort_plan(((unsigned long long)idv << 32) | idv,
                                              !no_source_dedup && L->box_valid && L->box_all_known, L->grp_next, L->grp_next_dups);
        // the sort may go out ahead of the scan's sizes when this frame looks like the last one (bp_layer::AheadHint)
        static const bool no_ahead = getenv("BP_NO_SORT_AHEAD") != nullptr;  // A/B switch for measurements
        const bp_layer::AheadHint& h = L->ahead;
        const unsigned long long varying = ((unsigned long long)idv << 32) | idv;
        L->ahead_done = false;
        L->ahead_try = !no_ahead && h.valid && !no_source_dedup && L->box_valid && L->box_all_known && L->grp_next && L->grp_off <= 1 &&
                       h.varying == varying && h.first_lo == L->id_and && h.first_hi == L->id_or && same_plan(h.plan, plan) &&
                       L->m <= h.m + h.m / 16 && L->m + L->m / 16 >= h.m &&
                       group_finish_groups(h.n_raw, varying, L->id_and, L->id_or) == h.n_groups && h.n_groups != 0;
        BP_TRY(scan_fused_device(L, coarse_depth, emit_coarse, !no_source_dedup, nullptr, 0, true, &plan));
        L->ahead_try = false;
        if (L->ahead_done) { L->ahead_done = false; return BP_OK; }
        BP_TRY(sort_unique_pairs(L, L->n_raw, L->id_or ^ L->id_and, L->raw_is_unique, L->raw_holes, true));
        if (L->n_raw == 0) BP_TRY(sync_stream(L));
        return BP_OK;
    }
    static const bool no_expand = getenv("BP_NO_EXPAND_SCAN") != nullptr;
    if (!L->ids_distinct && !no_fused && !no_expand) {  // ids may repeat: mark + emit expanding passes, then sort + dedup
        int done = 0;
        BP_TRY(scan_general_expand_device(L, coarse_depth, emit_coarse, &done));
        if (done) {
            BP_TRY(sort_unique_pairs(L, L->n_raw, L->id_or ^ L->id_and, false, L->raw_holes, true));
            if (L->n_raw == 0) BP_TRY(sync_stream(L));
            return BP_OK;
        }
    }
    BP_TRY(scan_raw_device(L, coarse_depth, emit_coarse, !no_source_dedup));
    BP_TRY(sort_unique_pairs(L, L->n_raw, L->id_or ^ L->id_and, L->raw_is_unique, false, true));
    if (L->n_raw == 0) BP_TRY(sync_stream(L));
    return BP_OK;
}

int pairs_to_host(bp_layer* L, const uint32_t** pairs, size_t* n_pairs) {
    const size_t bytes = L->n_pairs * 2 * sizeof(uint32_t);
    BP_TRY(ensure_host(L, L->h_pairs, std::max<size_t>(bytes, 8)));
    if (bytes) {
        HIP_TRY(L, hipMemcpyAsync(L->h_pairs.p, L->out_pairs.p, bytes, hipMemcpyDeviceToHost, L->stream));
        BP_TRY(sync_stream(L));
    }
    *pairs = (const uint32_t*)L->h_pairs.p;
    *n_pairs = L->n_pairs;
    return BP_OK;
}

bool valid_kind(int k) { return k >= 0 && k <= 2; }

}  // namespace

// ======================================================================================================
// C ABI
// ======================================================================================================
extern "C" {

const char* bp_version(void) { return "broadphase_amd 0.1 gfx950"; }

const char* bp_status_string(int status) {
    switch (status) {
        case BP_OK: return "BP_OK";
        case BP_ERR_INVALID_ARG: return "BP_ERR_INVALID_ARG";
        case BP_ERR_NO_DEVICE: return "BP_ERR_NO_DEVICE";
        case BP_ERR_HIP: return "BP_ERR_HIP";
        case BP_ERR_OOM: return "BP_ERR_OOM";
        case BP_ERR_TOO_LARGE: return "BP_ERR_TOO_LARGE";
        case BP_ERR_KERNEL: return "BP_ERR_KERNEL";
        case BP_ERR_COMM: return "BP_ERR_COMM";
        default: return "BP_ERR_UNKNOWN";
    }
}

const char* bp_layer_last_error(const bp_layer_t* layer) {
    return layer ? layer->err.c_str() : g_last_global_error.c_str();
}

int bp_layer_create(bp_layer_t** out, int index_kind, uint32_t min_depth, size_t index_capacity,
                    size_t collision_capacity, int device) {
    if (!out || !valid_kind(index_kind)) return fail(nullptr, BP_ERR_INVALID_ARG, "bp_layer_create: bad argument");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(nullptr, BP_ERR_NO_DEVICE, "no HIP device available (no CPU fallback exists)", e);
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return fail(nullptr, BP_ERR_NO_DEVICE, "hipGetDevice failed");
    }
    if (device >= count) return fail(nullptr, BP_ERR_INVALID_ARG, "bp_layer_create: device ordinal out of range");
    bp_layer* L = new (std::nothrow) bp_layer();
    if (!L) return fail(nullptr, BP_ERR_OOM, "out of host memory");
    L->kind = index_kind;
    L->min_depth = min_depth;
    L->device = device;
    DeviceGuard g(device);
    if (!g.ok) { delete L; return fail(nullptr, BP_ERR_NO_DEVICE, "hipSetDevice failed"); }
    e = hipStreamCreateWithFlags(&L->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete L; return fail(nullptr, BP_ERR_HIP, "hipStreamCreateWithFlags", e); }
    int rc = ensure_dev(L, L->d_sc, SC_BLOCK_BYTES);
    if (rc == BP_OK) rc = ensure_host(L, L->h_sc, SC_BLOCK_BYTES);
    if (rc == BP_OK && index_capacity) {
        rc = ensure_dev(L, L->keys[0], index_capacity * sizeof(uint64_t));
        if (rc == BP_OK) rc = ensure_dev(L, L->ids[0], index_capacity * sizeof(uint32_t));
    }
    if (rc == BP_OK && collision_capacity) rc = ensure_dev(L, L->out_pairs, collision_capacity * 2 * sizeof(uint32_t));
    // Both capacities given (LayerBuilder::with_index_capacity + with_collision_capacity, src/layer.rs:652-668: "pre-size the
    // vectors"): everything a frame of that size allocates on the way is allocated here, so that a layer's FIRST frame -- what a
    // reference test or a one-shot caller runs -- pays no hipMalloc (0.05 - 0.1 ms each, ~0.4 ms per 10 M-object frame).  Sizes
    // mirror the requests of extend / sort_tree / the scan; a request that turns out larger simply grows its buffer as before.
    if (rc == BP_OK && index_capacity && collision_capacity) {
        const size_t m = index_capacity, pc = collision_capacity;
        const size_t raw_slots = std::max<size_t>(6 * m, pc + pc / 4 + 4096);   // the scan's block-structured output (scan_fused_device)
        auto want = [&](DevBuf& b, size_t bytes) { if (rc == BP_OK) rc = ensure_dev(L, b, bytes); };
        want(L->keys[1], m * sizeof(uint64_t));
        want(L->ids[1], m * sizeof(uint32_t));
        want(L->box[0], m); want(L->box[1], m);
        want(L->idctx, (m + SCAN_PAD_ENTRIES) * sizeof(uint64_t));
        want(L->dcnt, (m + SCAN_PAD_ENTRIES) * sizeof(uint16_t));
        want(L->big_list, std::max<size_t>(65536, m / 16) * 2 * sizeof(uint64_t));
        want(L->pairs[0], raw_slots * sizeof(uint64_t));
        want(L->pairs[1], (pc + pc / 8 + 4096) * sizeof(uint64_t));
        want(L->out_pairs, (pc + pc / 8 + 4096) * 2 * sizeof(uint32_t));  // (the finish asks for the margin the NEXT frame's sort ahead will want)
        want(L->grp_bounds, ((size_t)CELL_COUNT + 2) * sizeof(uint32_t));
        want(L->tgrp_bounds, ((size_t)CELL_COUNT + 2) * sizeof(uint32_t));
        want(L->blockfill, (raw_slots / SX_BLOCK + 1) * sizeof(uint32_t));
        want(L->pair_hist, PAIR_HIST_WORDS * sizeof(unsigned long long));
        want(L->tree_hist, (size_t)HIST_COPIES * SX_HIST_MAX * RS_BINS * sizeof(unsigned long long));
        if (rc == BP_OK) {   // (the radix passes' status words carry launch tags: zero-filled once per allocation -- here, not in the first frame)
            uint32_t* unused = nullptr;
            rc = radix_ws_tagged(L, std::max(radix_workspace_bytes(m, 3), radix_workspace_bytes(raw_slots, 2)), &unused);
        }
    }
    if (rc != BP_OK) {
        g_last_global_error = L->err;
        bp_layer_destroy(L);
        return rc;
    }
    *out = L;
    return BP_OK;
}

int bp_layer_destroy(bp_layer_t* L) {
    if (!L) return BP_OK;
    DeviceGuard g(L->device);
    if (L->stream) (void)hipStreamSynchronize(L->stream);
    prof_collect(L);
    for (auto e : L->prof_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) { free_dev(L->keys[i]); free_dev(L->ids[i]); free_dev(L->pairs[i]); free_dev(L->box[i]); }
    free_dev(L->in_bounds); free_dev(L->in_ids); free_dev(L->counts); free_dev(L->block_sums); free_dev(L->aux);
    free_dev(L->skip); free_dev(L->out_pairs); free_dev(L->radix_ws); free_dev(L->d_sc);
    free_dev(L->idctx); free_dev(L->dcnt); free_dev(L->big_list); free_dev(L->blockfill); free_dev(L->shards); free_dev(L->tree_hist);
    free_dev(L->pair_hist); free_dev(L->cell_ws); free_dev(L->tcell_keys); free_dev(L->tcell_ids); free_dev(L->tcell_ws); free_dev(L->qtab); free_dev(L->cell_hist_dev); free_dev(L->grp_bounds); free_dev(L->d_sort_sc); free_dev(L->tgrp_bounds); free_dev(L->cell_sum);
    free_host(L->h_sc); free_host(L->h_pairs); free_host(L->h_keys); free_host(L->h_ids); free_host(L->h_raw); free_host(L->h_cells);
    free_host(L->h_shards); free_host(L->h_sort_sc); free_host(L->h_part_err); free_dev(L->d_part_err);
    if (L->early_ev) (void)hipEventDestroy(L->early_ev);
    if (L->pub) (void)hipHostFree(L->pub);
    if (L->fence_ev) (void)hipEventDestroy(L->fence_ev);
    if (L->stream) (void)hipStreamDestroy(L->stream);
    delete L;
    return BP_OK;
}

// Stream-ordering contract (include/bp_layer.h): the layer works on its own non-blocking stream, so device
// inputs written by kernels still in flight on another stream must be fenced in explicitly.
int bp_layer_wait_stream(bp_layer_t* L, void* producer_stream) {
    if (!L) return BP_ERR_INVALID_ARG;
    DeviceGuard g(L->device);
    // a producer stream with nothing in flight needs no fence (the per-frame case: the boxes were written long ago) -- an event
    // wait in the layer's stream costs the frame's first kernel a few microseconds
    {
        const hipError_t q = hipStreamQuery((hipStream_t)producer_stream);
        if (q == hipSuccess) return BP_OK;
        if (q != hipErrorNotReady) (void)hipGetLastError();
    }
    if (!L->fence_ev) HIP_TRY(L, hipEventCreateWithFlags(&L->fence_ev, hipEventDisableTiming));
    HIP_TRY(L, hipEventRecord(L->fence_ev, (hipStream_t)producer_stream));
    HIP_TRY(L, hipStreamWaitEvent(L->stream, L->fence_ev, 0));
    return BP_OK;
}
// forget what earlier sorts learned about the scene (grouped tree / pair finish hints and back-offs): the next frame
// takes the plan of a layer's first frame.  Results never depend on the hints; only measurements need this.
int bp_layer_reset_plan_hints(bp_layer_t* L) {
    if (!L) return BP_ERR_INVALID_ARG;
    L->tgrp_ok = false; L->tgrp_known = false; L->tgrp_probe = 0;
    L->grp_next = true; L->grp_off = 0; L->grp_off_dups = 0; L->grp_next_dups = 0;
    L->enc_hint.valid = false;
    L->ahead.valid = false; L->cell_off = 0;
    L->tcell.valid = false; L->tcell_off = 0;
    L->scan_listed = 0;
    return BP_OK;
}
int bp_layer_stream(const bp_layer_t* L, void** stream) {
    if (!L || !stream) return BP_ERR_INVALID_ARG;
    *stream = (void*)L->stream;
    return BP_OK;
}

// A deferred tree sort whose tree is being dropped (clear): its result needs no check and no repair -- but what its scalars say about
// the NEXT sort's plan (do the groups fit the LDS finish? did the cells hold?) is exactly what a frame loop without a scan between its
// sorts would otherwise never learn: it stayed on a layer's first, optimistic plan for ever (one host round trip here instead;
// a loop that scans has settled the sort on the scan's round trip and does not come through here)
static void learn_from_dropped_sort(bp_layer* L) {
    if (!L->sort_pend.on) return;
    DeviceGuard g(L->device);
    const bool got = sort_sc_fetch(L) == BP_OK && sync_stream(L) == BP_OK && L->sort_pend.copied;
    L->sort_pend.on = false;
    if (!got) return;  // (best effort: the hints stay what they were)
    const DevScalars sc = *(const DevScalars*)L->h_sort_sc.p;
    const bool tgrp_before = L->tgrp_ok;
    const bool fits = sc.grp_max <= (uint32_t)tgroup_capacity();
    L->tgrp_known = true;
    L->tgrp_ok = fits;
    if (sc.err & KERR_SEGMENT_TOO_LONG) {
        if (!(L->sort_pend.cells && fits)) L->tgrp_ok = false;
        if (L->sort_pend.cells) L->tcell_off = 16;
        L->tcell.valid = false;
    }
    if ((sc.err & KERR_SEGMENT_TOO_LONG) || L->tgrp_ok != tgrp_before) ++L->stats.tree_plan_flips;
}

int bp_layer_clear(bp_layer_t* L) {
    if (!L) return BP_ERR_INVALID_ARG;
    learn_from_dropped_sort(L);  // (its kernels stay ordered on the stream either way)
    L->tree_hist_valid = false; L->cell_hist_valid = false; L->cell_dev_valid = false;
    L->tail_run = false;
    ++L->prof_frame;
    L->m = 0;
    L->sorted = true;  // src/layer.rs:84-88
    reset_tree_facts(L);
    L->stats.entries = 0;
    L->stats.objects_valid = L->stats.objects_invalid = 0;
    return BP_OK;
}

int bp_layer_extend_device(bp_layer_t* L, const float* sys_min, const float* sys_max, const float* d_bounds,
                           const uint32_t* d_ids, size_t n) {
    if (!L || !sys_min || !sys_max || (n && (!d_bounds || !d_ids))) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_extend: null argument");
    DeviceGuard g(L->device);
    return extend_device(L, sys_min, sys_max, d_bounds, d_ids, n);
}

int bp_layer_extend_host(bp_layer_t* L, const float* sys_min, const float* sys_max, const float* bounds,
                         const uint32_t* ids, size_t n) {
    if (!L || !sys_min || !sys_max || (n && (!bounds || !ids))) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_extend: null argument");
    if (n == 0) return BP_OK;
    DeviceGuard g(L->device);
    const KindInfo ki = kind_info(L->kind);
    const size_t bb = n * 2 * ki.dim * sizeof(float), ib = n * sizeof(uint32_t);
    BP_TRY(ensure_dev(L, L->in_bounds, bb));
    BP_TRY(ensure_dev(L, L->in_ids, ib));
    HIP_TRY(L, hipMemcpyAsync(L->in_bounds.p, bounds, bb, hipMemcpyHostToDevice, L->stream));
    HIP_TRY(L, hipMemcpyAsync(L->in_ids.p, ids, ib, hipMemcpyHostToDevice, L->stream));
    return extend_device(L, sys_min, sys_max, (const float*)L->in_bounds.p, (const uint32_t*)L->in_ids.p, n);
}

int bp_layer_extend(bp_layer_t* L, const float* sys_min, const float* sys_max, const float* bounds,
                    const uint32_t* ids, size_t n) {
    if (!L) return BP_ERR_INVALID_ARG;
    if (n == 0) return BP_OK;
    DeviceGuard g(L->device);
    const bool db = is_device_ptr(bounds), di = is_device_ptr(ids);
    if (db != di) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_extend: bounds and ids must both be host or both be device pointers");
    return db ? bp_layer_extend_device(L, sys_min, sys_max, bounds, ids, n)
              : bp_layer_extend_host(L, sys_min, sys_max, bounds, ids, n);
}

int bp_layer_append_entries(bp_layer_t* L, const uint64_t* keys, const uint32_t* ids, size_t m) {
    if (!L || (m && (!keys || !ids))) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_append_entries: null argument");
    if (m == 0) return BP_OK;
    DeviceGuard g(L->device);
    const bool dk = is_device_ptr(keys), di = is_device_ptr(ids);
    if (dk != di) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_append_entries: keys and ids must live in the same memory space");
    BP_TRY(ensure_unpacked(L));
    BP_TRY(ensure_tree_capacity(L, L->m + m));
    const hipMemcpyKind kind = dk ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    HIP_TRY(L, hipMemcpyAsync(cur_keys(L) + L->m, keys, m * sizeof(uint64_t), kind, L->stream));
    HIP_TRY(L, hipMemcpyAsync(cur_ids(L) + L->m, ids, m * sizeof(uint32_t), kind, L->stream));
    HIP_TRY(L, hipMemsetAsync((uint8_t*)L->box[L->cur].p + L->m, (int)BOX_UNKNOWN, m, L->stream));
    L->box_all_known = false;
    BP_TRY(absorb_raw_entries(L, cur_keys(L) + L->m, cur_ids(L) + L->m, m));
    L->tree_hist_valid = false; L->cell_hist_valid = false; L->cell_dev_valid = false;
    L->tail_run = false;
    L->m += m;
    L->sorted = false;
    L->stats.entries = L->m;
    return BP_OK;
}

int bp_layer_merge(bp_layer_t* L, const bp_layer_t* other) {
    if (!L || !other) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_merge: null argument");
    if (L->kind != other->kind) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_merge: index kinds differ");
    if (L->device != other->device) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_merge: layers live on different devices");
    DeviceGuard g(L->device);
    BP_TRY(ensure_unpacked(L));
    BP_TRY(ensure_unpacked(const_cast<bp_layer*>(other)));
    if (other->min_depth < L->min_depth) {  // src/layer.rs:131-134 (warn! + lower)
        fprintf(stderr, "[broadphase_amd] warning: merging layer of lesser min_depth (lhs: %u, rhs: %u)\n", L->min_depth,
                other->min_depth);
        L->min_depth = other->min_depth;
    }
    const size_t m = other->m;
    if (m) {
        (void)hipStreamSynchronize(other->stream);
        BP_TRY(ensure_tree_capacity(L, L->m + m));
        const int oc = other->cur;
        HIP_TRY(L, launch_copy_bytes(cur_keys(L) + L->m, other->keys[oc].p, m * sizeof(uint64_t), L->stream));
        HIP_TRY(L, launch_copy_bytes(cur_ids(L) + L->m, other->ids[oc].p, m * sizeof(uint32_t), L->stream));
        if (other->box_valid && other->box[oc].p) {
            HIP_TRY(L, launch_copy_bytes((uint8_t*)L->box[L->cur].p + L->m, other->box[oc].p, m, L->stream));
            L->box_all_known = L->box_all_known && other->box_all_known;
        } else {
            HIP_TRY(L, hipMemsetAsync((uint8_t*)L->box[L->cur].p + L->m, (int)BOX_UNKNOWN, m, L->stream));
            L->box_all_known = false;
        }
        BP_TRY(sync_stream(L));
        // facts: the union keeps whatever both sides can still prove.  The crate's intended pattern
        // -- a static layer merged into each frame's dynamic layer (src/layer.rs:123-126) -- has
        // disjoint id ranges, which keeps ids provably distinct and the fast kernels selected.
        const bool was_empty = (L->m == 0);
        L->key_or |= other->key_or; L->key_and &= other->key_and;
        L->id_or |= other->id_or; L->id_and &= other->id_and;
        L->depth_mask |= other->depth_mask;
        if (was_empty) {
            L->order_ok = other->order_ok; L->ids_distinct = other->ids_distinct;
            L->max_id = other->max_id; L->min_id = other->min_id; L->have_ids = other->have_ids;
        } else {
            const bool other_after = L->have_ids && other->have_ids && L->max_id < other->min_id;
            const bool other_before = L->have_ids && other->have_ids && other->max_id < L->min_id;
            L->ids_distinct = L->ids_distinct && other->ids_distinct && (other_after || other_before);
            L->order_ok = L->order_ok && other->order_ok && other_after;  // appended ids must not go down
            L->max_id = std::max(L->max_id, other->max_id);
            L->min_id = std::min(L->min_id, other->min_id);
            L->have_ids = L->have_ids || other->have_ids;
        }
        L->tree_hist_valid = false; L->cell_hist_valid = false; L->cell_dev_valid = false;
        // the appended entries form one sorted run if `other` was sorted (and no earlier run is pending)
        const bool had_run = L->tail_run;
        L->tail_run = !had_run && other->sorted;
        L->run_split = L->m;
        L->m += m;
    }
    L->sorted = false;  // unconditional: src/layer.rs:137
    L->stats.entries = L->m;
    return BP_OK;
}

int bp_layer_sort(bp_layer_t* L) {
    if (!L) return BP_ERR_INVALID_ARG;
    DeviceGuard g(L->device);
    return sort_tree(L, /*defer_check=*/true);  // completion and the check fold into the next call that needs the tree
}
int bp_layer_par_sort(bp_layer_t* L) { return bp_layer_sort(L); }

int bp_layer_scan_device(bp_layer_t* L, const uint32_t** d_pairs, size_t* n_pairs) {
    if (!L || !d_pairs || !n_pairs) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan: null argument");
    DeviceGuard g(L->device);
    BP_TRY(scan_device(L, 0, 1));
    *d_pairs = (const uint32_t*)L->out_pairs.p;
    *n_pairs = L->n_pairs;
    return BP_OK;
}

int bp_layer_scan_shard_device(bp_layer_t* L, uint32_t coarse_depth, int emit_coarse, const uint32_t** d_pairs,
                               size_t* n_pairs) {
    if (!L || !d_pairs || !n_pairs) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_shard: null argument");
    DeviceGuard g(L->device);
    BP_TRY(scan_device(L, coarse_depth, emit_coarse));
    *d_pairs = (const uint32_t*)L->out_pairs.p;
    *n_pairs = L->n_pairs;
    return BP_OK;
}

int bp_layer_scan(bp_layer_t* L, const uint32_t** pairs, size_t* n_pairs) {
    if (!L || !pairs || !n_pairs) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan: null argument");
    DeviceGuard g(L->device);
    BP_TRY(scan_device(L, 0, 1));
    return pairs_to_host(L, pairs, n_pairs);
}
int bp_layer_par_scan(bp_layer_t* L, const uint32_t** pairs, size_t* n_pairs) { return bp_layer_scan(L, pairs, n_pairs); }

int bp_layer_scan_raw(bp_layer_t* L, const uint32_t** pairs, size_t* n_raw) {
    if (!L || !pairs || !n_raw) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_raw: null argument");
    DeviceGuard g(L->device);
    BP_TRY(scan_raw_device(L, 0, 1));
    const size_t n = L->n_raw;
    BP_TRY(ensure_host(L, L->h_raw, std::max<size_t>(n * 2 * sizeof(uint32_t), 8)));
    if (n) {
        // unpack into the (not yet needed) second pair buffer as interleaved u32, then copy out
        BP_TRY(ensure_dev(L, L->pairs[1], n * sizeof(uint64_t)));
        HIP_TRY(L, launch_unpack_pairs((const uint64_t*)L->pairs[0].p, n, (uint32_t*)L->pairs[1].p, L->stream));
        HIP_TRY(L, hipMemcpyAsync(L->h_raw.p, L->pairs[1].p, n * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, L->stream));
    }
    BP_TRY(sync_stream(L));
    *pairs = (const uint32_t*)L->h_raw.p;
    *n_raw = n;
    return BP_OK;
}

int bp_layer_scan_filtered(bp_layer_t* L, bp_filter_fn filter, void* user, const uint32_t** pairs, size_t* n_pairs) {
    if (!L || !pairs || !n_pairs) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_filtered: null argument");
    if (!filter) return bp_layer_scan(L, pairs, n_pairs);
    DeviceGuard g(L->device);
    const uint32_t* raw = nullptr;
    size_t n = 0;
    BP_TRY(bp_layer_scan_raw(L, &raw, &n));
    // the filter runs on the host over the raw stream, in the reference's call order (src/layer.rs:566-570)
    uint32_t* w = (uint32_t*)L->h_raw.p;
    size_t kept = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t a = raw[2 * i], b = raw[2 * i + 1];
        if (filter(a, b, user)) {
            w[2 * kept] = a;
            w[2 * kept + 1] = b;
            ++kept;
        }
    }
    if (kept) {
        BP_TRY(ensure_dev(L, L->pairs[1], kept * sizeof(uint64_t)));
        HIP_TRY(L, hipMemcpyAsync(L->pairs[1].p, w, kept * 2 * sizeof(uint32_t), hipMemcpyHostToDevice, L->stream));
        BP_TRY(ensure_dev(L, L->pairs[0], kept * sizeof(uint64_t)));
        HIP_TRY(L, launch_pack_pairs((const uint32_t*)L->pairs[1].p, kept, (uint64_t*)L->pairs[0].p, L->stream));
    }
    L->n_raw = kept;
    BP_TRY(sort_unique_pairs(L, kept, L->id_or ^ L->id_and));
    if (kept == 0) BP_TRY(sync_stream(L));
    return pairs_to_host(L, pairs, n_pairs);
}
int bp_layer_par_scan_filtered(bp_layer_t* L, bp_filter_fn filter, void* user, const uint32_t** pairs, size_t* n_pairs) {
    return bp_layer_scan_filtered(L, filter, user, pairs, n_pairs);
}

// scan_filtered with a built-in narrow-phase predicate over per-id rows (boxes or spheres); result in L->out_pairs
static int scan_filtered_rows(bp_layer* L, const float* bounds, size_t n_rows, int shape) {
    const KindInfo ki = kind_info(L->kind);
    const float* d_bounds = bounds;
    if (!is_device_ptr(bounds)) {
        const size_t bytes = n_rows * (shape == FILT_SPHERE ? ki.dim + 1 : 2 * ki.dim) * sizeof(float);
        BP_TRY(ensure_dev(L, L->in_bounds, bytes));
        HIP_TRY(L, hipMemcpyAsync(L->in_bounds.p, bounds, bytes, hipMemcpyHostToDevice, L->stream));
        d_bounds = (const float*)L->in_bounds.p;
    }
    if (L->ids_distinct) {  // predicate evaluated inside the single-pass scan, before anything is sorted
        // (deferred like scan_device's: a sort still pending is settled on the scan's own round trip -- settling it here cost a host
        // round trip AND materialised the id / box arrays the compact sort had left to the context words: 0.12 ms per 10 M frame)
        BP_TRY(sort_tree(L, /*defer_check=*/true));
        const uint32_t idv = L->id_or ^ L->id_and;
        const RadixPlan plan = pair_sort_plan(((unsigned long long)idv << 32) | idv, L->box_valid && L->box_all_known);
        BP_TRY(scan_fused_device(L, 0, 1, true, d_bounds, (uint32_t)n_rows, true, &plan));
        BP_TRY(sort_unique_pairs(L, L->n_raw, L->id_or ^ L->id_and, L->raw_is_unique, L->raw_holes, true));
        if (L->n_raw == 0) BP_TRY(sync_stream(L));
        return BP_OK;
    }
    // literal path: raw stream (skip rule applied), then the predicate as a compaction pass
    BP_TRY(scan_raw_device(L, 0, 1));
    const size_t n = L->n_raw;
    size_t kept = 0;
    if (n) {
        BP_TRY(ensure_dev(L, L->pairs[1], n * sizeof(uint64_t)));
        BP_TRY(sc_reset(L));
        {
            ProfScope ps(L, BP_K_MISC);
            HIP_TRY(L, launch_filter_pairs_aabb((int)ki.dim, (const uint64_t*)L->pairs[0].p, n, d_bounds, shape, (uint64_t*)L->pairs[1].p,
                                                (DevScalars*)L->d_sc.p, L->stream));
        }
        DevScalars sc;
        BP_TRY(sc_fetch(L, &sc));
        kept = (size_t)sc.total;
    }
    L->n_raw = kept;
    BP_TRY(sort_unique_pairs_in(L, (uint64_t*)L->pairs[1].p, kept, (((unsigned long long)(L->id_or ^ L->id_and)) << 32) | (L->id_or ^ L->id_and), false));
    if (kept == 0) BP_TRY(sync_stream(L));
    return BP_OK;
}
static int scan_filtered_aabb_impl(bp_layer* L, const float* bounds, size_t n_rows, int shape = FILT_AABB) {
    if (!bounds) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_filtered_aabb/sphere: null rows");
    if (L->m && (!L->have_ids || L->max_id == 0xFFFFFFFFu || (size_t)L->max_id >= n_rows))
        return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_filtered_aabb/sphere: need one row per id (ids must be bounded by n_rows)");
    L->filt_shape = shape;
    const int rc = scan_filtered_rows(L, bounds, n_rows, shape);
    L->filt_shape = FILT_AABB;
    return rc;
}

int bp_layer_scan_filtered_aabb_device(bp_layer_t* L, const float* bounds_by_id, size_t n_rows, const uint32_t** d_pairs, size_t* n_pairs) {
    if (!L || !d_pairs || !n_pairs) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_filtered_aabb: null argument");
    DeviceGuard g(L->device);
    BP_TRY(scan_filtered_aabb_impl(L, bounds_by_id, n_rows));
    *d_pairs = (const uint32_t*)L->out_pairs.p;
    *n_pairs = L->n_pairs;
    return BP_OK;
}

int bp_layer_scan_filtered_aabb(bp_layer_t* L, const float* bounds_by_id, size_t n_rows, const uint32_t** pairs, size_t* n_pairs) {
    if (!L || !pairs || !n_pairs) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_filtered_aabb: null argument");
    DeviceGuard g(L->device);
    BP_TRY(scan_filtered_aabb_impl(L, bounds_by_id, n_rows));
    return pairs_to_host(L, pairs, n_pairs);
}

int bp_layer_scan_filtered_sphere_device(bp_layer_t* L, const float* spheres_by_id, size_t n_rows, const uint32_t** d_pairs, size_t* n_pairs) {
    if (!L || !d_pairs || !n_pairs) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_filtered_sphere: null argument");
    DeviceGuard g(L->device);
    BP_TRY(scan_filtered_aabb_impl(L, spheres_by_id, n_rows, FILT_SPHERE));
    *d_pairs = (const uint32_t*)L->out_pairs.p;
    *n_pairs = L->n_pairs;
    return BP_OK;
}

int bp_layer_scan_filtered_sphere(bp_layer_t* L, const float* spheres_by_id, size_t n_rows, const uint32_t** pairs, size_t* n_pairs) {
    if (!L || !pairs || !n_pairs) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_filtered_sphere: null argument");
    DeviceGuard g(L->device);
    BP_TRY(scan_filtered_aabb_impl(L, spheres_by_id, n_rows, FILT_SPHERE));
    return pairs_to_host(L, pairs, n_pairs);
}

// Batched Layer::test_box / test_ray; result (query, id) rows in L->out_pairs, L->n_pairs
static int region_query_batch_impl(bp_layer* L, int ray, const float* sys_min, const float* sys_max, const float* boxes,
                                   size_t n_queries, int has_max_depth, uint32_t max_depth) {
    if (!sys_min || !sys_max || (n_queries && !boxes)) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_test_*_batch: null argument");
    if (n_queries > 0x7FFFFFFFull) return fail(L, BP_ERR_TOO_LARGE, "bp_layer_test_box_batch: more than 2^31-1 queries");
    BP_TRY(sort_tree(L));  // src/layer.rs:260 (Layer::test sorts first)
    L->n_pairs = 0;
    L->n_raw = 0;
    L->raw_holes = false;
    L->pair_hist_valid = false;
    if (n_queries == 0 || L->m == 0) return sync_stream(L);
    if (L->m >= 0xFFFFFFFFull) return fail(L, BP_ERR_TOO_LARGE, "test_box: more than 2^32-1 index entries");
    const KindInfo ki = kind_info(L->kind);
    SysBounds sb;
    for (uint32_t a = 0; a < 3; ++a) {
        sb.mn[a] = a < ki.dim ? sys_min[a] : 0.f;
        sb.mx[a] = a < ki.dim ? sys_max[a] : 0.f;
    }
    const float* d_boxes = boxes;
    if (!is_device_ptr(boxes)) {
        const size_t bytes = n_queries * (2 * ki.dim + (ray ? 2 : 0)) * sizeof(float);
        BP_TRY(ensure_dev(L, L->in_bounds, bytes));
        HIP_TRY(L, hipMemcpyAsync(L->in_bounds.p, boxes, bytes, hipMemcpyHostToDevice, L->stream));
        d_boxes = (const float*)L->in_bounds.p;
    }
    // slice bounds of the top levels (<= 2^18 cells), once per sorted tree; only worth it for a batch
    const uint32_t want_levels = n_queries >= 4096 ? std::min<uint32_t>(18u / ki.dim, ki.axis_bits) : 0u;
    if (want_levels && (L->qtab_epoch != L->epoch || L->qtab_levels != want_levels)) {
        BP_TRY(ensure_dev(L, L->qtab, (size_t)query_table_entries(L->kind, want_levels) * sizeof(uint32_t)));
        ProfScope ps(L, BP_K_MISC);
        HIP_TRY(L, launch_query_table(L->kind, cur_keys(L), L->m, want_levels, (uint32_t*)L->qtab.p, L->stream));
        L->qtab_epoch = L->epoch; L->qtab_levels = want_levels;
    }
    const size_t shard_bytes = 3 * SX_SHARDS * sizeof(uint64_t);
    BP_TRY(ensure_dev(L, L->shards, shard_bytes));
    BP_TRY(ensure_host(L, L->h_shards, shard_bytes));
    // every wave ends with one partial block: room for that plus a first guess of 64 ids per query
    const size_t n_waves = (n_queries + 63) / 64;
    size_t cap = std::max<size_t>(L->pairs[0].cap / sizeof(uint64_t), (n_waves + SX_SHARDS * SX_RESERVE) * SX_BLOCK + 64 * n_queries);
    for (int attempt = 0; attempt < 4; ++attempt) {
        BP_TRY(ensure_dev(L, L->pairs[0], cap * sizeof(uint64_t)));
        cap = L->pairs[0].cap / sizeof(uint64_t);
        const size_t out_blocks = cap / SX_BLOCK;
        BP_TRY(ensure_dev(L, L->blockfill, out_blocks * sizeof(uint32_t)));
        HIP_TRY(L, hipMemsetAsync(L->blockfill.p, 0, out_blocks * sizeof(uint32_t), L->stream));
        HIP_TRY(L, hipMemsetAsync(L->shards.p, 0, shard_bytes, L->stream));
        {
            ProfScope ps(L, BP_K_SCAN_EMIT);
            HIP_TRY(L, launch_region_query(L->kind, ray, cur_keys(L), cur_ids(L), L->m, sb, d_boxes, (uint32_t)n_queries,
                                           has_max_depth, max_depth, (uint64_t*)L->pairs[0].p, out_blocks,
                                           (uint32_t*)L->blockfill.p, (uint64_t*)L->shards.p, L->stream,
                                           want_levels ? (const uint32_t*)L->qtab.p : nullptr, want_levels));
        }
        HIP_TRY(L, hipMemcpyAsync(L->h_shards.p, L->shards.p, shard_bytes, hipMemcpyDeviceToHost, L->stream));
        BP_TRY(sync_stream(L));
        const uint64_t* sh = (const uint64_t*)L->h_shards.p;
        uint64_t kmax = 0, total = 0;
        for (int i = 0; i < SX_SHARDS; ++i) { kmax = std::max<uint64_t>(kmax, sh[i]); total += sh[SX_SHARDS + i]; }
        const size_t extent_blocks = (size_t)kmax * SX_SHARDS;
        if (extent_blocks > out_blocks) {  // the buffer was too small: grow and redo
            cap = extent_blocks * SX_BLOCK + extent_blocks * SX_BLOCK / 16;
            continue;
        }
        L->n_raw = (size_t)total;
        L->raw_holes = true;
        L->raw_extent = extent_blocks * SX_BLOCK;
        // sort by (query, id) + dedup = results.sort(); results.dedup() per query (src/layer.rs:274-275)
        unsigned long long qv = 0;
        for (size_t v = n_queries - 1; v; v >>= 1) qv = (qv << 1) | 1ull;
        const unsigned long long varying = (qv << 32) | (unsigned long long)(L->id_or ^ L->id_and);
        const bool holes = L->raw_holes;
        L->raw_holes = false;
        BP_TRY(sort_unique_pairs_in(L, (uint64_t*)L->pairs[0].p, L->n_raw, varying, false, holes));
        if (L->n_raw == 0) BP_TRY(sync_stream(L));
        return BP_OK;
    }
    return fail(L, BP_ERR_KERNEL, "test_box: output size did not converge");
}

int bp_layer_test_box_batch_device(bp_layer_t* L, const float* sys_min, const float* sys_max, const float* boxes, size_t n_queries,
                                   int has_max_depth, uint32_t max_depth, const uint32_t** d_rows, size_t* n_rows) {
    if (!L || !d_rows || !n_rows) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_test_box_batch: null argument");
    DeviceGuard g(L->device);
    BP_TRY(region_query_batch_impl(L, 0, sys_min, sys_max, boxes, n_queries, has_max_depth, max_depth));
    *d_rows = (const uint32_t*)L->out_pairs.p;
    *n_rows = L->n_pairs;
    return BP_OK;
}

int bp_layer_test_box_batch(bp_layer_t* L, const float* sys_min, const float* sys_max, const float* boxes, size_t n_queries,
                            int has_max_depth, uint32_t max_depth, const uint32_t** rows, size_t* n_rows) {
    if (!L || !rows || !n_rows) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_test_box_batch: null argument");
    DeviceGuard g(L->device);
    BP_TRY(region_query_batch_impl(L, 0, sys_min, sys_max, boxes, n_queries, has_max_depth, max_depth));
    return pairs_to_host(L, rows, n_rows);
}

int bp_layer_test_ray_batch_device(bp_layer_t* L, const float* sys_min, const float* sys_max, const float* rays, size_t n_queries,
                                   int has_max_depth, uint32_t max_depth, const uint32_t** d_rows, size_t* n_rows) {
    if (!L || !d_rows || !n_rows) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_test_ray_batch: null argument");
    DeviceGuard g(L->device);
    BP_TRY(region_query_batch_impl(L, 1, sys_min, sys_max, rays, n_queries, has_max_depth, max_depth));
    *d_rows = (const uint32_t*)L->out_pairs.p;
    *n_rows = L->n_pairs;
    return BP_OK;
}

int bp_layer_test_ray_batch(bp_layer_t* L, const float* sys_min, const float* sys_max, const float* rays, size_t n_queries,
                            int has_max_depth, uint32_t max_depth, const uint32_t** rows, size_t* n_rows) {
    if (!L || !rows || !n_rows) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_test_ray_batch: null argument");
    DeviceGuard g(L->device);
    BP_TRY(region_query_batch_impl(L, 1, sys_min, sys_max, rays, n_queries, has_max_depth, max_depth));
    return pairs_to_host(L, rows, n_rows);
}

// Batched Layer::pick_ray with the built-in ray/AABB distance (src/layer.rs:418-446)
int bp_layer_pick_ray_aabb_batch(bp_layer_t* L, const float* sys_min, const float* sys_max, const float* rays, size_t n_queries,
                                 int has_max_depth, uint32_t max_depth, const float* bounds_by_id, size_t n_rows, float* out_dist,
                                 uint32_t* out_id) {
    if (!L || !sys_min || !sys_max || (n_queries && (!rays || !out_dist || !out_id)) || !bounds_by_id)
        return fail(L, BP_ERR_INVALID_ARG, "bp_layer_pick_ray_aabb_batch: null argument");
    if (n_queries > 0x7FFFFFFFull || n_rows > 0xFFFFFFFFull) return fail(L, BP_ERR_TOO_LARGE, "bp_layer_pick_ray_aabb_batch: too many rays / rows");
    DeviceGuard g(L->device);
    BP_TRY(sort_tree(L));  // src/layer.rs:374
    if (n_queries == 0) return sync_stream(L);
    if (L->m >= 0xFFFFFFFFull) return fail(L, BP_ERR_TOO_LARGE, "pick_ray: more than 2^32-1 index entries");
    const KindInfo ki = kind_info(L->kind);
    SysBounds sb;
    for (uint32_t a = 0; a < 3; ++a) {
        sb.mn[a] = a < ki.dim ? sys_min[a] : 0.f;
        sb.mx[a] = a < ki.dim ? sys_max[a] : 0.f;
    }
    const float* d_rays = rays;
    if (!is_device_ptr(rays)) {
        const size_t bytes = n_queries * (2 * ki.dim + 1) * sizeof(float);
        BP_TRY(ensure_dev(L, L->in_ids, bytes));
        HIP_TRY(L, hipMemcpyAsync(L->in_ids.p, rays, bytes, hipMemcpyHostToDevice, L->stream));
        d_rays = (const float*)L->in_ids.p;
    }
    const float* d_bounds = bounds_by_id;
    if (!is_device_ptr(bounds_by_id)) {
        const size_t bytes = n_rows * 2 * ki.dim * sizeof(float);
        BP_TRY(ensure_dev(L, L->in_bounds, std::max<size_t>(bytes, 4)));
        if (bytes) HIP_TRY(L, hipMemcpyAsync(L->in_bounds.p, bounds_by_id, bytes, hipMemcpyHostToDevice, L->stream));
        d_bounds = (const float*)L->in_bounds.p;
    }
    const bool out_dev = is_device_ptr(out_dist);
    if (out_dev != is_device_ptr(out_id)) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_pick_ray_aabb_batch: outputs must live in the same memory space");
    float* d_dist = out_dist;
    uint32_t* d_id = out_id;
    if (!out_dev) {
        BP_TRY(ensure_dev(L, L->counts, n_queries * (sizeof(float) + sizeof(uint32_t))));
        d_dist = (float*)L->counts.p;
        d_id = (uint32_t*)((float*)L->counts.p + n_queries);
    }
    BP_TRY(sc_reset(L));
    {
        ProfScope ps(L, BP_K_SCAN_EMIT);
        HIP_TRY(L, launch_pick_ray(L->kind, cur_keys(L), cur_ids(L), L->m, sb, d_rays, (uint32_t)n_queries, has_max_depth, max_depth,
                                   d_bounds, (uint32_t)n_rows, d_dist, d_id, (DevScalars*)L->d_sc.p, L->stream));
    }
    if (!out_dev) {
        HIP_TRY(L, hipMemcpyAsync(out_dist, d_dist, n_queries * sizeof(float), hipMemcpyDeviceToHost, L->stream));
        HIP_TRY(L, hipMemcpyAsync(out_id, d_id, n_queries * sizeof(uint32_t), hipMemcpyDeviceToHost, L->stream));
    }
    DevScalars sc;
    BP_TRY(sc_fetch(L, &sc));
    if (sc.err & KERR_BAD_ID) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_pick_ray_aabb_batch: an id of the layer has no row in bounds_by_id");
    return BP_OK;
}

int bp_layer_sort_unique_pairs_device(bp_layer_t* L, const uint32_t* d_pairs_in, size_t n, const uint32_t** d_pairs,
                                      size_t* n_pairs) {
    if (!L || !d_pairs || !n_pairs || (n && !d_pairs_in)) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_sort_unique_pairs: null argument");
    DeviceGuard g(L->device);
    if (n) {
        BP_TRY(ensure_dev(L, L->pairs[0], n * sizeof(uint64_t)));
        HIP_TRY(L, launch_pack_pairs(d_pairs_in, n, (uint64_t*)L->pairs[0].p, L->stream));
    }
    L->n_raw = n;
    BP_TRY(sort_unique_pairs(L, n, 0xFFFFFFFFu));  // nothing is known about foreign ids: all 64 bits vary
    if (n == 0) BP_TRY(sync_stream(L));
    *d_pairs = (const uint32_t*)L->out_pairs.p;
    *n_pairs = L->n_pairs;
    return BP_OK;
}

int bp_layer_append_entries_assume(bp_layer_t* L, const uint64_t* keys, const uint32_t* ids, size_t m, uint32_t facts) {
    if (!L) return BP_ERR_INVALID_ARG;
    const bool was_empty = L->m == 0;
    const bool order_before = L->order_ok, distinct_before = L->ids_distinct, known_before = L->box_all_known;
    const size_t at = L->m;
    BP_TRY(bp_layer_append_entries(L, keys, ids, m));
    if (m) {  // append_entries dropped the facts; restore what the caller vouches for
        L->order_ok = (was_empty || order_before) && (facts & BP_FACT_ORDER_OK);
        L->ids_distinct = (was_empty || distinct_before) && (facts & BP_FACT_IDS_DISTINCT);
        if (facts & BP_FACT_IDS_CARRY_BOX) {
            DeviceGuard g(L->device);
            HIP_TRY(L, launch_split_box(cur_ids(L) + at, (uint8_t*)L->box[L->cur].p + at, m, L->stream));
            BP_TRY(sync_stream(L));
            // the id OR/AND were reduced over the packed values: the code bits are not id bits
            L->id_or &= BOX_ID_MASK;
            L->id_and &= BOX_ID_MASK;
            L->box_all_known = known_before;  // codes may still be BOX_UNKNOWN: the sender says so via facts
            if (!(facts & BP_FACT_BOX_ALL_KNOWN)) L->box_all_known = false;
        }
    }
    return BP_OK;
}

int bp_layer_can_pack_box(const bp_layer_t* L, int* can) {
    if (!L || !can) return BP_ERR_INVALID_ARG;
    *can = (box_can_ride(L) ? 1 : 0) | (L->box_all_known ? 2 : 0);
    return BP_OK;
}

int bp_layer_get_facts(const bp_layer_t* L, bp_facts* out) {
    if (!L || !out) return BP_ERR_INVALID_ARG;
    // an empty layer reports neutral-but-representable AND fields (0 = "no bit is known to be set": a superset
    // description, always safe) instead of the reductions' identities ~0, which do not survive a signed 64-bit wire
    out->key_or = L->key_or; out->key_and = L->m ? L->key_and : 0ull;
    out->id_or = L->id_or; out->id_and = L->m ? L->id_and : 0u;
    out->id_min = L->min_id; out->id_max = L->max_id;
    out->depth_mask = L->depth_mask;
    out->flags = (L->order_ok ? BP_FACT_ORDER_OK : 0u) | (L->ids_distinct ? BP_FACT_IDS_DISTINCT : 0u) |
                 (box_can_ride(L) ? BP_FACT_IDS_CARRY_BOX : 0u) | (L->box_all_known ? BP_FACT_BOX_ALL_KNOWN : 0u);
    return BP_OK;
}

int bp_layer_begin_entries(bp_layer_t* L, size_t m, uint64_t** d_keys, uint32_t** d_ids) {
    if (!L || !d_keys || !d_ids) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_begin_entries: null argument");
    DeviceGuard g(L->device);
    // the buffers are handed to writers on OTHER streams: a tree sort that was only enqueued (deferred check) may still be
    // running on this layer's stream -- wait for it before exposing them
    if (L->sort_pend.on) BP_TRY(sync_stream(L));
    BP_TRY(bp_layer_clear(L));
    BP_TRY(ensure_tree_capacity(L, std::max<size_t>(m, 1)));
    *d_keys = cur_keys(L);
    *d_ids = cur_ids(L);
    return BP_OK;
}

int bp_layer_commit_entries(bp_layer_t* L, size_t m, const bp_facts* f) {
    if (!L || !f) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_commit_entries: null argument");
    if (L->m != 0) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_commit_entries: call bp_layer_begin_entries first");
    if (m * sizeof(uint64_t) > L->keys[L->cur].cap) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_commit_entries: more entries than begun");
    if (m == 0) return BP_OK;
    L->tree_hist_valid = false; L->cell_hist_valid = false; L->cell_dev_valid = false;
    L->tail_run = false;
    L->m = m;
    L->sorted = false;
    L->stats.entries = m;
    L->key_or = f->key_or; L->key_and = f->key_and;
    L->id_or = f->id_or; L->id_and = f->id_and;
    L->depth_mask = f->depth_mask;
    L->min_id = f->id_min; L->max_id = f->id_max;
    L->have_ids = true;
    L->order_ok = (f->flags & BP_FACT_ORDER_OK) != 0;
    L->ids_distinct = (f->flags & BP_FACT_IDS_DISTINCT) != 0;
    if (f->flags & BP_FACT_IDS_CARRY_BOX) {
        L->ids_packed = true;
        L->box_valid = false;  // the codes are inside the ids until the sort splits them off
        L->box_all_known = (f->flags & BP_FACT_BOX_ALL_KNOWN) != 0;
    } else {
        DeviceGuard g(L->device);
        HIP_TRY(L, hipMemsetAsync(L->box[L->cur].p, (int)BOX_UNKNOWN, m, L->stream));
        BP_TRY(sync_stream(L));
        L->box_valid = true;
        L->box_all_known = false;
    }
    return BP_OK;
}

int bp_layer_cell_histogram(bp_layer_t* L, uint32_t split_depth, uint64_t* hist, size_t n_cells) {
    if (!L || !hist) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_cell_histogram: null argument");
    const KindInfo ki = kind_info(L->kind);
    if (split_depth == 0 || split_depth > ki.axis_bits || n_cells != ((size_t)1 << (ki.dim * split_depth)) || n_cells > 4096)
        return fail(L, BP_ERR_INVALID_ARG, "bp_layer_cell_histogram: n_cells must be 2^(dim*split_depth) <= 4096");
    DeviceGuard g(L->device);
    const size_t bytes = (n_cells + 1) * sizeof(unsigned long long);
    if (L->cell_hist_valid && L->cell_hist_depth == split_depth && L->cell_hist_cache.size() == n_cells + 1) {
        std::memcpy(hist, L->cell_hist_cache.data(), bytes);  // extend brought it back with its own round trip
        return BP_OK;
    }
    if (L->cell_dev_valid && L->cell_fuse_depth == split_depth) {  // counted by extend: sum the copies, no kernel
        const size_t all = (size_t)HIST_COPIES * bytes;
        BP_TRY(ensure_host(L, L->h_raw, all));
        HIP_TRY(L, hipMemcpyAsync(L->h_raw.p, L->cell_hist_dev.p, all, hipMemcpyDeviceToHost, L->stream));
        BP_TRY(sync_stream(L));
        const unsigned long long* c = (const unsigned long long*)L->h_raw.p;
        for (size_t k = 0; k <= n_cells; ++k) {
            unsigned long long t = 0;
            for (int q = 0; q < HIST_COPIES; ++q) t += c[(size_t)q * (n_cells + 1) + k];
            hist[k] = t;
        }
        L->cell_hist_cache.assign(hist, hist + n_cells + 1);
        L->cell_hist_depth = split_depth;
        L->cell_hist_valid = true;
        return BP_OK;
    }
    BP_TRY(ensure_plain(L));
    BP_TRY(ensure_dev(L, L->block_sums, bytes));
    BP_TRY(ensure_host(L, L->h_raw, bytes));
    HIP_TRY(L, hipMemsetAsync(L->block_sums.p, 0, bytes, L->stream));
    const int cell_shift = (int)(ki.dim * ki.axis_bits + ki.depth_bits - ki.dim * split_depth);
    {
        ProfScope ps(L, BP_K_MISC);
        HIP_TRY(L, launch_cell_hist(cur_keys(L), L->m, cell_shift, (1u << ki.depth_bits) - 1u, split_depth, (uint32_t)n_cells,
                                    (unsigned long long*)L->block_sums.p, L->stream));
    }
    HIP_TRY(L, hipMemcpyAsync(L->h_raw.p, L->block_sums.p, bytes, hipMemcpyDeviceToHost, L->stream));
    BP_TRY(sync_stream(L));
    std::memcpy(hist, L->h_raw.p, bytes);
    L->cell_hist_cache.assign(hist, hist + n_cells + 1);
    L->cell_hist_depth = split_depth;
    L->cell_hist_valid = true;
    return BP_OK;
}

int bp_layer_set_cell_histogram_depth(bp_layer_t* L, uint32_t split_depth) {
    if (!L) return BP_ERR_INVALID_ARG;
    L->cell_fuse_depth = split_depth;
    L->cell_dev_valid = false;
    return BP_OK;
}

int bp_layer_partition_entries_device(bp_layer_t* L, uint32_t split_depth, const uint8_t* lut, size_t n_cells, uint32_t n_dest,
                                      uint64_t* counts, const uint64_t** d_keys, const uint32_t** d_ids) {
    return bp_layer_partition_entries_packed_device(L, split_depth, lut, n_cells, n_dest, 0, counts, d_keys, d_ids);
}

int bp_layer_partition_entries_packed_device(bp_layer_t* L, uint32_t split_depth, const uint8_t* lut, size_t n_cells, uint32_t n_dest,
                                             int flags, uint64_t* counts, const uint64_t** d_keys, const uint32_t** d_ids) {
    const int pack_box = flags & BP_PARTITION_PACK_BOX;
    if (!L || !lut || !counts || !d_keys || !d_ids) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_partition_entries: null argument");
    const KindInfo ki = kind_info(L->kind);
    if (split_depth == 0 || split_depth > ki.axis_bits || n_cells != ((size_t)1 << (ki.dim * split_depth)) || n_cells > 4096 ||
        n_dest == 0 || n_dest > 255)
        return fail(L, BP_ERR_INVALID_ARG, "bp_layer_partition_entries: bad split_depth / n_cells / n_dest");
    // (a layer without entries has nothing to pack: a rank whose objects all fell outside the system bounds must not fail here
    // while its peers wait for it in the exchange)
    if (pack_box && L->m != 0 && !box_can_ride(L)) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_partition_entries: ids cannot carry box codes");
    DeviceGuard g(L->device);
    BP_TRY(ensure_plain(L));
    const bool already_packed = pack_box && L->ids_packed;  // extend wrote the codes into the ids: nothing to do
    if (!already_packed) BP_TRY(ensure_unpacked(L));
    const int other = L->cur ^ 1;
    for (uint32_t i = 0; i <= n_dest; ++i) counts[i] = 0;
    BP_TRY(ensure_dev(L, L->keys[other], std::max<size_t>(L->m, 1) * sizeof(uint64_t)));
    BP_TRY(ensure_dev(L, L->ids[other], std::max<size_t>(L->m, 1) * sizeof(uint32_t)));
    *d_keys = (const uint64_t*)L->keys[other].p;
    *d_ids = (const uint32_t*)L->ids[other].p;
    if (L->m == 0) return BP_OK;
    BP_TRY(ensure_dev(L, L->skip, n_cells));  // device copy of the lut
    BP_TRY(ensure_dev(L, L->radix_ws, radix_workspace_bytes(L->m, 1)));
    BP_TRY(ensure_host(L, L->h_raw, 256 * sizeof(unsigned long long)));
    const int cell_shift = (int)(ki.dim * ki.axis_bits + ki.depth_bits - ki.dim * split_depth);
    // the bucket sizes follow from the cell histogram the caller just asked for (same tree, same split depth): table, sizes and
    // the scalars' reset then ride in ONE setup launch (partition_entries_known) instead of three H2D copies, a memset, a D2D copy
    // and a kernel -- on this path the GPU has nothing queued, and each of them cost the host's launch latency
    static const bool no_fused_setup = getenv("BP_NO_PARTITION_SETUP") != nullptr;  // A/B switch
    if (!no_fused_setup && n_cells <= PART_SETUP_MAX_CELLS && L->cell_hist_valid && L->cell_hist_depth == split_depth &&
        L->cell_hist_cache.size() == n_cells + 1) {
        unsigned long long h[256];
        std::memset(h, 0, sizeof h);
        for (size_t c = 0; c < n_cells; ++c) h[lut[c] < n_dest ? lut[c] : 0] += L->cell_hist_cache[c];
        h[n_dest] += L->cell_hist_cache[n_cells];  // coarse entries
        for (uint32_t i = 0; i <= n_dest; ++i) counts[i] = h[i];  // the bucket sizes ARE these sums
        const unsigned long long* d_hist_k = nullptr;
        {
            ProfScope ps(L, BP_K_MISC);
            HIP_TRY(L, partition_entries_known(cur_keys(L), (uint64_t*)L->keys[other].p, cur_ids(L), (uint32_t*)L->ids[other].p, L->m, lut,
                                               (uint32_t)n_cells, (unsigned char*)L->skip.p, h, cell_shift, (1u << ki.depth_bits) - 1u, split_depth,
                                               n_dest, L->radix_ws.p, L->radix_ws.cap, (DevScalars*)L->d_sc.p, neutral_scalars(), &d_hist_k,
                                               L->stream, (pack_box && !already_packed) ? (const uint8_t*)L->box[L->cur].p : nullptr));
        }
        if (flags & BP_PARTITION_ENQUEUE_ONLY) {
            // nobody waits here, but the pass's error word must not be lost (the next reset of the scalars would overwrite it):
            // it is copied behind the pass and checked by the layer's next host round trip
            BP_TRY(ensure_dev(L, L->d_part_err, sizeof(unsigned)));
            BP_TRY(ensure_host(L, L->h_part_err, sizeof(unsigned)));
            HIP_TRY(L, hipMemcpyAsync(L->d_part_err.p, &((DevScalars*)L->d_sc.p)->err, sizeof(unsigned), hipMemcpyDeviceToDevice, L->stream));
            HIP_TRY(L, hipMemcpyAsync(L->h_part_err.p, L->d_part_err.p, sizeof(unsigned), hipMemcpyDeviceToHost, L->stream));
            L->part_pend = true;
            return BP_OK;
        }
        DevScalars sck;
        BP_TRY(sc_fetch(L, &sck));
        return BP_OK;
    }
    HIP_TRY(L, hipMemcpyAsync(L->skip.p, lut, n_cells, hipMemcpyHostToDevice, L->stream));
    BP_TRY(sc_reset(L));
    // bucket sizes straight from the cell histogram the caller just asked for (same tree, same split depth):
    // the partition pass then needs no histogram read of its own
    const unsigned long long* ready_hist = nullptr;
    if (L->cell_hist_valid && L->cell_hist_depth == split_depth && L->cell_hist_cache.size() == n_cells + 1) {
        // upload source: the second half of h_raw (the first half receives the bucket sizes back below: no wait in between)
        BP_TRY(ensure_host(L, L->h_raw, 512 * sizeof(unsigned long long)));
        unsigned long long* h = (unsigned long long*)L->h_raw.p + 256;
        std::memset(h, 0, 256 * sizeof(unsigned long long));
        for (size_t c = 0; c < n_cells; ++c) h[lut[c] < n_dest ? lut[c] : 0] += L->cell_hist_cache[c];
        h[n_dest] += L->cell_hist_cache[n_cells];  // coarse entries
        BP_TRY(ensure_dev(L, L->pair_hist, SX_HIST_MAX * RS_BINS * sizeof(unsigned long long)));
        HIP_TRY(L, hipMemcpyAsync(L->pair_hist.p, h, 256 * sizeof(unsigned long long), hipMemcpyHostToDevice, L->stream));
        ready_hist = (const unsigned long long*)L->pair_hist.p;
        for (uint32_t i = 0; i <= n_dest; ++i) counts[i] = h[i];  // the bucket sizes ARE these sums
    }
    const unsigned long long* d_hist = nullptr;
    {
        ProfScope ps(L, BP_K_MISC);
        HIP_TRY(L, partition_entries(cur_keys(L), (uint64_t*)L->keys[other].p, cur_ids(L), (uint32_t*)L->ids[other].p, L->m,
                                     (const unsigned char*)L->skip.p, cell_shift, (1u << ki.depth_bits) - 1u, split_depth, n_dest,
                                     L->radix_ws.p, L->radix_ws.cap, (DevScalars*)L->d_sc.p, &d_hist, L->stream,
                                     (pack_box && !already_packed) ? (const uint8_t*)L->box[L->cur].p : nullptr, ready_hist));
    }
    // enqueue only: the bucket sizes are known from the histogram, the caller orders whatever reads the buckets behind this
    // layer's stream (bp_layer_stream) itself, and nobody waits here
    if ((flags & BP_PARTITION_ENQUEUE_ONLY) && ready_hist) {
        // nobody waits here, but the pass's error word must not be lost (the next reset of the scalars would overwrite it):
        // it is copied behind the pass and checked by the layer's next host round trip
        BP_TRY(ensure_dev(L, L->d_part_err, sizeof(unsigned)));
        BP_TRY(ensure_host(L, L->h_part_err, sizeof(unsigned)));
        HIP_TRY(L, hipMemcpyAsync(L->d_part_err.p, &((DevScalars*)L->d_sc.p)->err, sizeof(unsigned), hipMemcpyDeviceToDevice, L->stream));
        HIP_TRY(L, hipMemcpyAsync(L->h_part_err.p, L->d_part_err.p, sizeof(unsigned), hipMemcpyDeviceToHost, L->stream));
        L->part_pend = true;
        return BP_OK;
    }
    HIP_TRY(L, hipMemcpyAsync(L->h_raw.p, d_hist, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, L->stream));
    DevScalars sc;
    BP_TRY(sc_fetch(L, &sc));
    for (uint32_t i = 0; i <= n_dest; ++i) counts[i] = ((const unsigned long long*)L->h_raw.p)[i];
    return BP_OK;
}

int bp_layer_set_coarse_emit_range(bp_layer_t* L, uint32_t id_lo, uint32_t id_hi) {
    if (!L) return BP_ERR_INVALID_ARG;
    L->coarse_lo = id_lo; L->coarse_hi = id_hi;
    return BP_OK;
}

int bp_layer_scan_raw_shard_device(bp_layer_t* L, uint32_t coarse_depth, int emit_coarse, const uint64_t** d_packed, size_t* n_raw) {
    if (!L || !d_packed || !n_raw) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_raw_shard: null argument");
    DeviceGuard g(L->device);
    BP_TRY(scan_raw_device(L, coarse_depth, emit_coarse));
    BP_TRY(sync_stream(L));
    *d_packed = (const uint64_t*)L->pairs[0].p;
    *n_raw = L->n_raw;
    return BP_OK;
}

int bp_layer_scan_shard_pairs_device(bp_layer_t* L, uint32_t coarse_depth, int emit_coarse, const uint64_t** d_packed, size_t* n_raw,
                                     int* unique) {
    if (!L || !d_packed || !n_raw || !unique) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_shard_pairs: null argument");
    DeviceGuard g(L->device);
    if (L->ids_distinct) {
        BP_TRY(scan_fused_device(L, coarse_depth, emit_coarse, true));
    } else {
        BP_TRY(scan_raw_device(L, coarse_depth, emit_coarse));
        BP_TRY(sync_stream(L));
        L->raw_is_unique = false;
    }
    *d_packed = (const uint64_t*)L->pairs[0].p;
    *n_raw = L->n_raw;
    *unique = L->raw_is_unique ? 1 : 0;
    return BP_OK;
}

int bp_layer_par_scan_shard_device(bp_layer_t* L, uint32_t coarse_depth, int emit_coarse, const uint32_t** d_pairs, size_t* n_pairs,
                                   int* unique) {
    if (!L || !d_pairs || !n_pairs || !unique) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_par_scan_shard: null argument");
    DeviceGuard g(L->device);
    BP_TRY(scan_device(L, coarse_depth, emit_coarse));
    *d_pairs = (const uint32_t*)L->out_pairs.p;
    *n_pairs = L->n_pairs;
    *unique = L->raw_is_unique ? 1 : 0;
    return BP_OK;
}

int bp_layer_sort_packed_device(bp_layer_t* L, uint64_t* d_packed, size_t n, uint64_t varying_mask, const uint32_t** d_pairs,
                                size_t* n_pairs) {
    if (!L || !d_pairs || !n_pairs || (n && !d_packed)) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_sort_packed: null argument");
    DeviceGuard g(L->device);
    L->n_raw = n;
    BP_TRY(sort_distinct_pairs_in(L, d_packed, n, varying_mask ? varying_mask : ~0ull));
    if (n == 0) BP_TRY(sync_stream(L));
    *d_pairs = (const uint32_t*)L->out_pairs.p;
    *n_pairs = L->n_pairs;
    return BP_OK;
}

int bp_layer_sort_packed_range_device(bp_layer_t* L, uint64_t* d_packed, size_t n, uint64_t varying_mask, uint32_t first_lo,
                                      uint32_t first_hi, const uint32_t** d_pairs, size_t* n_pairs) {
    if (!L || !d_pairs || !n_pairs || (n && !d_packed)) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_sort_packed_range: null argument");
    if (first_hi < first_lo) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_sort_packed_range: empty first-id range");
    DeviceGuard g(L->device);
    L->n_raw = n;
    const uint32_t range[2] = {first_lo, first_hi};
    BP_TRY(sort_distinct_pairs_in(L, d_packed, n, varying_mask ? varying_mask : ~0ull, false, range));
    if (n == 0) BP_TRY(sync_stream(L));
    *d_pairs = (const uint32_t*)L->out_pairs.p;
    *n_pairs = L->n_pairs;
    return BP_OK;
}

namespace {
// n records at d_packed (block-structured when holes: extent slots, L->blockfill) -> grouped by owner rank
int partition_pairs_impl(bp_layer* L, const uint64_t* d_packed, size_t n, bool holes, const uint32_t* bounds, uint32_t n_bounds,
                         uint64_t* counts, const uint64_t** d_out, bool owner_hist_ready = false) {
    for (uint32_t i = 0; i <= n_bounds; ++i) counts[i] = 0;
    // write into whichever pair buffer is not the input
    DevBuf& dst = (d_packed == (const uint64_t*)L->pairs[1].p) ? L->pairs[0] : L->pairs[1];
    BP_TRY(ensure_dev(L, dst, std::max<size_t>(n, 1) * sizeof(uint64_t)));
    *d_out = (const uint64_t*)dst.p;
    if (n == 0) return BP_OK;
    const size_t extent = holes ? L->raw_extent : n;
    BP_TRY(ensure_dev(L, L->radix_ws, radix_workspace_bytes(extent, 1)));
    BP_TRY(ensure_host(L, L->h_raw, 256 * sizeof(unsigned long long)));
    BP_TRY(sc_reset(L));
    const unsigned long long* d_hist = nullptr;
    {
        ProfScope ps(L, BP_K_MISC);
        HIP_TRY(L, partition_pairs(d_packed, (uint64_t*)dst.p, extent, bounds, (int)n_bounds, L->radix_ws.p, L->radix_ws.cap,
                                   (DevScalars*)L->d_sc.p, &d_hist, L->stream, holes ? (const uint32_t*)L->blockfill.p : nullptr, n,
                                   owner_hist_ready ? (const unsigned long long*)L->pair_hist.p : nullptr));
    }
    HIP_TRY(L, hipMemcpyAsync(L->h_raw.p, d_hist, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost, L->stream));
    DevScalars sc;
    BP_TRY(sc_fetch(L, &sc));
    for (uint32_t i = 0; i <= n_bounds; ++i) counts[i] = ((const unsigned long long*)L->h_raw.p)[i];
    return BP_OK;
}
}  // namespace

int bp_layer_partition_pairs_device(bp_layer_t* L, const uint64_t* d_packed, size_t n, const uint32_t* bounds, uint32_t n_bounds,
                                    uint64_t* counts, const uint64_t** d_out) {
    if (!L || !counts || !d_out || (n && !d_packed) || (n_bounds && !bounds) || n_bounds > 15)
        return fail(L, BP_ERR_INVALID_ARG, "bp_layer_partition_pairs: bad argument");
    DeviceGuard g(L->device);
    return partition_pairs_impl(L, d_packed, n, false, bounds, n_bounds, counts, d_out);
}

int bp_layer_scan_shard_partition_device(bp_layer_t* L, uint32_t coarse_depth, int emit_coarse, const uint32_t* bounds,
                                         uint32_t n_bounds, uint64_t* counts, const uint64_t** d_out, size_t* n_raw, int* unique) {
    if (!L || !counts || !d_out || !n_raw || !unique || (n_bounds && !bounds) || n_bounds > 15)
        return fail(L, BP_ERR_INVALID_ARG, "bp_layer_scan_shard_partition: bad argument");
    DeviceGuard g(L->device);
    if (L->ids_distinct) {
        BP_TRY(scan_fused_device(L, coarse_depth, emit_coarse, true, nullptr, 0, /*allow_holes=*/true, nullptr, bounds, (int)n_bounds));
    } else {
        BP_TRY(scan_raw_device(L, coarse_depth, emit_coarse));
        BP_TRY(sync_stream(L));
        L->raw_is_unique = false;
    }
    *n_raw = L->n_raw;
    *unique = L->raw_is_unique ? 1 : 0;
    const bool holes = L->raw_holes;
    L->raw_holes = false;  // consumed here: the partition pass reads around the holes
    const bool ready = L->owner_hist_valid && holes;
    L->owner_hist_valid = false;
    return partition_pairs_impl(L, (const uint64_t*)L->pairs[0].p, L->n_raw, holes, bounds, n_bounds, counts, d_out, ready);
}

int bp_layer_sort_unique_packed_device(bp_layer_t* L, uint64_t* d_packed, size_t n, uint64_t varying_mask,
                                       const uint32_t** d_pairs, size_t* n_pairs) {
    if (!L || !d_pairs || !n_pairs || (n && !d_packed)) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_sort_unique_packed: null argument");
    DeviceGuard g(L->device);
    L->n_raw = n;
    BP_TRY(sort_unique_pairs_in(L, d_packed, n, varying_mask ? varying_mask : ~0ull));
    if (n == 0) BP_TRY(sync_stream(L));
    *d_pairs = (const uint32_t*)L->out_pairs.p;
    *n_pairs = L->n_pairs;
    return BP_OK;
}

int bp_layer_tree_device(bp_layer_t* L, const uint64_t** d_keys, const uint32_t** d_ids, size_t* m, int* sorted) {
    if (!L) return BP_ERR_INVALID_ARG;
    { DeviceGuard g(L->device); BP_TRY(ensure_unpacked(L)); }
    if (d_keys) *d_keys = cur_keys(L);
    if (d_ids) *d_ids = cur_ids(L);
    if (m) *m = L->m;
    if (sorted) *sorted = L->sorted ? 1 : 0;
    return BP_OK;
}

int bp_layer_tree(bp_layer_t* L, const uint64_t** keys, const uint32_t** ids, size_t* m, int* sorted) {
    if (!L) return BP_ERR_INVALID_ARG;
    DeviceGuard g(L->device);
    BP_TRY(ensure_unpacked(L));
    BP_TRY(ensure_host(L, L->h_keys, std::max<size_t>(L->m * sizeof(uint64_t), 8)));
    BP_TRY(ensure_host(L, L->h_ids, std::max<size_t>(L->m * sizeof(uint32_t), 8)));
    if (L->m) {
        HIP_TRY(L, hipMemcpyAsync(L->h_keys.p, cur_keys(L), L->m * sizeof(uint64_t), hipMemcpyDeviceToHost, L->stream));
        HIP_TRY(L, hipMemcpyAsync(L->h_ids.p, cur_ids(L), L->m * sizeof(uint32_t), hipMemcpyDeviceToHost, L->stream));
        BP_TRY(sync_stream(L));
    }
    if (keys) *keys = (const uint64_t*)L->h_keys.p;
    if (ids) *ids = (const uint32_t*)L->h_ids.p;
    if (m) *m = L->m;
    if (sorted) *sorted = L->sorted ? 1 : 0;
    return BP_OK;
}

int bp_layer_set_tree(bp_layer_t* L, uint32_t min_depth, const uint64_t* keys, const uint32_t* ids, size_t m, int sorted) {
    if (!L || (m && (!keys || !ids))) return fail(L, BP_ERR_INVALID_ARG, "bp_layer_set_tree: null argument");
    DeviceGuard g(L->device);
    BP_TRY(bp_layer_clear(L));
    L->min_depth = min_depth;
    if (m) BP_TRY(bp_layer_append_entries(L, keys, ids, m));
    L->sorted = sorted != 0;
    ++L->epoch;
    if (L->sorted) L->order_ok = true;
    return BP_OK;
}

int bp_layer_clone(const bp_layer_t* src, bp_layer_t** out) {
    if (!src || !out) return BP_ERR_INVALID_ARG;
    { DeviceGuard g0(src->device); BP_TRY(ensure_unpacked(const_cast<bp_layer*>(src))); }
    BP_TRY(bp_layer_create(out, src->kind, src->min_depth, src->m, 0, src->device));
    bp_layer* L = *out;
    DeviceGuard g(L->device);
    if (src->m) {
        (void)hipStreamSynchronize(src->stream);
        HIP_TRY(L, launch_copy_bytes(L->keys[0].p, src->keys[src->cur].p, src->m * sizeof(uint64_t), L->stream));
        HIP_TRY(L, launch_copy_bytes(L->ids[0].p, src->ids[src->cur].p, src->m * sizeof(uint32_t), L->stream));
        BP_TRY(ensure_dev(L, L->box[0], src->m));
        if (src->box_valid && src->box[src->cur].p)
            HIP_TRY(L, launch_copy_bytes(L->box[0].p, src->box[src->cur].p, src->m, L->stream));
        BP_TRY(sync_stream(L));
    }
    L->box_valid = src->box_valid && (src->m == 0 || src->box[src->cur].p != nullptr);
    L->box_all_known = src->box_all_known;
    L->m = src->m; L->sorted = src->sorted;
    L->have_ids = src->have_ids; L->order_ok = src->order_ok; L->ids_distinct = src->ids_distinct; L->max_id = src->max_id; L->min_id = src->min_id;
    L->key_or = src->key_or; L->key_and = src->key_and; L->id_or = src->id_or; L->id_and = src->id_and;
    L->depth_mask = src->depth_mask;
    L->stats.entries = L->m;
    return BP_OK;
}

int bp_layer_equals(const bp_layer_t* a, const bp_layer_t* b, int* equal) {
    if (!a || !b || !equal) return BP_ERR_INVALID_ARG;
    *equal = 0;
    if (a->kind != b->kind || a->min_depth != b->min_depth || a->m != b->m || a->sorted != b->sorted) return BP_OK;
    if (a->m == 0) { *equal = 1; return BP_OK; }
    const uint64_t *ka, *kb;
    const uint32_t *ia, *ib;
    size_t m;
    BP_TRY(bp_layer_tree((bp_layer_t*)a, &ka, &ia, &m, nullptr));
    BP_TRY(bp_layer_tree((bp_layer_t*)b, &kb, &ib, &m, nullptr));
    *equal = (std::memcmp(ka, kb, m * sizeof(uint64_t)) == 0 && std::memcmp(ia, ib, m * sizeof(uint32_t)) == 0) ? 1 : 0;
    return BP_OK;
}

int bp_layer_len(const bp_layer_t* L, size_t* m) { if (!L || !m) return BP_ERR_INVALID_ARG; *m = L->m; return BP_OK; }
int bp_layer_is_sorted(const bp_layer_t* L, int* s) { if (!L || !s) return BP_ERR_INVALID_ARG; *s = L->sorted ? 1 : 0; return BP_OK; }
int bp_layer_min_depth(const bp_layer_t* L, uint32_t* d) { if (!L || !d) return BP_ERR_INVALID_ARG; *d = L->min_depth; return BP_OK; }
int bp_layer_index_kind(const bp_layer_t* L, int* k) { if (!L || !k) return BP_ERR_INVALID_ARG; *k = L->kind; return BP_OK; }
int bp_layer_device(const bp_layer_t* L, int* d) { if (!L || !d) return BP_ERR_INVALID_ARG; *d = L->device; return BP_OK; }

int bp_layer_stats(const bp_layer_t* L, bp_stats* out) {
    if (!L || !out) return BP_ERR_INVALID_ARG;
    *out = L->stats;
    out->entries = L->m;
    out->depth_mask = L->depth_mask;
    out->flags = (L->order_ok ? 1u : 0u) | (L->ids_distinct ? 2u : 0u);
    return BP_OK;
}

int bp_layer_profile_enable(bp_layer_t* L, int on) { if (!L) return BP_ERR_INVALID_ARG; L->prof_on = on != 0; return BP_OK; }
int bp_layer_profile_select(bp_layer_t* L, uint32_t class_mask) {
    if (!L) return BP_ERR_INVALID_ARG;
    L->prof_mask = class_mask;
    return BP_OK;
}
int bp_layer_profile_sample(bp_layer_t* L, uint32_t every) {
    if (!L || every == 0) return BP_ERR_INVALID_ARG;
    L->prof_every = every;
    L->prof_frame = 0;
    return BP_OK;
}
int bp_layer_profile_reset(bp_layer_t* L) {
    if (!L) return BP_ERR_INVALID_ARG;
    for (int i = 0; i < BP_K_COUNT; ++i) { L->prof_ms[i] = 0; L->prof_n[i] = 0; }
    return BP_OK;
}
int bp_layer_profile_get(bp_layer_t* L, int cls, double* total_ms, uint64_t* launches) {
    if (!L || cls < 0 || cls >= BP_K_COUNT) return BP_ERR_INVALID_ARG;
    if (total_ms) *total_ms = L->prof_ms[cls];
    if (launches) *launches = L->prof_n[cls];
    return BP_OK;
}

}  // extern "C"
