// fake_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl at its C ABI, for ranks that are PROCESSES sharing one GPU.
//
// The C sharded engine (csrc/bp_sharded.hip) binds RCCL with dlopen; BP_RCCL_LIBRARY points that dlopen at this library instead.
// Every byte then travels through a POSIX shared-memory segment (device -> host staging -> peer's host -> device), which is
// slow and beside the point: what the test exercises is the ENGINE's own multi-rank code over a real process boundary -- the
// order in which both ends of a peer pair post their grouped sends / receives, the cut of messages into rounds, the status
// words of the first all-gather, the bounded waits and the abort path -- which a one-GPU pool can otherwise only run with
// ranks as threads (LOCAL groups) or with a single RCCL rank.  Only the entry points bp_sharded.hip binds are provided, with
// the semantics it relies on:
//   * ncclSend / ncclRecv between ncclGroupStart / ncclGroupEnd are matched per peer pair in posting order;
//   * operations are ordered behind what the caller enqueued on `stream` before them, and complete before anything enqueued
//     after them starts (here: the stream is synchronised and the exchange is done by the time the call returns);
//   * ncclCommAbort ends the peers' waits; ncclCommGetAsyncError reports it.
// Built by tests/test_gpu_sharded_procs.py with hipcc (host code + the HIP runtime); never shipped, never linked by the product.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int MAX_RANKS = 16;
enum { OK = 0, ERR_SYSTEM = 2, ERR_INTERNAL = 3, ERR_INVALID = 4, ERR_REMOTE = 6 };

struct Header {
    std::atomic<uint32_t> ready;        // ranks that mapped the segment
    std::atomic<uint32_t> arrived;      // sense-reversing barrier
    std::atomic<uint32_t> sense;
    std::atomic<uint32_t> aborted;
    uint32_t world;
    uint64_t box_bytes;                 // capacity of one (source, destination) mailbox
    uint64_t used[MAX_RANKS][MAX_RANKS];  // bytes source s has put into its mailbox for destination d in the current round
};

struct Comm {
    Header* h = nullptr;
    char* boxes = nullptr;              // world x world mailboxes of box_bytes
    size_t map_bytes = 0;
    int rank = 0, world = 1;
    uint32_t my_sense = 0;
    std::string name;
    char* box(int s, int d) const { return boxes + ((size_t)s * world + d) * h->box_bytes; }
};

struct Op { bool send; void* ptr; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
Comm* g_comm = nullptr;   // the process's communicator (one per process in these tests): an EMPTY group still takes part in the rounds' barriers

long timeout_ms() {
    static const long t = getenv("FAKE_RCCL_TIMEOUT_MS") ? atol(getenv("FAKE_RCCL_TIMEOUT_MS")) : 30000;
    return t;
}

// process-shared barrier with a way out: false when the communicator was aborted or a peer never arrives
bool barrier(Comm* c) {
    Header* h = c->h;
    if (h->aborted.load()) return false;
    const uint32_t want = c->my_sense ^ 1u;
    if (h->arrived.fetch_add(1) + 1 == (uint32_t)c->world) {
        h->arrived.store(0);
        h->sense.store(want);
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (h->sense.load() != want) {
            if (h->aborted.load()) return false;
            if ((++spins & 0x3FFu) == 0) {
                std::this_thread::yield();
                if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) {
                    h->aborted.store(1);
                    return false;
                }
            }
        }
    }
    c->my_sense = want;
    return true;
}

int run_ops(std::vector<Op>& ops) {
    Comm* c = ops.empty() ? g_comm : ops[0].comm;
    if (!c) return OK;
    Header* h = c->h;
    // everything the caller enqueued before the collective has to be finished: the sources are read by the host below
    for (const Op& o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) { h->aborted.store(1); return ERR_SYSTEM; }
    for (int d = 0; d < c->world; ++d) h->used[c->rank][d] = 0;
    int rc = OK;
    for (const Op& o : ops) {                               // sends: device -> this rank's mailbox for the peer, in posting order
        if (!o.send) continue;
        uint64_t& u = h->used[c->rank][o.peer];
        if (u + o.bytes > h->box_bytes) { rc = ERR_INTERNAL; fprintf(stderr, "fake_rccl: mailbox of %llu bytes too small (FAKE_RCCL_BOX_BYTES)\n", (unsigned long long)h->box_bytes); break; }
        if (o.bytes && hipMemcpy(c->box(c->rank, o.peer) + u, o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) { rc = ERR_SYSTEM; break; }
        u += o.bytes;
    }
    if (rc != OK) h->aborted.store(1);
    if (!barrier(c)) return rc != OK ? rc : ERR_REMOTE;     // every mailbox is filled
    uint64_t taken[MAX_RANKS] = {0};
    for (const Op& o : ops) {                               // receives: the peer's mailbox for this rank, in posting order
        if (o.send) continue;
        uint64_t& t = taken[o.peer];
        if (t + o.bytes > h->used[o.peer][c->rank]) { rc = ERR_INVALID; fprintf(stderr, "fake_rccl: rank %d expects more from rank %d than it sent\n", c->rank, o.peer); break; }
        if (o.bytes && hipMemcpy(o.ptr, c->box(o.peer, c->rank) + t, o.bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = ERR_SYSTEM; break; }
        t += o.bytes;
    }
    if (rc == OK)
        for (int s = 0; s < c->world; ++s)
            if (taken[s] != h->used[s][c->rank]) { rc = ERR_INVALID; fprintf(stderr, "fake_rccl: rank %d left %llu bytes of rank %d unreceived\n", c->rank, (unsigned long long)(h->used[s][c->rank] - taken[s]), s); break; }
    if (rc != OK) h->aborted.store(1);
    if (!barrier(c)) return rc != OK ? rc : ERR_REMOTE;     // the mailboxes may be refilled
    return rc;
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%lld", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return OK;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ERR_INVALID;
    Comm* c = new Comm();
    c->rank = rank; c->world = nranks; c->name = id.internal;
    const uint64_t box_bytes = getenv("FAKE_RCCL_BOX_BYTES") ? strtoull(getenv("FAKE_RCCL_BOX_BYTES"), nullptr, 0) : (32ull << 20);
    c->map_bytes = sizeof(Header) + (size_t)nranks * nranks * box_bytes;
    int fd = -1;
    const auto t0 = std::chrono::steady_clock::now();
    if (rank == 0) {
        fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd >= 0 && ftruncate(fd, (off_t)c->map_bytes) != 0) { close(fd); fd = -1; }
    } else {
        while ((fd = shm_open(c->name.c_str(), O_RDWR, 0600)) < 0) {   // rank 0 creates the segment: wait for it
            struct stat st;
            (void)st;
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) break;
            usleep(2000);
        }
        if (fd >= 0) {                                                   // ... and for its size
            struct stat st;
            while (fstat(fd, &st) == 0 && (size_t)st.st_size < c->map_bytes) {
                if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) { close(fd); fd = -1; break; }
                usleep(2000);
            }
        }
    }
    if (fd < 0) { delete c; return ERR_SYSTEM; }
    void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ERR_SYSTEM; }
    c->h = (Header*)p;
    c->boxes = (char*)p + sizeof(Header);
    if (rank == 0) { c->h->world = (uint32_t)nranks; c->h->box_bytes = box_bytes; }   // (a fresh segment is zero-filled)
    c->h->ready.fetch_add(1);
    while (c->h->ready.load() < (uint32_t)nranks) {                                       // all ranks mapped it: safe to unlink below
        if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > timeout_ms()) { munmap(p, c->map_bytes); delete c; return ERR_REMOTE; }
        usleep(1000);
    }
    while (c->h->box_bytes == 0) usleep(100);
    if (rank == 0) shm_unlink(c->name.c_str());   // the mappings keep it alive; nothing is left behind in /dev/shm
    g_comm = c;
    *comm = c;
    return OK;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return OK;
    if (g_comm == c) g_comm = nullptr;
    if (c->h) munmap((void*)c->h, c->map_bytes);
    delete c;
    return OK;
}

int ncclCommAbort(void* comm) {
    Comm* c = (Comm*)comm;
    if (c && c->h) c->h->aborted.store(1);
    return ncclCommDestroy(comm);
}

int ncclCommGetAsyncError(void* comm, int* async) {
    Comm* c = (Comm*)comm;
    if (async) *async = (c && c->h && c->h->aborted.load()) ? ERR_REMOTE : OK;
    return OK;
}

const char* ncclGetErrorString(int rc) {
    switch (rc) {
        case OK: return "no error";
        case ERR_SYSTEM: return "fake_rccl: system / HIP error";
        case ERR_INTERNAL: return "fake_rccl: mailbox overflow";
        case ERR_INVALID: return "fake_rccl: the two ends of a peer pair posted different transfers";
        case ERR_REMOTE: return "fake_rccl: a peer aborted or never arrived";
        default: return "fake_rccl: error";
    }
}

int ncclGroupStart() { ++g_depth; return OK; }

int ncclGroupEnd() {
    if (g_depth <= 0) return ERR_INVALID;
    if (--g_depth > 0) return OK;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_ops(ops);
}

static int post(bool send, void* ptr, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c || peer < 0 || peer >= c->world) return ERR_INVALID;
    const size_t elem = dtype == 4 || dtype == 5 ? 8 : (dtype == 2 || dtype == 3 ? 4 : 1);   // ncclInt64 / ncclUint64, ncclInt32 / ncclUint32, else bytes
    g_ops.push_back(Op{send, ptr, count * elem, peer, c, stream});
    if (g_depth == 0) {
        std::vector<Op> ops;
        ops.swap(g_ops);
        return run_ops(ops);
    }
    return OK;
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) { return post(true, const_cast<void*>(buf), count, dtype, peer, comm, stream); }
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) { return post(false, buf, count, dtype, peer, comm, stream); }

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c) return ERR_INVALID;
    const size_t elem = dtype == 4 || dtype == 5 ? 8 : (dtype == 2 || dtype == 3 ? 4 : 1);
    std::vector<Op> ops;
    for (int p = 0; p < c->world; ++p) ops.push_back(Op{true, const_cast<void*>(send), count * elem, p, c, stream});
    for (int p = 0; p < c->world; ++p) ops.push_back(Op{false, (char*)recv + (size_t)p * count * elem, count * elem, p, c, stream});
    return run_ops(ops);
}

}  // extern "C"
