// Microbenchmark of the radix passes (bp::radix_sort of csrc/bp_kernels.hip) on synthetic records of the hot path's shape:
//   pairs : n packed pairs (first id << 32 | second id, ids uniform below `ids`), key-only, passes on first-id bits 8..23
//   tree  : m (key, id) entries, 16 key bits, u32 payload
// Every pass is bracketed by HIP events (the library's own `mark` callback); the result is verified (sorted by the sorted
// bits, checksum of the records unchanged) unless BP_BENCH_NOVERIFY is set (timing-only experiment builds).
// build (repo root):  hipcc -O2 --offload-arch=gfx950 -Ibroadphase_rs_amd/csrc -Iinclude tools/micro/radix_bench.hip \
//                       -o tools/micro/radix_bench -Lbroadphase_rs_amd/lib -lbroadphase_amd
// run:                LD_LIBRARY_PATH=broadphase_rs_amd/lib[_ab/<variant>] tools/micro/radix_bench [n_pairs] [m_entries] [reps]
#include "bp_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

__global__ void fill_pairs(u64* k, size_t n, u32 ids, u64 seed, u32 skew) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u64 x = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 29;
        u32 a = (u32)(x % ids);
        const u32 b = (u32)((x >> 32) % ids);
        if (skew && (x >> 60) != 0) a = (a & 0xFFu) | (skew << 8);  // 15 of 16 records share the sorted bits: one bin takes almost everything
        k[i] = ((u64)a << 32) | b;
    }
}
__global__ void fill_ids(u32* v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) v[i] = (u32)i;
}
// out[0] += sum k, out[1] ^= fold, out[2] += order violations of (k >> lo) & mask between neighbours
__global__ void check(const u64* k, size_t n, int lo, u64 mask, u64* out) {
    u64 s = 0, x = 0, bad = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u64 v = k[i];
        s += v; x ^= v * 0x9E3779B97F4A7C15ull;
        if (lo >= 0 && i + 1 < n && ((v >> lo) & mask) > ((k[i + 1] >> lo) & mask)) ++bad;
    }
    atomicAdd(&out[0], s); atomicXor(&out[1], x); atomicAdd(&out[2], bad);
}
// payload check: every (key, val) of the output must be a pair of the input: sum of key * (val + 1)
__global__ void check_kv(const u64* k, const u32* v, size_t n, u64* out) {
    u64 s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += k[i] * (u64)(v[i] + 1u);
    atomicAdd(&out[3], s);
}

struct Marks {
    hipStream_t s;
    std::vector<hipEvent_t> ev;
    std::vector<int> cls;
};
static void mark(void* ctx, int c) {
    Marks* m = (Marks*)ctx;
    hipEvent_t e; CK(hipEventCreate(&e)); CK(hipEventRecord(e, m->s));
    m->ev.push_back(e); m->cls.push_back(c);
}

static void run(const char* name, size_t n, bool with_vals, u32 ids, int shift0, int reps, hipStream_t s) {
    u64 *k0, *k1, *chk; u32 *v0 = nullptr, *v1 = nullptr;
    // BP_BENCH_SHIFT=k: the output buffers start k records into their allocation (line alignment of the claiming pass)
    const size_t shift = getenv("BP_BENCH_SHIFT") ? (size_t)atoi(getenv("BP_BENCH_SHIFT")) : 0;
    const u32 skew = getenv("BP_BENCH_SKEW") ? (u32)atoi(getenv("BP_BENCH_SKEW")) : 0u;
    CK(hipMalloc(&k0, (n + 64) * 8)); CK(hipMalloc(&k1, (n + 64) * 8)); CK(hipMalloc(&chk, 64));
    if (with_vals) { CK(hipMalloc(&v0, (n + 64) * 4)); CK(hipMalloc(&v1, (n + 64) * 4)); }
    u64* const k0_alloc = k0; u64* const k1_alloc = k1; u32* const v0_alloc = v0; u32* const v1_alloc = v1;
    k1 += shift; if (with_vals) v1 += shift;
    bp::RadixPlan plan; memset(&plan, 0, sizeof plan);
    plan.n_passes = 2; plan.shift[0] = shift0; plan.shift[1] = shift0 + 8;
    const size_t wsb = bp::radix_workspace_bytes(n, 2);
    void* ws; CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
    bp::DevScalars* sc; CK(hipMalloc(&sc, sizeof(bp::DevScalars))); CK(hipMemset(sc, 0, sizeof(bp::DevScalars)));
    u32 epoch = 0;
    const bool verify = getenv("BP_BENCH_NOVERIFY") == nullptr;
    double tot[3] = {0, 0, 0};
    for (int r = 0; r < reps + 1; ++r) {
        fill_pairs<<<2048, 256, 0, s>>>(k0, n, ids, 12345 + r, skew);
        if (with_vals) fill_ids<<<2048, 256, 0, s>>>(v0, n);
        u64 before[4] = {0, 0, 0, 0}, after[4] = {0, 0, 0, 0};
        if (verify) {
            CK(hipMemsetAsync(chk, 0, 64, s));
            check<<<1024, 256, 0, s>>>(k0, n, -1, 0, chk);
            if (with_vals) check_kv<<<1024, 256, 0, s>>>(k0, v0, n, chk);
            CK(hipMemcpyAsync(before, chk, 32, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        }
        Marks m; m.s = s;
        int res = 0;
        CK(bp::radix_sort((uint64_t*)k0, (uint64_t*)k1, v0, v1, n, plan, ws, wsb, sc, &res, s, mark, &m, 100, 200, nullptr, nullptr, nullptr, 0,
                          nullptr, 1, /*unstable_first=*/1, &epoch, nullptr));
        CK(hipStreamSynchronize(s));
        // marks come in (begin, end) pairs: hist/prepare, pass 0, pass 1
        double t[3] = {0, 0, 0};
        for (size_t i = 0, q = 0; i + 1 < m.ev.size() && q < 3; i += 2, ++q) { float ms = 0; CK(hipEventElapsedTime(&ms, m.ev[i], m.ev[i + 1])); t[q] = ms; }
        for (hipEvent_t e : m.ev) CK(hipEventDestroy(e));
        if (r > 0) for (int q = 0; q < 3; ++q) tot[q] += t[q];
        if (verify) {
            CK(hipMemsetAsync(chk, 0, 64, s));
            const u64* out = res ? k1 : k0;
            check<<<1024, 256, 0, s>>>(out, n, shift0, 0xFFFFull, chk);
            if (with_vals) check_kv<<<1024, 256, 0, s>>>(out, res ? v1 : v0, n, chk);
            CK(hipMemcpyAsync(after, chk, 32, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
            bp::DevScalars hs; CK(hipMemcpy(&hs, sc, sizeof hs, hipMemcpyDeviceToHost));
            if (before[0] != after[0] || before[1] != after[1] || after[2] != 0 || before[3] != after[3] || hs.err) {
                printf("%s: VERIFY FAILED rep %d: sum %d xor %d unsorted %llu kv %d err %u\n", name, r, before[0] == after[0], before[1] == after[1],
                       after[2], before[3] == after[3], hs.err);
                exit(1);
            }
        }
    }
    const double bytes = (double)n * (with_vals ? 24.0 : 16.0);
    printf("%-6s n %zu  prepare %.1f us  pass0 %.1f us (%.2f TB/s)  pass1 %.1f us (%.2f TB/s)%s\n", name, n, tot[0] / reps * 1e3, tot[1] / reps * 1e3,
           bytes / (tot[1] / reps * 1e-3) / 1e12, tot[2] / reps * 1e3, bytes / (tot[2] / reps * 1e-3) / 1e12, verify ? "  verified" : "  NOT verified");
    CK(hipFree(k0_alloc)); CK(hipFree(k1_alloc)); CK(hipFree(chk)); CK(hipFree(ws)); CK(hipFree(sc));
    if (with_vals) { CK(hipFree(v0_alloc)); CK(hipFree(v1_alloc)); }
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : 87363311ull;
    const size_t m = argc > 2 ? strtoull(argv[2], nullptr, 0) : 33775659ull;
    const int reps = argc > 3 ? atoi(argv[3]) : 10;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    if (n) run("pairs", n, false, 10000000u, 40, reps, s);
    if (m) run("tree", m, true, 1u << 24, 40, reps, s);
    return 0;
}
