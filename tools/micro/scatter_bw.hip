// Microbenchmark: HBM write rate of a radix pass's store pattern.  Every wave streams `run` consecutive bytes (8 bytes per
// lane and store instruction, as the pass kernels do) to a pseudo-random run-sized slot of a large buffer; every slot is
// written exactly once per launch.  Variants: slots aligned to the run size or shifted by 8 bytes (a tile's runs start
// wherever the previous tile's ended), stores alone or beside a sequential read stream of the same volume.
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/scatter_bw.hip -o tools/micro/scatter_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned long long u64;

// run_words: 8-byte words per run (multiple of 64 or a divisor of 64); slots: number of runs; mul: odd multiplier (permutation mod 2^k)
template <bool READ>
__global__ __launch_bounds__(256) void scatter(u64* __restrict__ dst, const u64* __restrict__ src, size_t slots, unsigned run_words, size_t mask,
                                               u64 mul, unsigned shift_words, int mode) {
    const unsigned lane = threadIdx.x & 63u;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    if (run_words >= 64) {
        for (size_t s = wave; s < slots; s += n_waves) {
            size_t slot = (s * mul) & mask;  // slots is a power of two
            if (mode) {  // which waves write NEIGHBOURING slots at the same time (do partial lines of two writers merge in L2?)
                const size_t k = s / n_waves, wv = threadIdx.x >> 6, nb = gridDim.x, b = blockIdx.x;
                if (mode == 1) slot = s;                                                     // the waves of one workgroup
                else if (mode == 2) slot = k * n_waves + wv * nb + b;                        // workgroups b, b + 1: different XCDs
                else slot = k * n_waves + wv * nb + (b % 8) * (nb / 8) + b / 8;              // workgroups b, b + 8: one XCD, different CUs
            }
            u64* d = dst + slot * run_words + shift_words;
            const u64* r = src + s * run_words;
            for (unsigned o = lane; o < run_words; o += 64) d[o] = READ ? r[o] : (u64)o;
        }
    } else {  // several runs per store instruction: 64 / run_words slots per wave step
        const unsigned per = 64 / run_words, sub = lane / run_words, o = lane % run_words;
        for (size_t s = wave * per + sub; s < slots; s += n_waves * per) {
            const size_t slot = (s * mul) & mask;
            dst[slot * run_words + shift_words + o] = READ ? src[s * run_words + o] : (u64)o;
        }
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB written per launch
    u64 *dst, *src;
    CK(hipMalloc(&dst, bytes + 4096)); CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes + 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%8s %9s | %22s | %22s\n", "run B", "shift 8B", "stores only: us  TB/s", "read+store: us  TB/s(total)");
    for (int mode = 1; mode <= 3; ++mode) {
        for (unsigned run : {512u, 1024u}) {
            const unsigned run_words = run / 8;
            const size_t slots = bytes / run;
            float t = 0;
            for (int rep = 0; rep < 4; ++rep) {
                if (rep == 1) CK(hipEventRecord(e0));
                scatter<true><<<256 * 8, 256>>>(dst, src, slots, run_words, slots - 1, 1ull, 1, mode);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&t, e0, e1)); t /= 3.0f;
            printf("neighbour mode %d (%s), run %u B shifted by 8: read+store %.1f us %.2f TB/s\n", mode,
                   mode == 1 ? "same workgroup" : mode == 2 ? "workgroups b, b+1 (other XCD)" : "workgroups b, b+8 (same XCD)", run, t * 1e3,
                   2.0 * bytes / (t * 1e-3) / 1e12);
        }
    }
    for (unsigned run : {64u, 128u, 256u, 512u, 1024u, 2048u, 4096u, 16384u}) {
        for (unsigned shift : {0u, 1u}) {
            const unsigned run_words = run / 8;
            const size_t slots = bytes / run;
            float t[2];
            for (int rd = 0; rd < 2; ++rd) {
                for (int rep = 0; rep < 4; ++rep) {
                    if (rep == 1) CK(hipEventRecord(e0));
                    if (rd) scatter<true><<<256 * 8, 256>>>(dst, src, slots, run_words, slots - 1, 0x9E3779B97F4A7C15ull | 1ull, shift, 0);
                    else scatter<false><<<256 * 8, 256>>>(dst, src, slots, run_words, slots - 1, 0x9E3779B97F4A7C15ull | 1ull, shift, 0);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&t[rd], e0, e1)); t[rd] /= 3.0f;
            }
            printf("%8u %9s | %10.1f %10.2f | %10.1f %10.2f\n", run, shift ? "yes" : "no", t[0] * 1e3, bytes / (t[0] * 1e-3) / 1e12, t[1] * 1e3,
                   2.0 * bytes / (t[1] * 1e-3) / 1e12);
        }
    }
    return 0;
}
