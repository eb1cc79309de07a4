// Microbenchmark: write rate of ALIGNED chunks scattered at random -- 128-byte lines (8 lanes x 16 bytes, what the claiming passes
// store) against 64-byte half lines (8 lanes x 8 bytes: the same pass writing 4-byte records), each beside a sequential read stream
// of twice the written volume (the pass reads 8-byte records either way).
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/chunk_bw.hip -o tools/micro/chunk_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned long long u64;
template <int BYTES>  // bytes per lane: 16 -> 128-byte chunks, 8 -> 64-byte chunks
__global__ __launch_bounds__(1024) void chunks(char* __restrict__ dst, const uint4* __restrict__ src, size_t n_chunks, size_t mask, u64 mul) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    for (size_t i = t; i < n_chunks * 8; i += nt) {
        const size_t c = ((i >> 3) * mul) & mask;
        const uint4 v = src[i];  // 16 bytes read per lane either way
        char* d = dst + c * (8 * BYTES) + (i & 7) * BYTES;
        if (BYTES == 16) *reinterpret_cast<uint4*>(d) = v;
        else *reinterpret_cast<uint2*>(d) = make_uint2(v.x ^ v.z, v.y ^ v.w);
    }
}
int main() {
    const size_t n_chunks = (size_t)1 << 23;  // 8 M chunks: 1 GiB of lines or 512 MiB of half lines; 1 GiB read
    char* dst; uint4* src;
    CK(hipMalloc(&dst, n_chunks * 128)); CK(hipMalloc(&src, n_chunks * 128)); CK(hipMemset(src, 1, n_chunks * 128)); CK(hipMemset(dst, 0, n_chunks * 128));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int var = 0; var < 2; ++var) {
        float t = 0;
        for (int rep = 0; rep < 6; ++rep) {
            if (rep == 1) CK(hipEventRecord(e0));
            if (var == 0) chunks<16><<<256, 1024>>>(dst, src, n_chunks, n_chunks - 1, 0x9E3779B97F4A7C15ull | 1ull);
            else chunks<8><<<256, 1024>>>(dst, src, n_chunks, n_chunks - 1, 0x9E3779B97F4A7C15ull | 1ull);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); t /= 5.0f;
        const double rd = (double)n_chunks * 128, wr = (double)n_chunks * (var == 0 ? 128 : 64);
        printf("%3d-byte chunks: %.1f us per launch, read %.2f + write %.2f GB -> %.2f TB/s total, %.2f G records/s\n", var == 0 ? 128 : 64, t * 1e3,
               rd / 1e9, wr / 1e9, (rd + wr) / (t * 1e-3) / 1e12, n_chunks * 16.0 / (t * 1e-3) / 1e9);
    }
    return 0;
}
