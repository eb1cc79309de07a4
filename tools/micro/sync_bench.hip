// Microbenchmark: how long does the host wait for 64 bytes of device scalars?
//  (a) hipMemcpyAsync D2H + hipStreamSynchronize   (b) a kernel stores them to mapped host memory, the host polls a sequence word
// build: hipcc -O2 --offload-arch=gfx950 sync_bench.hip -o sync_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
struct Sc { unsigned long long v[15]; unsigned long long seq; };
__global__ void work(unsigned long long* d, int spin) {
    unsigned long long x = d[threadIdx.x & 7];
    for (int i = 0; i < spin; ++i) x = x * 6364136223846793005ull + 1442695040888963407ull;
    if (threadIdx.x == 0) d[0] = x;
}
__global__ void publish(const Sc* d, Sc* h, unsigned long long seq) {
    if (threadIdx.x < 15) h->v[threadIdx.x] = d->v[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&h->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    Sc* d; hipMalloc(&d, sizeof(Sc)); hipMemset(d, 0, sizeof(Sc));
    Sc* h; hipHostMalloc(&h, sizeof(Sc), hipHostMallocMapped); std::memset(h, 0, sizeof(Sc));
    Sc* hd; hipHostGetDevicePointer((void**)&hd, h, 0);
    Sc* h2; hipHostMalloc(&h2, sizeof(Sc), hipHostMallocDefault);
    const int iters = 2000;
    for (int spin : {0, 20000}) {
        for (int mode = 0; mode < 3; ++mode) {
            hipStreamSynchronize(s);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 1; i <= iters; ++i) {
                work<<<1, 64, 0, s>>>((unsigned long long*)d, spin);
                if (mode == 0) {
                    hipMemcpyAsync(h2, d, sizeof(Sc), hipMemcpyDeviceToHost, s);
                    hipStreamSynchronize(s);
                } else if (mode == 1) {
                    publish<<<1, 64, 0, s>>>(d, hd, (unsigned long long)i + (unsigned long long)spin * 100000ull);
                    const unsigned long long want = (unsigned long long)i + (unsigned long long)spin * 100000ull;
                    while (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) != want) __builtin_ia32_pause();
                } else {
                    hipStreamSynchronize(s);
                }
            }
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
            printf("kernel spin %5d  %-28s %7.2f us per round\n", spin, mode == 0 ? "memcpyAsync+streamSync" : mode == 1 ? "publish kernel + host poll" : "streamSync only", us);
        }
    }
    return 0;
}
