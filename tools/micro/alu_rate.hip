// Microbenchmark: issue rate of the integer instructions the kernels lean on -- 32-bit shift / add (the yardstick), v_lshlrev_b64 / v_lshrrev_b64,
// v_mul_lo_u32, v_mul_u32_u24, v_mad_u64_u32, v_alignbit_b32 -- each as a chain-free stream of 8 independent accumulators per lane.
// build: hipcc -O2 --offload-arch=gfx950 tools/micro/alu_rate.hip -o tools/micro/alu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 2; } } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;
template <int OP>
__global__ __launch_bounds__(256) void rate(u64* out, u32 s, int iters) {
    u64 a[8];
    u32 b[8];
    for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * 0x9E3779B97F4A7C15ull + k; b[k] = threadIdx.x * 2654435761u + k; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == 0) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(b[k]) : "s"(s));
            if (OP == 1) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a[k]) : "s"(s));
            if (OP == 2) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a[k]) : "s"(s));
            if (OP == 3) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b[k]) : "v"(b[(k + 1) & 7]));
            if (OP == 4) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(b[k]) : "v"(b[(k + 1) & 7]));
            if (OP == 5) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[k]) : "v"(b[k]), "v"(b[(k + 1) & 7]) : "vcc");
            if (OP == 6) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(b[k]) : "v"(b[(k + 1) & 7]), "s"(s));
            if (OP == 7) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(b[k]) : "v"(b[(k + 1) & 7]) : "vcc");
            if (OP == 8) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(b[k]) : "v"(b[(k + 1) & 7]));
            if (OP == 9) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(b[k]) : "v"(b[(k + 1) & 7]));
        }
    }
    u64 r = 0;
    for (int k = 0; k < 8; ++k) r += a[k] + b[k];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    u64* out; CK(hipMalloc(&out, 1024 * 8 * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"v_lshlrev_b32", "v_lshlrev_b64", "v_lshrrev_b64", "v_mul_lo_u32", "v_mul_u32_u24", "v_mad_u64_u32", "v_alignbit_b32", "v_add_co_u32", "v_mul_hi_u32", "v_mul_hi_u32_u24"};
    const int iters = 4096;
    for (int op = 0; op < 10; ++op) {
        float t = 0;
        for (int rep = 0; rep < 3; ++rep) {
            if (rep == 1) CK(hipEventRecord(e0));
            switch (op) {
                case 0: rate<0><<<1024 * 8, 256>>>(out, 3, iters); break; case 1: rate<1><<<1024 * 8, 256>>>(out, 3, iters); break;
                case 2: rate<2><<<1024 * 8, 256>>>(out, 3, iters); break; case 3: rate<3><<<1024 * 8, 256>>>(out, 3, iters); break;
                case 4: rate<4><<<1024 * 8, 256>>>(out, 3, iters); break; case 5: rate<5><<<1024 * 8, 256>>>(out, 3, iters); break;
                case 6: rate<6><<<1024 * 8, 256>>>(out, 3, iters); break; case 7: rate<7><<<1024 * 8, 256>>>(out, 3, iters); break;
                case 8: rate<8><<<1024 * 8, 256>>>(out, 3, iters); break; default: rate<9><<<1024 * 8, 256>>>(out, 3, iters); break;
            }
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t, e0, e1)); t /= 2.0f;
        const double insts = 1024.0 * 8 * 4 /*waves per wg*/ * iters * 8;  // wave-instructions
        printf("%-16s %8.3f ms  %.2f cycles per wave-instruction and SIMD (2.4 GHz, 1024 SIMDs)\n", names[op], t, t * 1e-3 * 2.4e9 * 1024 / insts);
    }
    return 0;
}
