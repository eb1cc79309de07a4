// bp_device.h -- device-side helpers shared by the gfx950 kernels (wave64, CDNA4).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bp {

// ---- compile-time index parameters (reference: src/index.rs:293-295) -------------------------
template <int KIND> struct KT;
template <> struct KT<0> { static constexpr uint32_t DIM = 2, BITS = 32, DB = 4, AB = 14; };
template <> struct KT<1> { static constexpr uint32_t DIM = 2, BITS = 64, DB = 5, AB = 29; };
template <> struct KT<2> { static constexpr uint32_t DIM = 3, BITS = 64, DB = 5, AB = 19; };

// bit k of `a` -> bit DIM*k (Morton lane 0).  Restates encode_axis (src/index.rs:155-172,
// 193-207) with the usual doubling masks; the caller has already reduced the u32 coordinate to
// its top AXIS_BITS.
template <int DIM> __device__ __forceinline__ uint64_t spread_bits(uint64_t x);
template <> __device__ __forceinline__ uint64_t spread_bits<2>(uint64_t x) {
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
template <> __device__ __forceinline__ uint64_t spread_bits<3>(uint64_t x) {
    x = (x | (x << 32)) & 0x001F00000000FFFFull;
    x = (x | (x << 16)) & 0x001F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

// inverse of spread_bits: bit DIM*k -> bit k
template <int DIM> __device__ __forceinline__ uint32_t compact_bits(uint64_t x);
template <> __device__ __forceinline__ uint32_t compact_bits<2>(uint64_t x) {
    x &= 0x5555555555555555ull;
    x = (x ^ (x >> 1)) & 0x3333333333333333ull;
    x = (x ^ (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x ^ (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x ^ (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x ^ (x >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)x;
}
template <> __device__ __forceinline__ uint32_t compact_bits<3>(uint64_t x) {
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10C30C30C30C30C3ull;
    x = (x ^ (x >> 4)) & 0x100F00F00F00F00Full;
    x = (x ^ (x >> 8)) & 0x001F0000FF0000FFull;
    x = (x ^ (x >> 16)) & 0x001F00000000FFFFull;
    x = (x ^ (x >> 32)) & 0x00000000001FFFFFull;
    return (uint32_t)x;
}

// level_mask (src/index.rs:82-86)
template <int KIND> __device__ __forceinline__ uint64_t level_mask(uint32_t depth) {
    constexpr uint32_t DIM = KT<KIND>::DIM, TOP = KT<KIND>::DIM * KT<KIND>::AB + KT<KIND>::DB;
    if (depth == 0) return 0ull;
    return ((1ull << (DIM * depth)) - 1ull) << (TOP - DIM * depth);
}
template <int KIND> __device__ __forceinline__ uint32_t key_depth(uint64_t key) {
    return (uint32_t)key & ((1u << KT<KIND>::DB) - 1u);
}

// box code: memory byte (2 bits per axis) <-> 5-bit ternary field of the packed sort payload
__device__ __forceinline__ uint32_t box_to_packed(uint32_t b) {
    return b == 0xFFu ? 31u : (b & 3u) + 3u * ((b >> 2) & 3u) + 9u * ((b >> 4) & 3u);
}
__device__ __forceinline__ uint32_t box_from_packed(uint32_t c) {
    if (c >= 27u) return 0xFFu;
    const uint32_t tz = (c * 57u) >> 9;  // c / 9
    const uint32_t r = c - 9u * tz;
    const uint32_t ty = (r * 11u) >> 5;  // r / 3
    return (r - 3u * ty) | (ty << 2) | (tz << 4);
}

// Rust `f32 as u32`: saturating, NaN -> 0
__device__ __forceinline__ uint32_t f32_to_u32_sat(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

// ---- wave64 primitives --------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// wave-wide vote straight from the compare's lane mask (HIP's __ballot takes an int: the bool is first materialised as
// 0 / 1 in a VGPR and compared again -- two VALU instructions per vote in the hot loops)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// wave-uniform value -> SGPR.  (__builtin_amdgcn_readfirstlane returns int: without the cast back to
// unsigned a 64-bit composition sign-extends its low half.)
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    return ((uint64_t)uniform_u32((uint32_t)(v >> 32)) << 32) | (uint64_t)uniform_u32((uint32_t)v);
}

// inclusive wave64 prefix sum on the DPP data path (no LDS crossbar): row shifts build the scan
// inside each row of 16 lanes, the two row broadcasts carry the row totals forward
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    // update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes masked out or without a source get `old` (0)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xE, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xC, false);  // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_and_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v &= __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ uint64_t wave_or_u64(uint64_t v) {
    uint32_t lo = wave_or_u32((uint32_t)v), hi = wave_or_u32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t wave_and_u64(uint64_t v) {
    uint32_t lo = wave_and_u32((uint32_t)v), hi = wave_and_u32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Exclusive scan across a THREADS-wide workgroup (THREADS a multiple of 64).  Every thread of the
// workgroup must call it.  wave_sums: shared scratch of THREADS/64 words.
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* wave_sums, uint32_t& total) {
    constexpr int NW = THREADS / 64;
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const uint32_t incl = wave_incl_scan_u32(v);
    if (lane == 63) wave_sums[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint32_t s = wave_sums[i];
        if (i < w) base += s;
        tot += s;
    }
    total = tot;
    __syncthreads();
    return base + incl - v;
}

}  // namespace bp
