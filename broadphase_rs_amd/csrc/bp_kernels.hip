// bp_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Layer extend -> sort -> scan
// path.  Integer / bit-op work, HBM-bound: no MFMA.  Reference behaviour (file:line into
// zvxryb/broadphase-rs) is cited at each kernel; the algorithms themselves are GPU-first:
//   encode      count -> block-sum scan -> emit (deterministic append order)
//   radix sort  one-sweep LSD, 8-bit digits, decoupled look-back across 8192-key tiles,
//               wave64 ballot ranking, LDS-staged coalesced scatter, constant-bit digit skipping
//   overlap scan closed form of the reference's stack sweep: ancestors-by-search, count -> emit
//   dedup       adjacent-unique flag/compact
#include "bp_kernels.h"
#include "bp_device.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace bp {
namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr u32 MAX_CELLS_PER_OBJECT = 1u << 20;

// shard scans: does a COARSE entry (depth < coarse_depth, replicated on every rank) act as the deeper side of a pair here?
// emit_coarse 0: never, 1: always, 2: when its id lies in [coarse_id_lo, coarse_id_hi] (the rank's first-id range)
__device__ __forceinline__ bool coarse_emits(const ScanParams& p, u32 id) {
    return p.emit_coarse == 1 || (p.emit_coarse == 2 && id >= p.coarse_id_lo && id <= p.coarse_id_hi);
}

// =================================================================================================
// encode: Layer::extend (src/layer.rs:94-121)
// =================================================================================================
template <int KIND> struct Cells {
    u32 depth;
    u32 lo[3];   // first cell coordinate per axis (depth bits)
    u32 n[3];    // cells per axis
    u32 count;   // product (0 if the object is invalid or overflows)
    bool valid, overflow;
};

// contains (src/geom.rs:121-128) + to_local (src/geom.rs:148-163) + indices() depth choice
// (src/geom.rs:189-200 / 247-258) + the cell range of indices_at_depth (src/geom.rs:202-231 /
// 260-297).  f32 arithmetic is evaluated exactly as written there: IEEE sub, div, mul, add with
// no contraction (-ffp-contract=off) and a saturating cast.
template <int KIND>
__device__ __forceinline__ Cells<KIND> object_cells(const SysBounds& sb, const float* __restrict__ row, u32 min_depth) {
    constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB;
    Cells<KIND> c;
    c.valid = true;
    c.overflow = false;
    float bmin[3], bmax[3];
#pragma unroll
    for (u32 a = 0; a < DIM; ++a) {
        bmin[a] = row[a];
        bmax[a] = row[DIM + a];
        if (sb.mn[a] > bmin[a] || sb.mx[a] < bmax[a]) c.valid = false;  // NaN passes, like the reference
    }
    u32 lmin[3], lmax[3], max_axis = 0;
#pragma unroll
    for (u32 a = 0; a < DIM; ++a) {
        const float size = sb.mx[a] - sb.mn[a];
        lmin[a] = f32_to_u32_sat((bmin[a] - sb.mn[a]) / size * 4294967040.0f + 0.0f);
        lmax[a] = f32_to_u32_sat((bmax[a] - sb.mn[a]) / size * 4294967040.0f + 0.0f);
        const u32 s = (lmax[a] - lmin[a]) + 1u;  // sizei(), wrapping like release-mode Rust
        max_axis = s > max_axis ? s : max_axis;
    }
    const u32 ext = max_axis - 1u;
    u32 depth = ext == 0 ? 32u : (u32)__clz(ext);
    depth = depth < min_depth ? min_depth : depth;
    depth = depth < AB ? depth : AB;  // clamp_depth (src/index.rs:93-95)
    c.depth = depth;
    u64 count = 1;
#pragma unroll
    for (u32 a = 0; a < 3; ++a) {
        c.lo[a] = 0;
        c.n[a] = 1;
    }
    if (depth != 0) {
        const u32 sh = 32u - depth;
#pragma unroll
        for (u32 a = 0; a < DIM; ++a) {
            const u32 lo = lmin[a] >> sh, hi = lmax[a] >> sh;
            c.lo[a] = lo;
            c.n[a] = hi >= lo ? hi - lo + 1u : 1u;  // the reference loop emits min once when min > max
            count *= c.n[a];
            if (count > MAX_CELLS_PER_OBJECT) {
                c.overflow = true;
                count = MAX_CELLS_PER_OBJECT + 1ull;
            }
        }
    }
    if (!c.valid || c.overflow) count = 0;
    c.count = (u32)count;
    return c;
}

// Index::default().set_depth(depth).set_origin(cell << (32-depth))  (src/index.rs:106-112, 230-250)
template <int KIND>
__device__ __forceinline__ u64 make_key(u32 depth, u32 cx, u32 cy, u32 cz) {
    constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB, DB = KT<KIND>::DB;
    const u32 up = AB - depth;  // cell coordinate -> top-AXIS_BITS coordinate
    u64 morton = spread_bits<DIM>((u64)cx << up) | (spread_bits<DIM>((u64)cy << up) << 1);
    if (DIM == 3) morton |= spread_bits<DIM>((u64)cz << up) << 2;
    return (morton << DB) | (u64)depth;
}

// OR / AND of all integers in [lo, hi]
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u32 t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}

__device__ __forceinline__ u32 wave_min_u32(u32 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const u32 t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}

__device__ __forceinline__ void range_or_and(u32 lo, u32 n, u32& o, u32& a) {
    const u32 hi = lo + n - 1u;
    const u32 x = lo ^ hi;
    const u32 low = x == 0 ? 0u : (0xFFFFFFFFu >> __clz(x));
    o = hi | low;
    a = lo & ~low;
}

// Per-object record the count pass leaves for the emit pass (which then neither reads the bounds again nor repeats the
// divisions): x, y, z = first cell per axis; w = depth | cells on axis 0 / 1 / 2 (1 or 2) at bits 8 / 10 / 12 | ENC_REC_OK.
// ENC_REC_SLOW instead: more than 2 cells on some axis (min_depth forced it) -- x = entry count, the emit pass recomputes
// the object from its bounds.  w == 0: no entries (outside the system bounds, or refused as too large).
enum : u32 { ENC_REC_OK = 1u << 16, ENC_REC_SLOW = 1u << 17 };
__device__ __forceinline__ u32 enc_rec_count(const uint4& r) {
    if (r.w & ENC_REC_OK) return ((r.w >> 8) & 3u) * ((r.w >> 10) & 3u) * ((r.w >> 12) & 3u);
    return (r.w & ENC_REC_SLOW) ? r.x : 0u;
}

template <int KIND>
__global__ __launch_bounds__(ENC_THREADS) void encode_count_kernel(SysBounds sb, const float* __restrict__ bounds,
                                                                  const u32* __restrict__ ids, size_t n,
                                                                  u32 min_depth, int have_prev, u32 prev_id,
                                                                  uint4* __restrict__ recs, u64* __restrict__ block_sums,
                                                                  BlockAux* __restrict__ aux, DevScalars* sc,
                                                                  u64* __restrict__ zero_a, u32 n_a, u64* __restrict__ zero_b, u32 n_b) {
    constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB, DB = KT<KIND>::DB;
    // the frame's zero fills ride along (the histograms encode_emit adds into): every workgroup clears its slice -- one
    // workgroup of scan_block_sums spent a third of its time on the 256 KB
    for (size_t k = (size_t)blockIdx.x * ENC_THREADS + threadIdx.x; k < n_a; k += (size_t)gridDim.x * ENC_THREADS) zero_a[k] = 0ull;
    for (size_t k = (size_t)blockIdx.x * ENC_THREADS + threadIdx.x; k < n_b; k += (size_t)gridDim.x * ENC_THREADS) zero_b[k] = 0ull;
    __shared__ u32 s_sum[ENC_THREADS / 64], s_dm[ENC_THREADS / 64], s_fl[ENC_THREADS / 64], s_er[ENC_THREADS / 64],
        s_inv[ENC_THREADS / 64], s_ior[ENC_THREADS / 64], s_iand[ENC_THREADS / 64], s_imax[ENC_THREADS / 64],
        s_imin[ENC_THREADS / 64];
    __shared__ u64 s_kor[ENC_THREADS / 64], s_kand[ENC_THREADS / 64];
    // ENC_ROUNDS blocks of ENC_THREADS objects per workgroup: a quarter of the partials for scan_block_sums to fold
    u32 cnt = 0, dmask = 0, flags = 0, err = 0, inv = 0, ior = 0, iand = 0xFFFFFFFFu, imax = 0, imin = 0xFFFFFFFFu;
    // OR / AND of every index value of the workgroup's objects, without enumerating cells and without spreading per object:
    // the Morton spread is bitwise (it distributes over OR and AND, and the axes' lanes are disjoint), so the coordinates
    // are accumulated per axis -- as top-AXIS_BITS coordinates, whatever the object's depth -- and spread once at the end
    u32 co[3] = {0, 0, 0}, ca[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, dor = 0, dand = 0xFFFFFFFFu;
    for (int round = 0; round < ENC_ROUNDS; ++round) {
        const size_t i = ((size_t)blockIdx.x * ENC_ROUNDS + round) * ENC_THREADS + threadIdx.x;
        if (i >= n) break;
        const Cells<KIND> c = object_cells<KIND>(sb, bounds + i * (2 * DIM), min_depth);
        const u32 id = ids[i];
        u32 prev = prev_id;
        bool hp = have_prev != 0;
        if (i > 0) {
            prev = ids[i - 1];
            hp = true;
        }
        if (hp) {
            if (id < prev) flags |= EXT_IDS_DECREASE | EXT_IDS_REPEAT;
            else if (id == prev) flags |= EXT_IDS_REPEAT;
        }
        imax = id > imax ? id : imax;
        imin = id < imin ? id : imin;
        if (c.overflow && c.valid) err |= KERR_COUNT_OVERFLOW;
        if (!c.valid) inv += 1;
        const u32 c_cnt = c.count;
        cnt += c_cnt;
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        if (c_cnt > 0) {
            const bool boxy = c.n[0] <= 2 && c.n[1] <= 2 && c.n[2] <= 2;
            if (!boxy) flags |= EXT_BOX_UNKNOWN;
            dmask |= 1u << c.depth;
            ior |= id;
            iand &= id;
            const u32 up = AB - c.depth;
#pragma unroll
            for (u32 ax = 0; ax < DIM; ++ax) {
                u32 o, a;
                range_or_and(c.lo[ax], c.n[ax], o, a);
                co[ax] |= o << up;
                ca[ax] &= a << up;
            }
            dor |= c.depth;
            dand &= c.depth;
            rec = boxy ? make_uint4(c.lo[0], c.lo[1], c.lo[2], c.depth | (c.n[0] << 8) | (c.n[1] << 10) | (c.n[2] << 12) | ENC_REC_OK)
                       : make_uint4(c_cnt, 0u, 0u, ENC_REC_SLOW);
        }
        recs[i] = rec;
    }
    u64 kor = 0, kand = ~0ull;
    if (dmask != 0) {  // this thread saw at least one entry
        u64 mo = spread_bits<DIM>((u64)co[0]) | (spread_bits<DIM>((u64)co[1]) << 1);
        u64 ma = spread_bits<DIM>((u64)ca[0]) | (spread_bits<DIM>((u64)ca[1]) << 1);
        if (DIM == 3) {
            mo |= spread_bits<DIM>((u64)co[2]) << 2;
            ma |= spread_bits<DIM>((u64)ca[2]) << 2;
        }
        kor = (mo << DB) | dor;
        kand = (ma << DB) | dand;
    }
    // workgroup reduction -> one BlockAux + one block sum per workgroup (no atomics)
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const u32 ws = wave_sum_u32(cnt), wd = wave_or_u32(dmask), wf = wave_or_u32(flags), we = wave_or_u32(err),
              wi = wave_sum_u32(inv), wo = wave_or_u32(ior), wa = wave_and_u32(iand), wm = wave_max_u32(imax),
              wn = wave_min_u32(imin);
    const u64 wko = wave_or_u64(kor), wka = wave_and_u64(kand);
    if (lane == 0) {
        s_sum[w] = ws; s_dm[w] = wd; s_fl[w] = wf; s_er[w] = we; s_inv[w] = wi; s_ior[w] = wo; s_iand[w] = wa; s_imax[w] = wm; s_imin[w] = wn;
        s_kor[w] = wko; s_kand[w] = wka;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 tot = 0;
        BlockAux b;
        b.key_or = 0; b.key_and = ~0ull; b.depth_mask = 0; b.flags = 0; b.err = 0; b.n_invalid = 0; b.id_or = 0;
        b.id_and = 0xFFFFFFFFu; b.id_max = 0; b.id_min = 0xFFFFFFFFu;
        for (int k = 0; k < ENC_THREADS / 64; ++k) {
            tot += s_sum[k];
            b.key_or |= s_kor[k]; b.key_and &= s_kand[k]; b.depth_mask |= s_dm[k]; b.flags |= s_fl[k];
            b.err |= s_er[k]; b.n_invalid += s_inv[k]; b.id_or |= s_ior[k]; b.id_and &= s_iand[k];
            b.id_max = s_imax[k] > b.id_max ? s_imax[k] : b.id_max;
            b.id_min = s_imin[k] < b.id_min ? s_imin[k] : b.id_min;
        }
        block_sums[blockIdx.x] = tot;
        aux[blockIdx.x] = b;
    }
}

// digit histograms counted by a producer kernel for the radix sort that follows
struct DigitCount {
    u64* hist;       // [passes][256], zeroed by the host; nullptr / passes == 0: nothing to count
    int passes;      // <= SX_HIST_MAX
    int shift[SX_HIST_MAX];
    // multi-GPU: entries per cell at cell_depth (slot n_cells: entries coarser than that), HIST_COPIES copies of
    // n_cells + 1 counters (zeroed by the host); nullptr: off.  n_cells <= ENC_CELLS_MAX.
    u64* cell_hist;
    int cell_shift;
    u32 cell_depth, n_cells, depth_mask;
};

// spread_bits<DIM>((c + 1) << up) from m = spread_bits<DIM>(c << up): an addition in the dilated number (the bits between
// the lane's bits are filled with ones so that the carry runs through them)
template <int DIM> __device__ __forceinline__ u64 dilated_next(u64 m, u32 up) {
    constexpr u64 LANE = DIM == 3 ? 0x1249249249249249ull : 0x5555555555555555ull;
    return ((m | ~LANE) + (1ull << (DIM * up))) & LANE;
}

template <int KIND, bool CELLS>
__global__ __launch_bounds__(ENC_THREADS) void encode_emit_kernel(SysBounds sb, const float* __restrict__ bounds,
                                                                 const u32* __restrict__ ids, size_t n, u32 min_depth,
                                                                 const uint4* __restrict__ recs,
                                                                 const u64* __restrict__ block_offs,
                                                                 u64* __restrict__ keys_out, u32* __restrict__ ids_out,
                                                                 uint8_t* __restrict__ box_out, DigitCount dc, int pack_box,
                                                                 DevScalars* sc_guard, u64 capacity) {
    // pack_box: the box code rides in bits 27..31 of the id from the start (ids < 2^27, see BOX_ID_BITS) and
    // no box byte is written: the first pass of the tree sort then has no byte stream to gather
    // sc_guard / capacity (launch without a host round trip after the count pass): sc_guard->total is the entry total the
    // block-sum scan left on the device; beyond `capacity` entries nothing is written and KERR_ENCODE_OVERFLOW is set
    constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB, DB = KT<KIND>::DB;
    if (sc_guard && sc_guard->total > capacity) {  // grid-uniform
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&sc_guard->err, (u32)KERR_ENCODE_OVERFLOW);
        return;
    }
    __shared__ u32 s_ws[ENC_THREADS / 64];
    __shared__ u32 s_hist[SX_HIST_MAX * 256];
    __shared__ u32 s_cell[CELLS ? ENC_CELLS_MAX + 1 : 1];  // CELLS: a separate instantiation, the default kernel pays nothing
    for (int k = (int)threadIdx.x; k < dc.passes * 256; k += ENC_THREADS) s_hist[k] = 0;  // visible after the scan's barriers
    if (CELLS)
        for (u32 k = threadIdx.x; k <= dc.n_cells; k += ENC_THREADS) s_cell[k] = 0;
    auto count_digits = [&](u64 key) {
#pragma unroll
        for (int q = 0; q < SX_HIST_MAX; ++q) {
            if (q < dc.passes)
                __hip_atomic_fetch_add(&s_hist[q * 256 + ((u32)(key >> dc.shift[q]) & 255u)], 1u, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (CELLS) {
            const u32 slot = ((u32)key & dc.depth_mask) < dc.cell_depth ? dc.n_cells : (u32)(key >> dc.cell_shift);
            __hip_atomic_fetch_add(&s_cell[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // ENC_ROUNDS blocks of ENC_THREADS objects per workgroup; block_offs[blockIdx.x] = first entry of the workgroup.
    // A round's entries (<= 8 per object unless min_depth forces more) are staged in LDS in output order and
    // stored with consecutive threads on consecutive entries; a round that does not fit stores directly.
    __shared__ u64 s_key[ENC_STAGE];
    __shared__ u32 s_id[ENC_STAGE];
    __shared__ uint8_t s_box[ENC_STAGE];
    size_t round_base = (size_t)block_offs[blockIdx.x];
    for (int round = 0; round < ENC_ROUNDS; ++round) {
        const size_t blk = (size_t)blockIdx.x * ENC_ROUNDS + round;
        if (blk * ENC_THREADS >= n) break;  // workgroup-uniform
        const size_t i = blk * ENC_THREADS + threadIdx.x;
        const uint4 rec = i < n ? recs[i] : make_uint4(0u, 0u, 0u, 0u);
        const u32 cnt = enc_rec_count(rec);
        u32 total;
        const u32 excl = block_excl_scan_u32<ENC_THREADS>(cnt, s_ws, total);
        const size_t this_base = round_base;
        round_base += total;
        const bool staged = total <= (u32)ENC_STAGE;  // workgroup-uniform
        auto put = [&](u32 local, u64 key, u32 idv, u32 code) {
            if (staged) {
                s_key[local] = key;
                s_id[local] = idv;
                s_box[local] = (uint8_t)code;
            } else {
                keys_out[this_base + local] = key;
                ids_out[this_base + local] = idv;
                if (box_out) box_out[this_base + local] = (uint8_t)code;
            }
        };
        if (cnt != 0) {
            const u32 id = ids[i];
            u32 local = excl;
            if (rec.w & ENC_REC_OK) {
                const u32 depth = rec.w & 0xFFu;
                if (depth == 0) {  // Index::default(): src/geom.rs:203-205 / 261-263
                    put(local, 0ull, id, 0u);  // box code 0 packs to 0
                    count_digits(0ull);
                } else {
                    // at most two cells per axis: both Morton lanes of every axis once per object (the second from the first by
                    // a dilated increment), an entry is then two ORs; cells in z outer, y, x inner order: src/geom.rs:271-297
                    // (2-D: y outer, x inner: src/geom.rs:213-231)
                    const u32 up = AB - depth;
                    const u32 s0 = ((rec.w >> 8) & 3u) - 1u, s1 = ((rec.w >> 10) & 3u) - 1u, s2 = DIM == 3 ? ((rec.w >> 12) & 3u) - 1u : 0u;
                    // per axis: the index value's contribution of the first cell and what the second cell changes in it (XOR)
                    const u64 m0 = spread_bits<DIM>((u64)rec.x << up), m1 = spread_bits<DIM>((u64)rec.y << up);
                    const u64 dx = (m0 ^ dilated_next<DIM>(m0, up)) << DB, dy = (m1 ^ dilated_next<DIM>(m1, up)) << (DB + 1);
                    u64 key0 = ((m0 | (m1 << 1)) << DB) | (u64)depth, dz = 0;
                    if (DIM == 3) {
                        const u64 m2 = spread_bits<DIM>((u64)rec.z << up);
                        key0 |= m2 << (DB + 2);
                        dz = (m2 ^ dilated_next<DIM>(m2, up)) << (DB + 2);
                    }
                    // box code of the first cell (ternary, the form that rides in the id) and what a step on each axis adds
                    const u32 t0 = s0 + 3u * s1 + 9u * s2, idp = pack_box ? id : 0u;
                    for (u32 e = 0; e < cnt; ++e) {
                        const u32 x = e & s0, y = (e >> s0) & s1, z = e >> (s0 + s1);
                        const u64 key = key0 ^ (x ? dx : 0ull) ^ (y ? dy : 0ull) ^ (z ? dz : 0ull);
                        const u32 packed = t0 + x + 3u * y + 9u * z;  // tx + 3 ty + 9 tz with t = 0 (one cell) or 1 + step
                        if (pack_box) {
                            if (staged) { s_key[local] = key; s_id[local] = idp | (packed << BOX_ID_BITS); }
                            else { keys_out[this_base + local] = key; ids_out[this_base + local] = idp | (packed << BOX_ID_BITS); }
                        } else {
                            put(local, key, id, box_from_packed(packed));
                        }
                        count_digits(key);
                        ++local;
                    }
                }
            } else {  // more than two cells on some axis (min_depth forced it): from the bounds, cell by cell
                const Cells<KIND> c = object_cells<KIND>(sb, bounds + i * (2 * DIM), min_depth);
                for (u32 z = 0; z < c.n[2]; ++z)
                    for (u32 y = 0; y < c.n[1]; ++y)
                        for (u32 x = 0; x < c.n[0]; ++x) {
                            const u64 key = make_key<KIND>(c.depth, c.lo[0] + x, c.lo[1] + y, c.lo[2] + z);
                            put(local, key, pack_box ? (id | (box_to_packed((u32)BOX_UNKNOWN) << BOX_ID_BITS)) : id, (u32)BOX_UNKNOWN);
                            count_digits(key);
                            ++local;
                        }
            }
        }
        if (staged) {
            __syncthreads();
            for (u32 k = threadIdx.x; k < total; k += ENC_THREADS) {
                keys_out[this_base + k] = s_key[k];
                ids_out[this_base + k] = s_id[k];
                if (box_out) box_out[this_base + k] = s_box[k];
            }
            __syncthreads();  // the stage is reused by the next round
        }
    }
    if (CELLS) {
        __syncthreads();
        u64* copy = dc.cell_hist + (size_t)(blockIdx.x % HIST_COPIES) * (dc.n_cells + 1);
        for (u32 k = threadIdx.x; k <= dc.n_cells; k += ENC_THREADS) {
            const u32 v = s_cell[k];
            if (v) atomicAdd(&copy[k], (u64)v);
        }
    }
    if (dc.passes) {  // workgroup-uniform.  One device atomic per bin per workgroup would serialise on the bin's
        __syncthreads();  // address (~11 ns each): the workgroups spread over HIST_COPIES copies, summed by the sort
        u64* copy = dc.hist + (size_t)(blockIdx.x % HIST_COPIES) * (SX_HIST_MAX * 256);
        for (int k = (int)threadIdx.x; k < dc.passes * 256; k += ENC_THREADS) {
            const u32 v = s_hist[k];
            if (v) atomicAdd(&copy[k], (u64)v);
        }
    }
}

// The emit pass for COMPACT records (TreeCompact, bp_kernels.h): one composite word per entry -- index bits | id | box code --
// goes to words_out and nothing else is written; the digits counted are the composite's.  One thread per ENTRY, not per object:
// a thread per object runs its cell loop to the wave's longest (eight cells, with 3.4 on average: two lanes in five busy).  Here
// every lane prepares its object (first cell's index value, what a step on each axis changes in it, id, code, depth term) and
// leaves it in LDS; the wave's entries -- consecutive in the output -- are then dealt to the lanes 64 at a time, an entry finds
// its object from a row of end-of-range bits (v_mbcnt: the expanding scan's device) and the words leave coalesced, without a
// staging pass.  A round with an object of more than two cells on an axis (min_depth forced it) takes the per-object loop.
template <int KIND>
__global__ __launch_bounds__(ENC_THREADS) void encode_emit_compact_kernel(SysBounds sb, const float* __restrict__ bounds,
                                                                          const u32* __restrict__ ids, size_t n, u32 min_depth,
                                                                          const uint4* __restrict__ recs, const u64* __restrict__ block_offs,
                                                                          u64* __restrict__ words_out, DigitCount dc, DevScalars* sc_guard,
                                                                          u64 capacity, TreeCompact fmt, int key_shift_delta) {
    constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB, DB = KT<KIND>::DB, CBASE = (1u << DIM) + 1u;
    if (sc_guard && sc_guard->total > capacity) {  // grid-uniform
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(&sc_guard->err, (u32)KERR_ENCODE_OVERFLOW);
        return;
    }
    constexpr int NW = ENC_THREADS / 64;
    __shared__ u32 s_ws[NW];
    __shared__ u32 s_hist[SX_HIST_MAX * 256];
    __shared__ u32 s_pow[16];
    __shared__ __attribute__((aligned(16))) uint4 s_obj[NW][3][64];  // per wave and object with entries: {key0, dx} {dy, dz} {id << 5 | code, depth term, first entry | cell bits}
    __shared__ u64 s_ends[NW][8];                                     // bit e: some object's last entry is the wave's entry e
    for (int k = (int)threadIdx.x; k < dc.passes * 256; k += ENC_THREADS) s_hist[k] = 0;  // visible after the scan's barriers
    if (threadIdx.x == 0) {
        u32 pw = 1;
        for (int k = 0; k <= fmt.T && k < 16; ++k) { s_pow[k] = pw; pw *= CBASE; }
    }
    auto count_digits = [&](u64 key) {
#pragma unroll
        for (int q = 0; q < SX_HIST_MAX; ++q) {
            if (q < dc.passes)
                __hip_atomic_fetch_add(&s_hist[q * 256 + ((u32)(key >> dc.shift[q]) & 255u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    // index value -> composite word: the Morton bits of the levels every entry has, the deeper levels as digits 1 + octant in base
    // 2^DIM + 1 (0 below the entry's own depth), then id and code.  tail = sum over the levels the entry has of (1 + octant)
    // BASE^(T - k) = sum over ALL tail levels of octant BASE^(T - k) (the index value has zeros below its depth) + (BASE^T -
    // BASE^(T - nlev)) / (BASE - 1): two instructions per level and one table read per object
    const int c_tsh = fmt.pshift - (int)DIM * fmt.T;  // the tail levels' Morton bits start here in the index value
    auto comp_depth_term = [&](u32 depth) -> u32 {    // (depth >= dmin: the host checks the frame's depths against the format)
        const u32 nlev = min(depth - fmt.dmin, (u32)fmt.T);
        return (s_pow[fmt.T] - s_pow[(u32)fmt.T - nlev]) >> DIM;
    };
    auto comp_of = [&](u64 key, u32 dterm, u32 idcode) -> u64 {
        const u64 kk = key >> c_tsh;                           // [prefix | Morton bits of the tail levels]
        const u32 lv = (u32)kk;                                // (DIM * T <= 30 bits of it are the tail's)
        u32 tail = 0;
        for (int k = 1; k <= fmt.T; ++k)                       // (wave-uniform trip count)
            tail = tail * CBASE + ((lv >> ((int)DIM * (fmt.T - k))) & ((1u << DIM) - 1u));
        return (((kk >> ((int)DIM * fmt.T)) << fmt.tbits | (u64)(tail + dterm)) << fmt.low) | (u64)idcode;
    };
    const int lane = lane_id(), w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    size_t round_base = (size_t)block_offs[blockIdx.x];
    // (a round's record and id are loaded a round ahead: the round then starts with its block scan, not with a memory latency)
    const size_t i0 = (size_t)blockIdx.x * ENC_ROUNDS * ENC_THREADS + threadIdx.x;
    uint4 rec_next = i0 < n ? recs[i0] : make_uint4(0u, 0u, 0u, 0u);
    u32 id_next = i0 < n ? ids[i0] : 0u;
    for (int round = 0; round < ENC_ROUNDS; ++round) {
        const size_t blk = (size_t)blockIdx.x * ENC_ROUNDS + round;
        if (blk * ENC_THREADS >= n) break;  // workgroup-uniform
        const size_t i = blk * ENC_THREADS + threadIdx.x;
        const uint4 rec = rec_next;
        const u32 id = id_next - fmt.id_base;  // (what the word's id field holds; the host checks the frame's ids against base and width)
        if (round + 1 < ENC_ROUNDS) {
            const size_t i1 = i + ENC_THREADS;
            rec_next = i1 < n ? recs[i1] : make_uint4(0u, 0u, 0u, 0u);
            id_next = i1 < n ? ids[i1] : 0u;
        }
        const u32 cnt = enc_rec_count(rec);
        u32 total;
        const u32 excl = block_excl_scan_u32<ENC_THREADS>(cnt, s_ws, total);  // (its barriers also fence the previous round's LDS use)
        const size_t this_base = round_base;
        round_base += total;
        if (__syncthreads_or((rec.w & ENC_REC_SLOW) != 0u)) {
            // an object that spans more than two cells on an axis: every thread walks its own object's cells
            if (cnt != 0) {
                u32 local = excl;
                if (rec.w & ENC_REC_OK) {
                    const u32 depth = rec.w & 0xFFu;
                    const u32 n0 = (rec.w >> 8) & 3u, n1 = (rec.w >> 10) & 3u, n2 = DIM == 3 ? (rec.w >> 12) & 3u : 1u;
                    const u32 dterm = comp_depth_term(depth);
                    for (u32 z = 0; z < n2; ++z)
                        for (u32 y = 0; y < n1; ++y)
                            for (u32 x = 0; x < n0; ++x) {
                                const u64 key = depth == 0 ? 0ull : make_key<KIND>(depth, rec.x + x, rec.y + y, rec.z + z);
                                const u32 packed = (n0 - 1u + x) * (n0 > 1u) + 3u * (n1 - 1u + y) * (n1 > 1u) + 9u * (n2 - 1u + z) * (n2 > 1u);
                                const u64 comp = comp_of(key, dterm, (id << 5) | packed);
                                words_out[this_base + local] = comp;
                                count_digits(comp);
                                ++local;
                            }
                } else {
                    const Cells<KIND> c = object_cells<KIND>(sb, bounds + i * (2 * DIM), min_depth);
                    const u32 dterm = comp_depth_term(c.depth);
                    for (u32 z = 0; z < c.n[2]; ++z)
                        for (u32 y = 0; y < c.n[1]; ++y)
                            for (u32 x = 0; x < c.n[0]; ++x) {
                                const u64 comp = comp_of(make_key<KIND>(c.depth, c.lo[0] + x, c.lo[1] + y, c.lo[2] + z), dterm, (id << 5) | (u32)BOX_PACKED_UNKNOWN);
                                words_out[this_base + local] = comp;
                                count_digits(comp);
                                ++local;
                            }
                }
            }
            continue;
        }
        // ---- the wave's objects -> LDS, its entries -> lanes ------------------------------------------------------------
        const u32 incl = wave_incl_scan_u32(cnt);
        const u32 wtotal = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        const size_t wave_base = this_base + (size_t)(u32)__builtin_amdgcn_readfirstlane((int)(excl - (incl - cnt)));
        const u64 nz = ballot64(cnt != 0u);
        if (lane < 8) s_ends[w][lane] = 0ull;
        __builtin_amdgcn_wave_barrier();
        if (cnt != 0u) {
            const u32 depth = rec.w & 0xFFu;
            u64 key0 = 0, dx = 0, dy = 0, dz = 0;
            u32 sbits = 0, t0 = 0;
            if (depth != 0) {  // (depth 0: Index::default(), one entry, src/geom.rs:203-205 / 261-263)
                // at most two cells per axis: the first cell's Morton lanes once per object, the second cell of an axis by a dilated
                // increment; cells in z outer, y, x inner order: src/geom.rs:271-297 (2-D: y outer, x inner: src/geom.rs:213-231)
                const u32 up = AB - depth;
                const u32 s0 = ((rec.w >> 8) & 3u) - 1u, s1 = ((rec.w >> 10) & 3u) - 1u, s2 = DIM == 3 ? ((rec.w >> 12) & 3u) - 1u : 0u;
                const u64 m0 = spread_bits<DIM>((u64)rec.x << up), m1 = spread_bits<DIM>((u64)rec.y << up);
                dx = (m0 ^ dilated_next<DIM>(m0, up)) << DB;
                dy = (m1 ^ dilated_next<DIM>(m1, up)) << (DB + 1);
                key0 = ((m0 | (m1 << 1)) << DB) | (u64)depth;
                if (DIM == 3) {
                    const u64 m2 = spread_bits<DIM>((u64)rec.z << up);
                    key0 |= m2 << (DB + 2);
                    dz = (m2 ^ dilated_next<DIM>(m2, up)) << (DB + 2);
                }
                sbits = s0 | (s1 << 1) | (s2 << 2);
                t0 = s0 + 3u * s1 + 9u * s2;  // the first cell's code (ternary); a step on an axis adds 1 / 3 / 9
            }
            // The digits counted lie in the part of the word that is the index value shifted (the groups' bits: key_shift_delta
            // takes a digit's shift back to the index value).  Every cell of the object is key0 XOR a subset of {dx, dy, dz}, and
            // the three deltas occupy DISJOINT Morton lanes (x, y, z bits never share a position): so the XOR of the selected deltas
            // has a zero digit exactly when each selected delta does, and where the first and the far-corner cell (= all deltas
            // selected) agree on a digit every cell of the object does -- one LDS add of the cell count for the object instead of one
            // per entry (five objects in six on the uniform scene; the adds were a third of the LDS pipe's time).  (NOT because the
            // cells lie between the two in index order -- agreement of two ends on one 8-bit digit would not imply agreement in
            // between; a change that lets the deltas share a lane, e.g. counting the format's tail digits, breaks this silently:
            // tests/test_compact_format.py::test_digit_shortcut_of_the_emit_pass cross-checks per-object against per-entry counts)
            u32 per_entry = 0;  // bit q: pass q's digit differs inside the object -- its entries count themselves
            {
                const u64 kmax = key0 ^ ((sbits & 1u) ? dx : 0ull) ^ ((sbits & 2u) ? dy : 0ull) ^ ((sbits & 4u) ? dz : 0ull);
#pragma unroll
                for (int q = 0; q < SX_HIST_MAX; ++q) {
                    if (q < dc.passes) {
                        const int ks = dc.shift[q] + key_shift_delta;
                        const u32 d0 = (u32)(key0 >> ks) & 255u, d1 = (u32)(kmax >> ks) & 255u;
                        if (d0 == d1) __hip_atomic_fetch_add(&s_hist[q * 256 + d0], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else per_entry |= 1u << q;
                    }
                }
            }
            const u32 slot = __builtin_amdgcn_mbcnt_hi((u32)(nz >> 32), __builtin_amdgcn_mbcnt_lo((u32)nz, 0u));
            s_obj[w][0][slot] = make_uint4((u32)key0, (u32)(key0 >> 32), (u32)dx, (u32)(dx >> 32));
            s_obj[w][1][slot] = make_uint4((u32)dy, (u32)(dy >> 32), (u32)dz, (u32)(dz >> 32));
            s_obj[w][2][slot] = make_uint4((id << 5) | t0, comp_depth_term(depth), (incl - cnt) | (sbits << 16) | (per_entry << 20), 0u);
            const u32 last = incl - 1u;
            __hip_atomic_fetch_or(&s_ends[w][last >> 6], 1ull << (last & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_wave_barrier();
        u32 owners_before = 0;  // objects whose entries all lie before the current 64 (wave-uniform)
        for (u32 e0 = 0; e0 < wtotal; e0 += 64u) {  // wave-uniform
            const u64 word = s_ends[w][e0 >> 6];
            const u32 e = e0 + (u32)lane;
            const u32 o = __builtin_amdgcn_mbcnt_hi((u32)(word >> 32), __builtin_amdgcn_mbcnt_lo((u32)word, owners_before));
            owners_before += (u32)__popcll(word);
            if (e < wtotal) {
                const uint4 c = s_obj[w][2][o], a = s_obj[w][0][o], b = s_obj[w][1][o];
                const u32 el = e - (c.z & 0xFFFFu), sb3 = (c.z >> 16) & 7u;
                const u32 s0 = sb3 & 1u, s1 = (sb3 >> 1) & 1u;
                const u32 x = el & s0, y = (el >> s0) & s1, z = el >> (s0 + s1);
                const u64 key = (((u64)a.y << 32) | a.x) ^ (x ? (((u64)a.w << 32) | a.z) : 0ull) ^ (y ? (((u64)b.y << 32) | b.x) : 0ull) ^
                                (z ? (((u64)b.w << 32) | b.z) : 0ull);
                const u64 comp = comp_of(key, c.y, c.x + x + 3u * y + 9u * z);
                words_out[wave_base + e] = comp;
                const u32 pe = c.z >> 20;
                if (pe) {  // (an object that straddles a digit's boundary)
#pragma unroll
                    for (int q = 0; q < SX_HIST_MAX; ++q)
                        if (q < dc.passes && ((pe >> q) & 1u))
                            __hip_atomic_fetch_add(&s_hist[q * 256 + ((u32)(comp >> dc.shift[q]) & 255u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    if (dc.passes) {  // (see encode_emit_kernel: the workgroups spread their counts over HIST_COPIES copies)
        __syncthreads();
        u64* copy = dc.hist + (size_t)(blockIdx.x % HIST_COPIES) * (SX_HIST_MAX * 256);
        for (int k = (int)threadIdx.x; k < dc.passes * 256; k += ENC_THREADS) {
            const u32 v = s_hist[k];
            if (v) atomicAdd(&copy[k], (u64)v);
        }
    }
}

// decoupled look-back status words (fused encode below, radix one-sweep further down)
constexpr u64 ST_FLAG_AGG = 1ull << 62;   // tile-local digit count published
constexpr u64 ST_FLAG_PRE = 2ull << 62;   // inclusive prefix over tiles 0..t published
constexpr u64 ST_VALUE = (1ull << 32) - 1ull;                 // counts / prefixes (inputs are shorter than 2^32 records)
constexpr u32 ST_EPOCH_MASK = (1u << 30) - 1u;                 // launch tag, bits 32..61
constexpr u64 ST_TAG_BITS = (u64)ST_EPOCH_MASK << 32;
constexpr u32 LOOKBACK_SPIN_LIMIT = 1u << 22;
constexpr u32 HOLE_BLOCK = SX_BLOCK;  // records per block of a block-structured input (expanding scan output)

__device__ __forceinline__ void st_store(u64* p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 st_load(u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(RS_BINS) void hist_reduce_kernel(const u64* __restrict__ copies, int n_copies, u64* __restrict__ hist) {
    const size_t k = (size_t)blockIdx.x * RS_BINS + threadIdx.x;  // blockIdx.x = pass
    u64 t = 0;
    for (int c = 0; c < n_copies; ++c) t += copies[(size_t)c * (SX_HIST_MAX * 256) + k];
    hist[k] = t;
}

// single-workgroup exclusive scan of per-block sums; folds the encode partials into *sc
constexpr int SBS_THREADS = 1024;
constexpr int SBS_ITEMS = 8;
__device__ __forceinline__ u64 wave_incl_scan_u64(u64 v) {
    const int lane = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const u32 lo = __shfl_up((u32)v, o, 64), hi = __shfl_up((u32)(v >> 32), o, 64);
        if (lane >= o) v += ((u64)hi << 32) | lo;
    }
    return v;
}
// results the host waits for, written straight into pinned host memory: the data first, then (system-scope release) the number
// of the launch in *flag -- the host spins on the flag (bp_layer.hip: wait_published)
__device__ __forceinline__ void publish_flag(const Published& pub) {
    __threadfence_system();
    __hip_atomic_store(pub.flag, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Between the count and the emit kernel of an extend this kernel is also where the frame's small resets ride along: *sc starts
// from `init` (set_init) instead of an H2D copy, and the histograms the emit kernel adds into are zero-filled here (zero_a / zero_b).
__global__ __launch_bounds__(SBS_THREADS) void scan_block_sums_kernel(u64* __restrict__ sums, size_t n,
                                                                     const BlockAux* __restrict__ aux, DevScalars* sc, DevScalars init,
                                                                     int set_init, u64* __restrict__ zero_a, u32 n_a, u64* __restrict__ zero_b,
                                                                     u32 n_b, Published pub) {
    for (u32 k = threadIdx.x; k < n_a; k += SBS_THREADS) zero_a[k] = 0ull;
    for (u32 k = threadIdx.x; k < n_b; k += SBS_THREADS) zero_b[k] = 0ull;
    __shared__ u64 s_wave[SBS_THREADS / 64];
    __shared__ u64 s_carry;
    __shared__ u64 r_kor[SBS_THREADS / 64], r_kand[SBS_THREADS / 64];
    __shared__ u32 r_dm[SBS_THREADS / 64], r_fl[SBS_THREADS / 64], r_er[SBS_THREADS / 64], r_inv[SBS_THREADS / 64],
        r_ior[SBS_THREADS / 64], r_iand[SBS_THREADS / 64], r_imax[SBS_THREADS / 64], r_imin[SBS_THREADS / 64];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    // the per-workgroup partials of encode_count, folded per thread FIRST: their loads are in flight while the block sums are scanned
    u64 kor = 0, kand = ~0ull;
    u32 dm = 0, fl = 0, er = 0, inv = 0, ior = 0, iand = 0xFFFFFFFFu, imax = 0, imin = 0xFFFFFFFFu;
    if (aux) {
        // (four partials per step, loaded together: one memory latency per four instead of one each -- at 10 M objects a thread
        // folds ten of them, and the kernel is one workgroup waiting for its own loads)
        auto fold = [&](const BlockAux& b) {
            kor |= b.key_or; kand &= b.key_and; dm |= b.depth_mask; fl |= b.flags; er |= b.err; inv += b.n_invalid;
            ior |= b.id_or; iand &= b.id_and; imax = b.id_max > imax ? b.id_max : imax; imin = b.id_min < imin ? b.id_min : imin;
        };
        size_t i = threadIdx.x;
        for (; i + 3 * (size_t)SBS_THREADS < n; i += 4 * (size_t)SBS_THREADS) {
            const BlockAux b0 = aux[i], b1 = aux[i + SBS_THREADS], b2 = aux[i + 2 * (size_t)SBS_THREADS], b3 = aux[i + 3 * (size_t)SBS_THREADS];
            fold(b0); fold(b1); fold(b2); fold(b3);
        }
        for (; i < n; i += SBS_THREADS) fold(aux[i]);
    }
    for (size_t base = 0; base < n; base += (size_t)SBS_THREADS * SBS_ITEMS) {
        const size_t i0 = base + (size_t)threadIdx.x * SBS_ITEMS;
        u64 v[SBS_ITEMS], tsum = 0;
#pragma unroll
        for (int k = 0; k < SBS_ITEMS; ++k) {
            v[k] = (i0 + k) < n ? sums[i0 + k] : 0ull;
            tsum += v[k];
        }
        const u64 incl = wave_incl_scan_u64(tsum);
        if (lane == 63) s_wave[w] = incl;
        __syncthreads();
        u64 wbase = 0, tot = 0;
        for (int k = 0; k < SBS_THREADS / 64; ++k) {
            const u64 s = s_wave[k];
            if (k < w) wbase += s;
            tot += s;
        }
        const u64 carry = s_carry;
        u64 run = carry + wbase + incl - tsum;
#pragma unroll
        for (int k = 0; k < SBS_ITEMS; ++k) {
            if ((i0 + k) < n) sums[i0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (aux) {  // reduce the threads' folds over the workgroup
        kor = wave_or_u64(kor); kand = wave_and_u64(kand); dm = wave_or_u32(dm); fl = wave_or_u32(fl);
        er = wave_or_u32(er); inv = wave_sum_u32(inv); ior = wave_or_u32(ior); iand = wave_and_u32(iand);
        imax = wave_max_u32(imax);
        imin = wave_min_u32(imin);
        if (lane == 0) {
            r_imax[w] = imax;
            r_imin[w] = imin;
            r_kor[w] = kor; r_kand[w] = kand; r_dm[w] = dm; r_fl[w] = fl; r_er[w] = er; r_inv[w] = inv;
            r_ior[w] = ior; r_iand[w] = iand;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (set_init) *sc = init;
            for (int k = 0; k < SBS_THREADS / 64; ++k) {
                sc->key_or |= r_kor[k]; sc->key_and &= r_kand[k]; sc->depth_mask |= r_dm[k]; sc->flags |= r_fl[k];
                sc->err |= r_er[k]; sc->n_invalid += r_inv[k]; sc->id_or |= r_ior[k]; sc->id_and &= r_iand[k];
                sc->id_max = r_imax[k] > sc->id_max ? r_imax[k] : sc->id_max;
                sc->id_min = r_imin[k] < sc->id_min ? r_imin[k] : sc->id_min;
            }
        }
    }
    if (threadIdx.x == 0) {
        sc->total = s_carry;
        if (pub.dst) {  // the host polls pinned memory for this launch's number instead of waiting for a copy behind the kernel
            *reinterpret_cast<DevScalars*>(pub.dst) = *sc;
            publish_flag(pub);
        }
    }
}
// a block of device words -> pinned host memory, then the launch's number (Published): what a D2H copy + stream synchronisation does,
// without the copy engine's hand-over and the completion interrupt
__global__ __launch_bounds__(256) void publish_kernel(const u32* __restrict__ src, u32 words, Published pub, const u32* __restrict__ src2, u32 words2,
                                                      u32 off2) {
    u32* dst = reinterpret_cast<u32*>(pub.dst);
    for (u32 k = threadIdx.x; k < words; k += 256u) dst[k] = src[k];
    for (u32 k = threadIdx.x; k < words2; k += 256u) dst[off2 + k] = src2[k];
    __syncthreads();  // (every thread's stores precede thread 0's fence in program order of the workgroup: the fence below is device-wide)
    if (threadIdx.x == 0) publish_flag(pub);
}


// OR / AND / depth mask over an existing tree (set_tree, append_entries)
template <int KIND>
__global__ __launch_bounds__(256) void tree_reduce_kernel(const u64* __restrict__ keys, const u32* __restrict__ ids,
                                                          size_t m, DevScalars* sc) {
    u64 kor = 0, kand = ~0ull;
    u32 dm = 0, ior = 0, iand = 0xFFFFFFFFu;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256) {
        const u64 k = keys[i];
        const u32 id = ids[i];
        kor |= k; kand &= k; dm |= 1u << key_depth<KIND>(k); ior |= id; iand &= id;
    }
    kor = wave_or_u64(kor); kand = wave_and_u64(kand); dm = wave_or_u32(dm); ior = wave_or_u32(ior);
    iand = wave_and_u32(iand);
    if (lane_id() == 0) {
        atomicOr(&sc->key_or, kor);
        atomicAnd(&sc->key_and, kand);
        atomicOr(&sc->depth_mask, dm);
        atomicOr(&sc->id_or, ior);
        atomicAnd(&sc->id_and, iand);
    }
}

// =================================================================================================
// radix sort: one-sweep LSD with decoupled look-back
// =================================================================================================
constexpr int HIST_THREADS = 512;
// (one block per CU: every block ends with one global atomic per bin and pass on the SAME words -- 1024 blocks queued 1024 deep on each
// of them, ~11 ns a piece; measured on the sorts that count their own digits: one rank's shard sort of configs[3] 0.462 -> 0.444 ms,
// the one-shot 10 M frame 2.74 -> 2.64)
#ifndef HIST_MAX_BLOCKS_DEF
#define HIST_MAX_BLOCKS_DEF 256
#endif
constexpr int HIST_MAX_BLOCKS = HIST_MAX_BLOCKS_DEF;

__global__ __launch_bounds__(HIST_THREADS) void radix_hist_kernel(const u64* __restrict__ keys,
                                                                  const u32* __restrict__ vals, size_t n,
                                                                  RadixPlan plan, u64* __restrict__ hist,
                                                                  const u32* __restrict__ blockfill) {
    __shared__ u32 h[RS_MAX_PASSES * RS_BINS];
    const int np = plan.n_passes;
    for (int j = threadIdx.x; j < np * RS_BINS; j += HIST_THREADS) h[j] = 0;
    __syncthreads();
    // (four records per thread and step, their loads issued together: with one block per CU -- see HIST_MAX_BLOCKS -- a thread walks
    // many records, and one dependent load per step left a large input at a third of the rate)
    constexpr int UNR = 4;
    const size_t step = (size_t)gridDim.x * HIST_THREADS;
    for (size_t i0 = (size_t)blockIdx.x * HIST_THREADS + threadIdx.x; i0 < n; i0 += UNR * step) {
        u64 k[UNR];
        u32 v[UNR];
        bool on[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const size_t i = i0 + (size_t)q * step;
            on[q] = i < n;
            if (on[q] && blockfill) {  // 64 consecutive records per wave: the block is wave-uniform
                const u32 blk = uniform_u32((u32)(i / HOLE_BLOCK));
                on[q] = (u32)(i & (HOLE_BLOCK - 1u)) < blockfill[blk];  // (else a hole)
            }
            k[q] = on[q] ? keys[i] : 0ull;
            v[q] = (on[q] && vals) ? vals[i] : 0u;
        }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            if (!on[q]) continue;
            for (int p = 0; p < np; ++p) {
                const u32 dg = plan.from_val[p] ? ((v[q] >> plan.shift[p]) & 0xFFu) : ((u32)(k[q] >> plan.shift[p]) & 0xFFu);
                atomicAdd(&h[p * RS_BINS + dg], 1u);
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < np * RS_BINS; j += HIST_THREADS) {
        const u32 c = h[j];
        if (c) atomicAdd(&hist[j], (u64)c);
    }
}

// one digit histogram with an arbitrary digit functor (partition passes)
template <typename DigitFn>
__global__ __launch_bounds__(HIST_THREADS) void digit_hist_kernel(const u64* __restrict__ keys, size_t n, DigitFn digit,
                                                                  u64* __restrict__ hist, const u32* __restrict__ blockfill) {
    __shared__ u32 h[RS_BINS];
    for (int j = threadIdx.x; j < RS_BINS; j += HIST_THREADS) h[j] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * HIST_THREADS + threadIdx.x; i < n; i += (size_t)gridDim.x * HIST_THREADS) {
        if (blockfill && (u32)(i & (HOLE_BLOCK - 1u)) >= blockfill[uniform_u32((u32)(i / HOLE_BLOCK))]) continue;  // hole
        atomicAdd(&h[digit(keys[i])], 1u);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < RS_BINS; j += HIST_THREADS) {
        const u32 c = h[j];
        if (c) atomicAdd(&hist[j], (u64)c);
    }
}

// multi-GPU splitter selection: entries per cell at split_depth (+ one slot for coarse entries)
__global__ __launch_bounds__(256) void cell_hist_kernel(const u64* __restrict__ keys, size_t m, int cell_shift, u32 depth_mask,
                                                        u32 split_depth, u32 n_cells, u64* __restrict__ hist) {
    extern __shared__ u32 h_dyn[];
    for (u32 j = threadIdx.x; j <= n_cells; j += 256) h_dyn[j] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256) {
        const u64 k = keys[i];
        const u32 slot = ((u32)k & depth_mask) < split_depth ? n_cells : (u32)(k >> cell_shift);
        atomicAdd(&h_dyn[slot], 1u);
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j <= n_cells; j += 256) {
        const u32 c = h_dyn[j];
        if (c) atomicAdd(&hist[j], (u64)c);
    }
}

// per pass: exclusive scan of the 256 global digit counts -> bin_base
__global__ __launch_bounds__(RS_BINS) void radix_binbase_kernel(const u64* __restrict__ hist, u64* __restrict__ bin_base) {
    __shared__ u64 s_wave[RS_BINS / 64];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const u64 v = hist[(size_t)blockIdx.x * RS_BINS + threadIdx.x];
    const u64 incl = wave_incl_scan_u64(v);
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    u64 base = 0;
    for (int k = 0; k < w; ++k) base += s_wave[k];
    bin_base[(size_t)blockIdx.x * RS_BINS + threadIdx.x] = base + incl - v;
}

// Everything in front of the sharded engine's partition pass in ONE launch, its small inputs as kernel arguments: the owner table and
// the bucket sizes (host data: the splitters were just computed there) used to be two H2D copies, the scalars' reset a third, the
// workspace a memset, the sizes a D2D copy, the bases a kernel -- six operations of a few microseconds each with the host's launch
// latency between them, on a path where the GPU has nothing else queued (one rank's step of configs[3]: ~30 us).
struct PartSetupArgs {
    unsigned char lut[PART_SETUP_MAX_CELLS];   // cell -> owner
    unsigned long long hist[RS_BINS];          // bucket sizes
};
__global__ __launch_bounds__(RS_BINS) void partition_setup_kernel(PartSetupArgs a, u32 n_cells, unsigned char* __restrict__ d_lut,
                                                                  u64* __restrict__ ws_words, size_t ws_n_words, u64* __restrict__ hist,
                                                                  u64* __restrict__ bin_base, DevScalars* sc, DevScalars sc_init) {
    // every workgroup zero-fills its slice of the workspace (tile tickets and the look-back's status words) ...
    for (size_t k = (size_t)blockIdx.x * RS_BINS + threadIdx.x; k < ws_n_words; k += (size_t)gridDim.x * RS_BINS)
        if (k >= 2 * (size_t)RS_BINS) ws_words[k] = 0ull;   // (the first 512 words are hist and bin_base: written below)
    if (blockIdx.x != 0) return;
    // ... and the first one the rest
    for (u32 c = threadIdx.x; c < n_cells; c += RS_BINS) d_lut[c] = a.lut[c];
    if (threadIdx.x == 0) *sc = sc_init;
    __shared__ u64 s_wave[RS_BINS / 64];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const u64 v = a.hist[threadIdx.x];
    hist[threadIdx.x] = v;
    const u64 incl = wave_incl_scan_u64(v);
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    u64 base = 0;
    for (int k = 0; k < w; ++k) base += s_wave[k];
    bin_base[threadIdx.x] = base + incl - v;
}

// one wave: the sizes of a finished expanding scan from its shard cursors and counts (launch_scan_finalize, or the first
// kernel of the sort enqueued behind the scan)
__device__ __forceinline__ void scan_finalize_wave(const u64* __restrict__ shards, u64 out_blocks, u64 n_bound, u64 extent_bound,
                                                   const DevScalars* __restrict__ sc, u32* __restrict__ sizes) {
    static_assert(SX_SHARDS == 64, "one lane per shard");
    const int lane = lane_id();
    u64 kmax = shards[lane], total = shards[SX_SHARDS + lane];
    for (int o = 32; o; o >>= 1) {
        const u64 k2 = ((u64)(u32)__shfl_xor((int)(u32)(kmax >> 32), o, 64) << 32) | (u32)__shfl_xor((int)(u32)kmax, o, 64);
        const u64 t2 = ((u64)(u32)__shfl_xor((int)(u32)(total >> 32), o, 64) << 32) | (u32)__shfl_xor((int)(u32)total, o, 64);
        kmax = kmax > k2 ? kmax : k2;
        total += t2;
    }
    if (lane != 0) return;
    const u64 extent_blocks = kmax * (u64)SX_SHARDS, extent = extent_blocks * (u64)SX_BLOCK;
    const bool ok = extent_blocks <= out_blocks && total <= n_bound && extent <= extent_bound && extent < 0xFFFFFFFFull &&
                    !(sc->err & (u32)KERR_BIGLIST_OVERFLOW);
    sizes[SIZES_EXTENT] = ok ? (u32)extent : 0u;
    sizes[SIZES_N] = ok ? (u32)total : 0u;
    sizes[SIZES_MISS] = ok ? 0u : 1u;
    sizes[SIZES_BIG] = sc->big_count;  // (the sort's first kernel resets the scalars: what the host wants of them is kept here)
}
#ifndef RS_CUR_STRIDE
#define RS_CUR_STRIDE 1040  // u32 words between two groups of 16 bin cursors (4160 bytes): the groups spread over memory channels
#endif
constexpr u32 RS_CUR_STRIDE_WORDS = RS_CUR_STRIDE;
constexpr int RS_TICKETS = 1;  // ticket counters per pass
// The front end of a sort in one launch (blockIdx.x = pass): digit counts (summed over the producer's copies, or already
// in `hist`), their exclusive prefix = the bins' bases, the pass's tile ticket back to zero and -- optionally -- the
// scalars the sort's kernels report into reset to `sc_init`.
__device__ __forceinline__ void radix_prepare_body(const u64* __restrict__ copies, int n_copies, u64* __restrict__ hist,
                                                   u64* __restrict__ bin_base, u32* __restrict__ counters, DevScalars* sc,
                                                   const DevScalars& sc_init, int set_sc, u32* __restrict__ cursors, const ScanFinalize& fin) {
    __shared__ u64 s_wave[RS_BINS / 64];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    // the sizes of the scan this sort was enqueued behind (fin.sc may be the very scalars thread 0 resets below: read first)
    if (fin.shards && blockIdx.x == 0 && w == 0)
        scan_finalize_wave((const u64*)fin.shards, (u64)fin.out_blocks, (u64)fin.n_bound, (u64)fin.extent_bound, fin.sc, fin.sizes);
    const size_t k = (size_t)blockIdx.x * RS_BINS + threadIdx.x;
    u64 v;
    if (copies) {
        v = 0;
#pragma unroll 8
        for (int c = 0; c < n_copies; ++c) v += copies[(size_t)c * (SX_HIST_MAX * 256) + k];  // independent loads in flight
        hist[k] = v;
    } else {
        v = hist[k];
    }
    const u64 incl = wave_incl_scan_u64(v);
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    u64 base = 0;
    for (int q = 0; q < w; ++q) base += s_wave[q];
    bin_base[k] = base + incl - v;
    // radix_claim_kernel (first pass): the bins' line and fragment cursors start from zero
    if (cursors && blockIdx.x == 0) {
        const u32 slot = (threadIdx.x >> 4) * RS_CUR_STRIDE_WORDS + (threadIdx.x & 15u);
        cursors[slot] = 0u;
        cursors[16 * RS_CUR_STRIDE_WORDS + slot] = 0u;
    }
    if (threadIdx.x == 0) {
        for (int j = 0; j < RS_TICKETS; ++j) counters[blockIdx.x * RS_TICKETS + j] = 0;
        if (set_sc && blockIdx.x == 0) *sc = sc_init;
    }
}
__global__ __launch_bounds__(RS_BINS) void radix_prepare_kernel(const u64* __restrict__ copies, int n_copies, u64* __restrict__ hist,
                                                                u64* __restrict__ bin_base, u32* __restrict__ counters, DevScalars* sc,
                                                                DevScalars sc_init, int set_sc, u32* __restrict__ cursors = nullptr,
                                                                ScanFinalize fin = ScanFinalize{}) {
    radix_prepare_body(copies, n_copies, hist, bin_base, counters, sc, sc_init, set_sc, cursors, fin);
}

// digit functors: what a one-sweep pass buckets by
struct ShiftDigit {  // LSD radix digit
    int shift;
    template <typename K> __device__ __forceinline__ u32 operator()(K k) const { return (u32)(k >> shift) & 0xFFu; }
};
struct EntryDestDigit {  // multi-GPU: owner rank of an index entry (Morton range), coarse entries -> extra bucket
    const unsigned char* lut;  // [cells at split_depth] -> owner
    int cell_shift;
    u32 depth_mask, split_depth, coarse_bucket;
    __device__ __forceinline__ u32 operator()(u64 k) const {
        return ((u32)k & depth_mask) < split_depth ? coarse_bucket : (u32)lut[(size_t)(k >> cell_shift)];
    }
};
struct PairOwnerDigit {  // multi-GPU: owner rank of a packed pair = range of its first id
    u32 bounds[15];  // first id owned by rank 1, 2, ...
    int n;
    __device__ __forceinline__ u32 operator()(u64 k) const {
        const u32 id = (u32)(k >> 32);
        u32 d = 0;
        for (int i = 0; i < n; ++i) d += id >= bounds[i] ? 1u : 0u;
        return d;
    }
};

template <typename T> struct StageBytes { static constexpr size_t value = sizeof(T); };
struct NoVal {};
template <> struct StageBytes<NoVal> { static constexpr size_t value = 0; };

// One radix pass over n records.  K: type carrying the digit (u64 index value / packed pair, or
// u32 id); V: payload type (u32 id, u64 index value) or NoVal.  Stable.
//
// Phase order inside a tile (chosen so that no tile ever sits idle waiting for its predecessors):
//   1 ticket + load keys
//   4 rank (per-wave digit counters, lane-mask match in LDS)
//   2/3 tile digit counts -> publish AGGREGATE at once
//   5 tile-local scatter of the keys into LDS (needs no global information)
//   6 look-back for the global digit bases (predecessors have had phase 5 to publish), PREFIX publish
//   7 coalesced global stores (+ payload round trip through LDS)
#ifndef RS_MIN_WAVES_PER_EU
#define RS_MIN_WAVES_PER_EU 1
#endif
// STABLE = false: records of equal digit may leave a tile in any order (one returning LDS atomic per record instead of the
// mask-match rounds).  Valid for the FIRST pass of any LSD sort that does not rely on the input order of equal keys.
// FEW: the digit takes a handful of values (multi-GPU bucketing by owner, <= 17 buckets): a wave ranks a round by walking
// the distinct digits it holds (first-lane broadcast + ballot), since per-digit LDS words would have all 64 lanes on a few
// addresses.  Stable.
template <typename K, typename V, int THREADS, int ITEMS, typename DigitFn, bool STABLE = true, bool FEW = false>
__global__ __launch_bounds__(THREADS, RS_MIN_WAVES_PER_EU) void radix_onesweep_kernel(const K* __restrict__ keys_in, K* __restrict__ keys_out,
                                                                    const V* __restrict__ vals_in, V* __restrict__ vals_out,
                                                                    size_t n, DigitFn digit, const u64* __restrict__ bin_base,
                                                                    u64* status, u32* tile_counter, DevScalars* sc,
                                                                    const uint8_t* __restrict__ box_in, uint8_t* __restrict__ box_out,
                                                                    const u32* __restrict__ blockfill, size_t n_out, u32 epoch,
                                                                    const u32* __restrict__ dev_sizes) {
    // dev_sizes: the sizes live on the device (n / n_out are upper bounds that sized the grid): tiles beyond the end leave
    if (dev_sizes) {
        n_out = dev_sizes[SIZES_N];
        n = blockfill ? dev_sizes[SIZES_EXTENT] : dev_sizes[SIZES_N];
    }
    // epoch: tag of this launch's status words (bits 32..61; counts and prefixes are < 2^32).  A word with another tag is
    // a leftover of an earlier launch on the same workspace and reads as "nothing published": no zero-fill between launches
    const u64 st_tag = (u64)(epoch & ST_EPOCH_MASK) << 32;
    constexpr int RS_THREADS = THREADS, RS_ITEMS = ITEMS, RS_TILE = THREADS * ITEMS, RS_WAVES = THREADS / 64;
    static_assert(THREADS >= RS_BINS && RS_WAVES * RS_BINS * 8 <= RS_TILE * (int)sizeof(K), "tile geometry");
    constexpr bool HAS_VAL = StageBytes<V>::value != 0;
    constexpr size_t STAGE = RS_TILE * (sizeof(K) > StageBytes<V>::value ? sizeof(K) : StageBytes<V>::value);
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[STAGE];
#if defined(RS_CNT16) && RS_CNT16
    __shared__ unsigned short s_cnt[RS_WAVES][RS_BINS];  // per-wave counts <= 64 * ITEMS, tile offsets < RS_TILE <= 65535
    static_assert(RS_TILE <= 65535, "16-bit tile offsets");
#else
    __shared__ u32 s_cnt[RS_WAVES][RS_BINS];
#endif
    __shared__ u64 s_glob[RS_BINS];
    __shared__ u32 s_ws[RS_WAVES];
    __shared__ u32 s_tile;

    const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(tile_counter, 1u);  // ticket: tile t only ever waits on tiles < t
#pragma unroll
    for (int j = tid; j < RS_WAVES * RS_BINS; j += RS_THREADS) {
        (&s_cnt[0][0])[j] = 0;
        reinterpret_cast<u64*>(s_stage)[j] = 0ull;  // per-(wave, digit) lane masks of the ranking phase
    }
    __syncthreads();
    const u32 tile = s_tile;
    const size_t base = (size_t)tile * RS_TILE;
    if (base >= n) return;  // (workgroup-uniform; only with dev_sizes: no tile before this one waits for it)
    const u32 tile_n = (u32)((n - base) < (size_t)RS_TILE ? (n - base) : (size_t)RS_TILE);

    // ---- 1. load: wave w owns RS_ITEMS * 64 consecutive records, 64 per round (coalesced) -------
    // blockfill != nullptr: the input is the block-structured raw pair stream of the expanding scan
    // (HOLE_BLOCK records per block, only the first blockfill[b] of block b exist); the pass reads
    // around the holes and its output is dense (n_out records).
    K key[RS_ITEMS];
    u32 packed[RS_ITEMS];  // digit | rank-in-(wave,digit) << 8, later the tile-local destination
    u32 vmask = 0;         // bit r: record r of this lane exists
    const u32 wbase = (u32)w * (64 * RS_ITEMS) + (u32)lane;
    // a wave's 64 records of one round lie in one block, and its RS_ITEMS rounds in consecutive blocks:
    // the fills are wave-uniform (scalar loads)
    static_assert((64 * RS_ITEMS) % HOLE_BLOCK == 0 && RS_TILE % HOLE_BLOCK == 0, "waves must not straddle blocks");
    constexpr int WAVE_BLOCKS = 64 * RS_ITEMS / (int)HOLE_BLOCK;
    u32 fill[WAVE_BLOCKS];
    if (blockfill) {
        const size_t blk0 = (size_t)uniform_u32((u32)((base + (size_t)w * (64 * RS_ITEMS)) / HOLE_BLOCK));
        const size_t n_blk = (n + HOLE_BLOCK - 1) / HOLE_BLOCK;
#pragma unroll
        for (int q = 0; q < WAVE_BLOCKS; ++q) fill[q] = blk0 + q < n_blk ? blockfill[blk0 + q] : 0u;
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const u32 li = wbase + (u32)r * 64u;
        bool valid = li < tile_n;
        if (blockfill) valid = valid && (li & (HOLE_BLOCK - 1u)) < fill[r * 64 / (int)HOLE_BLOCK];
        vmask |= valid ? (1u << r) : 0u;
        key[r] = valid ? keys_in[base + li] : (K)0;
    }
    u64* my_status = status + (size_t)tile * RS_BINS + tid;

    // ---- 4. rank: peers of a key = lanes of this wave holding the same digit in this round.  Each
    // lane ORs its lane bit into an LDS mask word per (wave, digit) -- OR is order-independent, so
    // the result is deterministic -- then reads the word back: one ds_or_b64 + one ds_read_b64
    // replace an 8-ballot match.  The masks alias the (not yet used) staging buffer.
    u64* s_mask = reinterpret_cast<u64*>(s_stage) + (size_t)w * RS_BINS;
    const u64 lanebit = 1ull << lane;
    if constexpr (FEW) {
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            const bool valid = (vmask >> r) & 1u;
            const u32 dg = digit(key[r]);
            u64 todo = ballot64(valid);
            u32 pos = 0;
            while (todo) {  // wave-uniform: one trip per distinct digit of the round
                const int leader = __builtin_ctzll(todo);
                const u32 d0 = (u32)__builtin_amdgcn_readlane((int)dg, leader);
                const u64 m = ballot64(valid && dg == d0);
                const u32 cnt = s_cnt[w][d0];  // same address for the whole wave: a broadcast read
                if (valid && dg == d0) pos = cnt + __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
                __builtin_amdgcn_wave_barrier();
                if (lane == leader) s_cnt[w][d0] = cnt + (u32)__popcll(m);
                __builtin_amdgcn_wave_barrier();
                todo &= ~m;
            }
            packed[r] = dg | (pos << 8);
        }
    } else if constexpr (!STABLE && sizeof(s_cnt[0][0]) == 4) {   // (16-bit counters have no native LDS atomic: stable rounds then)
        // all RS_ITEMS atomics of a lane are independent: they issue back to back, one LDS round trip for the lot
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            const u32 dg = digit(key[r]);
            u32 old = 0;
            if ((vmask >> r) & 1u) old = (u32)__hip_atomic_fetch_add(&s_cnt[w][dg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            packed[r] = dg | (old << 8);
        }
    } else
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const bool valid = (vmask >> r) & 1u;
        const u32 dg = digit(key[r]);
        u64 peers = 0;
        u32 old = 0;
        if (valid) __hip_atomic_fetch_or(&s_mask[dg], lanebit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            peers = s_mask[dg];
            old = s_cnt[w][dg];
        }
        __builtin_amdgcn_wave_barrier();
        const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(peers >> 32), __builtin_amdgcn_mbcnt_lo((u32)peers, 0u));
        if (valid && below == 0) {  // lowest peer lane: bump the (wave, digit) counter, clear the mask
            s_cnt[w][dg] = old + (u32)__popcll(peers);
            s_mask[dg] = 0ull;
        }
        __builtin_amdgcn_wave_barrier();
        packed[r] = dg | ((old + below) << 8);
    }
    __syncthreads();

    // ---- 2./3. per digit: count over the waves = this tile's AGGREGATE, published at once; then the
    // exclusive prefix over the digits (tile-local offsets) and over the waves
    u32 total = 0;
    if (tid < RS_BINS) {
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ++ww) total += s_cnt[ww][tid];
        if (tile != 0) st_store(my_status, ST_FLAG_AGG | st_tag | (u64)total);
    }
    u32 tile_total;
    const u32 tile_off = block_excl_scan_u32<RS_THREADS>(total, s_ws, tile_total);
    if (tid < RS_BINS) {
        u32 run = tile_off;
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ++ww) {
            const u32 c = s_cnt[ww][tid];
            s_cnt[ww][tid] = run;
            run += c;
        }
    }
    __syncthreads();

    // ---- 5. keys: scatter into tile-sorted order in LDS -----------------------------------------
    K* stage_k = reinterpret_cast<K*>(s_stage);
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        if ((vmask >> r) & 1u) {
            const u32 dg = packed[r] & 0xFFu;
            const u32 pos = s_cnt[w][dg] + (packed[r] >> 8);
            packed[r] = pos;  // keep the tile-local destination for the payload phase
            stage_k[pos] = key[r];
        }
    }

    // ---- 6. look-back: global base of every digit = bin_base + sum of the predecessors' counts ---
    if (tid < RS_BINS) {
        u64 excl = 0;
        if (tile == 0) {
            st_store(my_status, ST_FLAG_PRE | st_tag | (u64)total);
        } else {
            // LB_BATCH independent status loads per step (one fabric round trip per step instead of
            // per predecessor).  Tile 0 always holds a prefix, so the walk ends there at the latest.
#ifndef RS_LB_BATCH
#define RS_LB_BATCH 8
#endif
            constexpr int LB_BATCH = RS_LB_BATCH;
            size_t p = tile;  // tiles [p, tile) are already summed into excl
            u32 spins = 0;
            bool done = false;
            while (!done) {
                u64 sv[LB_BATCH];
#pragma unroll
                for (int k = 0; k < LB_BATCH; ++k) {
                    const size_t q = p > (size_t)k ? p - 1 - (size_t)k : 0;
                    sv[k] = st_load(status + q * RS_BINS + tid);
                }
                size_t consumed = 0;
#pragma unroll
                for (int k = 0; k < LB_BATCH; ++k) {
                    if (!done && consumed == (size_t)k) {
                        const u64 f = (sv[k] & ST_TAG_BITS) == st_tag ? sv[k] >> 62 : 0ull;
                        if (f == 2) { excl += sv[k] & ST_VALUE; done = true; }
                        else if (f == 1) { excl += sv[k] & ST_VALUE; consumed = (size_t)k + 1; }
                    }
                }
                if (!done) {
                    p -= consumed;
                    if (consumed == 0) {
                        if (++spins > LOOKBACK_SPIN_LIMIT) {  // never expected; keeps every wave finite
                            atomicOr(&sc->err, KERR_LOOKBACK_TIMEOUT);
                            done = true;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
            }
            st_store(my_status, ST_FLAG_PRE | st_tag | (excl + (u64)total));
        }
        s_glob[tid] = bin_base[tid] + excl - (u64)tile_off;
    }
    __syncthreads();

    // ---- 7. stream the tile out: consecutive threads write consecutive records of a digit run ----
    u32 out_dg[RS_ITEMS];
#pragma unroll
    for (int k = 0; k < RS_ITEMS; ++k) {
        const u32 li = (u32)k * RS_THREADS + (u32)tid;
        out_dg[k] = 0;
        if (li < tile_total) {  // the tile's existing records, now contiguous in LDS
            const K kk = stage_k[li];
            const u32 dg = digit(kk);
            out_dg[k] = dg;
            size_t g = (size_t)(s_glob[dg] + (u64)li);
            if (g < n_out) keys_out[g] = kk;  // always holds unless a look-back timed out
        }
    }
    if constexpr (HAS_VAL) {
        V val[RS_ITEMS];
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            const u32 li = wbase + (u32)r * 64u;
            if ((vmask >> r) & 1u) {
                val[r] = vals_in[base + li];
                if constexpr (sizeof(V) == 4) {
                    if (box_in) val[r] |= (V)box_to_packed(box_in[base + li]) << BOX_ID_BITS;  // first pass: pack
                }
            }
        }
        __syncthreads();  // everyone is done reading stage_k
        V* stage_v = reinterpret_cast<V*>(s_stage);
#pragma unroll
        for (int r = 0; r < RS_ITEMS; ++r) {
            if ((vmask >> r) & 1u) stage_v[packed[r]] = val[r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RS_ITEMS; ++k) {
            const u32 li = (u32)k * RS_THREADS + (u32)tid;
            if (li < tile_total) {
                size_t g = (size_t)(s_glob[out_dg[k]] + (u64)li);
                if (g < n_out) {
                    V v = stage_v[li];
                    if constexpr (sizeof(V) == 4) {
                        if (box_out) {  // last pass: unpack
                            box_out[g] = (uint8_t)box_from_packed((u32)v >> BOX_ID_BITS);
                            v &= (V)BOX_ID_MASK;
                        }
                    }
                    vals_out[g] = v;
                }
            }
        }
    }
}

// =================================================================================================
// first radix pass with line-aligned stores (radix_claim_kernel)
//
// What bounds a radix pass on this part is the SHAPE of its stores (tools/micro/scatter_bw.hip, MI355X): runs of whole,
// aligned 128-byte lines scattered at random are written at 5.3 TB/s (5.2 TB/s beside a read stream of the same volume) --
// the same runs shifted by 8 bytes at 2.0 TB/s (256-byte runs) to 3.1 TB/s (1 KB runs): every run then starts and ends
// with a partial line.  A one-sweep tile leaves 256 runs of ~32 records at whatever offset the previous tile's run ended;
// about half of all records sit in a partial line.
//
// The FIRST pass of a sort that a finishing kernel completes has no order to preserve among records of equal digit --
// neither inside a tile nor across tiles.  So:
//   * no chain at all: a tile CLAIMS its share of every bin from the bin's cursor with one returning atomic (16 cursors
//     per 64-byte line, the lines a few KB apart so that they spread over memory channels);
//   * it claims and writes WHOLE, ALIGNED LINES only (16 records): workgroups are persistent (one per CU, 1024 threads,
//     static tile assignment) and carry the records that do not fill a line -- at most 15 per bin -- in LDS to their next
//     tile.  Per bin the aligned interior of its output range is handed out line by line, upwards; what the workgroups
//     still carry at the end is claimed record by record from a second cursor that fills the ragged head of the range and
//     then the range from its END downwards -- lines and records add up to the bin's size, so the two fronts meet exactly.
//     Should the lines claimed for a bin exceed its interior (by at most ONE line over the whole launch), that line's
//     records are claimed from the second cursor as well;
//   * the next tile's keys are loaded into the (then dead) key registers as soon as the current tile sits in LDS, the
//     payload at the top of its tile.
// The order this pass leaves inside a bin depends on timing; the sort's result does not (the finish orders every tie).
//
// CELLS (template flag): the SECOND pass of a sort whose first, claiming pass went by the HIGH digit.  No order to preserve
// here either, if every high bin can be spread over its 256 cells (high digit, low digit = the finishing kernel's groups)
// without knowing their sizes -- they are laid out beforehand, with room to spare, from the sizes the PREVIOUS sort of the
// same kind left behind (launch_cell_layout; CellRoute in bp_kernels.h).  Differences to the plain pass: a workgroup takes a
// contiguous range of tiles and makes one sub-pass per high bin a tile touches (the input is sorted by the high digit: all
// but ~1 % of the tiles lie inside one bin); a cell's whole lines are claimed upwards from its start, the workgroup's carry
// leaves record by record downwards from the cell's end whenever the workgroup moves on to another high bin; what does not
// fit a cell is dropped and flagged (the caller sorts again the plain way); every range is clipped to the layout's total.
// =================================================================================================
#ifndef RS_CUR_STRIDE
#define RS_CUR_STRIDE 1040  // u32 words between two groups of 16 bin cursors (4160 bytes)
#endif
constexpr int RC_THREADS = 1024, RC_ITEMS = 8, RC_TILE = RC_THREADS * RC_ITEMS, RC_WAVES = RC_THREADS / 64;
constexpr int RC_LINE = 16;                                    // records per 128-byte line of keys
constexpr int RC_CARRY_MAX = RS_BINS * (RC_LINE - 1);          // records a workgroup can carry between tiles
constexpr int RC_STAGE = RC_TILE + RC_CARRY_MAX;               // a tile's records + the carried ones
constexpr int RC_OUT_ITERS = (RC_STAGE / 2 + RC_THREADS - 1) / RC_THREADS;  // output iterations: two records per thread
constexpr size_t SW_CURSOR_WORDS = 16 * (size_t)RS_CUR_STRIDE;  // one set of 256 cursors
__host__ __device__ __forceinline__ u32 sw_cursor_slot(u32 d) { return (d >> 4) * (u32)RS_CUR_STRIDE + (d & 15u); }
static_assert(RC_TILE == 8192, "the pass kernels share the tile size (holes, workspace)");

struct ClaimArgs {
    const u64* keys_in; u64* keys_out;
    const void* vals_in; void* vals_out;
    size_t n;               // extent of the input (tiles are cut from it)
    size_t n_out;           // records that exist (== n without holes)
    int shift;              // this pass's digit
    const u64* bin_base;    // [256] exclusive prefix of the digit counts
    u32* line_cur;          // [sw_cursor_slot] lines claimed per bin (zero at launch)
    u32* frag_cur;          // [sw_cursor_slot] fragment records claimed per bin (zero at launch)
    const uint8_t* box_in;
    const u32* blockfill;
    u32 n_tiles;
    // sizes known only on the device (a sort enqueued behind its producer, see SIZES_*): n / n_out / n_tiles above are upper
    // bounds that sized the grid and the workspace
    const u32* dev_sizes;
    // CELLS mode (the second pass of a sort whose first, claiming pass went by the HIGH digit `hi_shift`): the records of high
    // bin h and digit d go to cell h * 256 + d, a range laid out beforehand with room to spare -- cell_base[c] (a multiple of
    // RC_LINE; keys_out is line-aligned) .. cell_base[c + 1]: whole lines are claimed upwards from its start
    // (cell_line_cur[c]), what is left over (a workgroup's carry when it leaves the high bin) record by record downwards from
    // its end (cell_frag_cur[c]).  A cell that would overflow drops the records and sets *miss.
    const u32* cell_base; u32* cell_line_cur; u32* cell_frag_cur; u32* miss;
    int hi_shift;
};

// Bin b's output range [B, E): the aligned interior starts at A and holds I whole lines, handed out upwards from A.  What
// does not leave in whole lines (a workgroup's final remainders, a line claimed beyond the interior) is claimed record by
// record: slot q of the bin -> the head fragment [B, A) first, then DOWNWARDS from E - 1.  Lines and records add up to
// the bin's size, so the two fronts meet exactly (records may end in interior lines no workgroup filled).
struct BinGeom {
    u32 B, E, A, I, head;
};
__device__ __forceinline__ BinGeom bin_geom(const u64* __restrict__ bin_base, u32 b, u32 n_out, u32 a0) {
    BinGeom g;
    g.B = (u32)bin_base[b];
    g.E = b + 1u < (u32)RS_BINS ? (u32)bin_base[b + 1u] : n_out;
    g.A = g.B + ((a0 - g.B) & (u32)(RC_LINE - 1));  // first position >= B whose address is line-aligned
    if (g.A < g.E) { g.I = (g.E - g.A) / RC_LINE; g.head = g.A - g.B; }
    else { g.A = g.E; g.I = 0; g.head = g.E - g.B; }
    return g;
}
__device__ __forceinline__ u32 frag_pos(const BinGeom& g, u32 q) { return q < g.head ? g.B + q : g.E - 1u - (q - g.head); }

#ifndef RC_PREFETCH_LATE
#define RC_PREFETCH_LATE 0          // 1: the next tile's keys are requested behind this tile's key stores instead of in front of them
#endif
#ifndef RC_OVERLAP
#define RC_OVERLAP 1                // the prefetch behind the owners' claims, the holes' masks at the top of the tile: nothing waits for the prefetch before the next tile
#endif
#ifndef RC_WAIT_BEFORE_PREFETCH
#define RC_WAIT_BEFORE_PREFETCH 1
#endif
#ifndef RC_CELLS_VALS_EARLY
#define RC_CELLS_VALS_EARLY 0          // (1: the CELLS pass requests its payload in front of the key stores; measured equal)
#endif
#ifndef RC_DRAIN_CELLS
#define RC_DRAIN_CELLS 0            // (CELLS: fetch's partial-tile path leaves loads pending into registers the store loop reuses)
#endif
#ifndef RC_DRAIN_BEFORE_STORES
#define RC_DRAIN_BEFORE_STORES (!RC_OVERLAP)
#endif
#ifndef SW_EXP
#define SW_EXP 0  // timing experiments (results are WRONG): 1 no key stores
#endif
// TAG: no effect on the code -- key-only passes over the tree's compact records (TAG = 1) get a kernel name of their own, so
// that a profile tells them from the pair sort's passes (same template otherwise, 2.6 times the records)
template <typename V, bool CELLS = false, int TAG = 0>
__global__ __launch_bounds__(RC_THREADS) void radix_claim_kernel(ClaimArgs a) {
    typedef u64 K;
    constexpr int T = RC_THREADS, ITEMS = RC_ITEMS, TILE = RC_TILE;
    constexpr bool HAS_VAL = StageBytes<V>::value != 0;
    static_assert((64 * ITEMS) % HOLE_BLOCK == 0 && TILE % HOLE_BLOCK == 0, "waves must not straddle blocks");
    constexpr int WAVE_BLOCKS = 64 * ITEMS / (int)HOLE_BLOCK;
    __shared__ __attribute__((aligned(16))) K s_stage[RC_STAGE];
    __shared__ __attribute__((aligned(16))) K s_carry[RS_BINS][RC_LINE];
    __shared__ u32 s_carry_v[HAS_VAL ? RS_BINS : 1][RC_LINE];
    __shared__ uint2 s_pa_tb[RS_BINS];  // .x: stage slot of a bin's first NEW record when it leaves with a line, .y: ... when it stays (both minus the rank offset)
    __shared__ int s_pt[RS_BINS];       // new records of rank < s_pt leave with a line, the others stay (tail region of the stage)
    __shared__ u32 s_cnt[2][RS_BINS];  // a tile's digit counts; double-buffered: the other tile's copy is zeroed meanwhile (no barrier at the top of a tile)
    __shared__ u32 s_c0[RS_BINS], s_keep[RS_BINS], s_ko[RS_BINS], s_glob[RS_BINS], s_lim[RS_BINS];
    __shared__ u32 s_ws[RC_WAVES];
    __shared__ u32 s_over;
    __shared__ u32 s_hb[2];  // CELLS: high digit of the tile's first / last record

    const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
    const ShiftDigit digit{a.shift};
    const ShiftDigit digit_hi{a.hi_shift};
    const K* __restrict__ keys_in = a.keys_in;
    K* __restrict__ keys_out = a.keys_out;
    const V* __restrict__ vals_in = (const V*)a.vals_in;
    V* __restrict__ vals_out = (V*)a.vals_out;
    const size_t n = a.dev_sizes ? (size_t)(a.blockfill ? a.dev_sizes[SIZES_EXTENT] : a.dev_sizes[SIZES_N]) : a.n;
    const u32 n_out = a.dev_sizes ? a.dev_sizes[SIZES_N] : (u32)a.n_out;
    const u32 n_tiles = a.dev_sizes ? (u32)((n + (size_t)TILE - 1) / (size_t)TILE) : a.n_tiles;
    const u32 wbase = (u32)w * (64 * ITEMS) + (u32)lane;
    // position p of the output is line-aligned <=> p = a0 (mod 16)
    const u32 a0 = (u32)(0u - (u32)((size_t)keys_out / sizeof(K))) & (u32)(RC_LINE - 1);
    // (addresses are formed from copies the optimiser cannot see through: hoisted out of the tile loop as loop invariants,
    // the per-lane pointers and offsets of every phase cost more registers than the tile's records)
    auto opaque = [](u32 v) { asm volatile("" : "+v"(v)); return v; };

    // owner threads (one per bin): the bin's aligned interior (CELLS: set when the workgroup enters a high bin)
    u32 own_A = 0, own_I = 0;
    if (tid < RS_BINS) {
        if (!CELLS) {
            const BinGeom g = bin_geom(a.bin_base, (u32)tid, n_out, a0);
            own_A = g.A; own_I = g.I;
            s_lim[tid] = g.A + g.I * RC_LINE;
        }
        s_c0[tid] = 0u; s_keep[tid] = 0u; s_cnt[0][tid] = 0u; s_cnt[1][tid] = 0u;
    }
    if (tid == 0) s_over = 0u;
    u32 cur_hi = 0xFFFFFFFFu;  // CELLS: the high bin the carried records belong to

    K key[ITEMS];
    u32 vmask = 0;  // bit r: record r of this lane exists
    // RC_OVERLAP: the fills of the fetched tile's blocks stay in registers until the tile is processed (apply_holes): consumed inside
    // fetch, behind the key loads, they cost a wait for everything in flight -- the prefetch would not overlap anything
    u32 fill[WAVE_BLOCKS];
    bool holes_pending = false;
    auto fetch = [&](u32 tile) {  // load the keys of a tile (the previous tile's are dead)
        vmask = 0;
        holes_pending = false;
        if (tile >= n_tiles) return;
        const u32 wb = opaque(wbase);
        const size_t base = (size_t)tile * TILE;
        const u32 tile_n = (u32)((n - base) < (size_t)TILE ? (n - base) : (size_t)TILE);
        // block-structured input: the fills of the wave's blocks first (wave-uniform, and OLDER than the key loads, so that
        // waiting for them does not wait for the keys); the keys of a hole are read like any other slot (the slots exist)
        // and masked out afterwards
        if (a.blockfill) {
            const size_t blk0 = (size_t)uniform_u32((u32)((base + (size_t)w * (64 * ITEMS)) / HOLE_BLOCK));
            const size_t n_blk = (n + HOLE_BLOCK - 1) / HOLE_BLOCK;
#pragma unroll
            for (int q = 0; q < WAVE_BLOCKS; ++q) fill[q] = blk0 + q < n_blk ? a.blockfill[blk0 + q] : 0u;
        }
        const K* __restrict__ src = keys_in + base + wb;
        if (tile_n == (u32)TILE) {  // every tile but the last
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) key[r] = src[r * 64];
            vmask = (1u << ITEMS) - 1u;
        } else {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const bool valid = wb + (u32)r * 64u < tile_n;
                vmask |= valid ? (1u << r) : 0u;
                key[r] = valid ? src[r * 64] : (K)0;
            }
        }
        if (a.blockfill) {
            if (RC_OVERLAP) { holes_pending = true; return; }
            const u32 in_block = wb & (HOLE_BLOCK - 1u);  // (wb + r * 64) % HOLE_BLOCK = in_block + (r * 64) % HOLE_BLOCK
            u32 keep = 0;
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) keep |= (in_block + (u32)((r * 64) % (int)HOLE_BLOCK)) < fill[r * 64 / (int)HOLE_BLOCK] ? (1u << r) : 0u;
            vmask &= keep;
        }
    };
    auto apply_holes = [&]() {  // (workgroup-uniform condition)
        if (!holes_pending) return;
        holes_pending = false;
        const u32 in_block = opaque(wbase) & (HOLE_BLOCK - 1u);
        u32 keep = 0;
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) keep |= (in_block + (u32)((r * 64) % (int)HOLE_BLOCK)) < fill[r * 64 / (int)HOLE_BLOCK] ? (1u << r) : 0u;
        vmask &= keep;
    };
    // CELLS: a workgroup takes a CONTIGUOUS range of tiles (it stays inside a high bin for as long as possible: its carry
    // is flushed whenever it leaves one); otherwise the tiles are dealt round robin
    const u32 tiles_per = CELLS ? (n_tiles + (u32)gridDim.x - 1u) / (u32)gridDim.x : 1u;
    const u32 tile_step = CELLS ? 1u : (u32)gridDim.x;
    const u32 tile_end = CELLS ? min(n_tiles, ((u32)blockIdx.x + 1u) * tiles_per) : n_tiles;
    u32 tile = CELLS ? (u32)blockIdx.x * tiles_per : (u32)blockIdx.x;
    auto fetch_next = [&](u32 t) { if (t < tile_end) fetch(t); else vmask = 0; };
    fetch_next(tile);
    u32 own_keep = 0;  // owner: records of its bin the workgroup carries into the next tile
    __syncthreads();

    // what the workgroup still carries goes to the fragment slots of its bins (the end of the kernel; CELLS: also whenever
    // it moves on to another high bin)
    auto flush_carry = [&]() {
        __syncthreads();
        if (tid < RS_BINS) {
            s_keep[tid] = own_keep;
            if (own_keep) {
                if (CELLS) s_glob[tid] = atomicAdd(a.cell_frag_cur + (size_t)cur_hi * RS_BINS + (u32)tid, own_keep);
                else s_glob[tid] = atomicAdd(a.frag_cur + sw_cursor_slot((u32)tid), own_keep);
            }
        }
        __syncthreads();
        for (int slot = tid; slot < RS_BINS * RC_LINE; slot += T) {
            const u32 b = (u32)slot / RC_LINE, i = (u32)slot % RC_LINE;
            if (i < s_keep[b]) {
                if (CELLS) {
                    const size_t c = (size_t)cur_hi * RS_BINS + b;
                    const u32 lim = a.cell_base[CELL_COUNT], ce = min(a.cell_base[c + 1], lim), cb = min(a.cell_base[c], ce);
                    const u32 q = s_glob[b] + i;
                    if (q < ce - cb) {
                        keys_out[ce - 1u - q] = s_carry[b][i];
                        if constexpr (HAS_VAL) vals_out[ce - 1u - q] = (V)s_carry_v[b][i];
                    } else {
                        *a.miss = 1u;
                    }
                } else {
                    const BinGeom g = bin_geom(a.bin_base, b, n_out, a0);
                    const u32 pos = frag_pos(g, s_glob[b] + i);
                    keys_out[pos] = s_carry[b][i];
                    if constexpr (HAS_VAL) vals_out[pos] = (V)s_carry_v[b][i];
                }
            }
        }
        __syncthreads();
        own_keep = 0;
    };

    u32 it = 0;
    for (; tile < tile_end; tile += tile_step) {
      if (RC_OVERLAP) apply_holes();
      // CELLS: the tile's records are sorted by the high digit: one sub-pass per high bin it touches (all but ~1 % of the
      // tiles lie inside one)
      u32 hb_lo = 0, hb_hi = 0;
      if (CELLS) {
          const size_t tbase = (size_t)tile * TILE;
          const u32 tile_n = (u32)((n - tbase) < (size_t)TILE ? (n - tbase) : (size_t)TILE);
          if (tid == 0) s_hb[0] = digit_hi(key[0]);
#pragma unroll
          for (int r = 0; r < ITEMS; ++r)
              if (wbase + (u32)r * 64u == tile_n - 1u) s_hb[1] = digit_hi(key[r]);
          __syncthreads();
          hb_lo = s_hb[0]; hb_hi = s_hb[1];
      }
      for (u32 hb = hb_lo; hb <= hb_hi; ++hb, ++it) {
        const bool last_sub = hb == hb_hi;
        u32 cur_vmask = vmask;
        if (CELLS) {
            if (hb != cur_hi) {  // (workgroup-uniform) leave the high bin: flush, then the new bin's cells
                if (cur_hi != 0xFFFFFFFFu) flush_carry();
                cur_hi = hb;
                if (tid < RS_BINS) {
                    // (clipped to the layout's total -- 0 after a failed layout --: whatever the ranges say, nothing leaves the buffer)
                    const size_t c = (size_t)cur_hi * RS_BINS + (u32)tid;
                    const u32 lim = a.cell_base[CELL_COUNT], ce = min(a.cell_base[c + 1], lim);
                    own_A = min(a.cell_base[c], ce);
                    own_I = (ce - own_A) / RC_LINE;
                    s_lim[tid] = own_A + own_I * RC_LINE;
                }
            }
            u32 m = 0;
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) m |= digit_hi(key[r]) == hb ? (1u << r) : 0u;
            cur_vmask &= m;
        }
        const size_t base = (size_t)tile * TILE;
        u32* const cnt_cur = s_cnt[it & 1u];

        // payload of THIS tile: in flight under the ranking (CELLS: loaded when the keys have left -- the registers it would occupy
        // through the ranking do not exist: 41 spilled at 128 per thread; before the next tile's prefetch: 73)
        V val[HAS_VAL ? ITEMS : 1];
        auto load_vals = [&]() {
            if constexpr (HAS_VAL) {
                const V* __restrict__ vsrc = vals_in + base + opaque(wbase);
#pragma unroll
                for (int r = 0; r < ITEMS; ++r) {
                    if ((cur_vmask >> r) & 1u) val[r] = vsrc[r * 64];
                }
                if constexpr (sizeof(V) == 4) {
                    if (a.box_in) {  // pack the box codes into the payload
                        const uint8_t* __restrict__ bsrc = a.box_in + base + opaque(wbase);
#pragma unroll
                        for (int r = 0; r < ITEMS; ++r) {
                            if ((cur_vmask >> r) & 1u) val[r] |= (V)box_to_packed(bsrc[r * 64]) << BOX_ID_BITS;
                        }
                    }
                }
            }
        };
        if constexpr (!CELLS) load_vals();

        // ---- rank: one counter per digit for the whole tile, one returning LDS atomic per record ----------------
        u32 rk[ITEMS / 2];  // two 16-bit fields each: the record's rank among the tile's records of its digit, later its stage slot
#pragma unroll
        for (int q = 0; q < ITEMS / 2; ++q) rk[q] = 0u;
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const u32 dg = digit(key[r]);
            u32 old = 0;
            if ((cur_vmask >> r) & 1u) old = (u32)__hip_atomic_fetch_add(&cnt_cur[dg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            rk[r >> 1] |= old << ((r & 1) * 16);
        }
        __syncthreads();

        // ---- per bin: carried + new records -> whole lines that leave now (claimed from the bin's cursor) + the rest ----
        u32 c0 = 0, nl = 0, keep = 0, l0 = 0, packed = 0;
        if (tid < RS_BINS) {
            c0 = own_keep;  // what the previous tile left of this bin
            s_c0[tid] = c0;
            const u32 c = c0 + cnt_cur[tid];
            s_cnt[(it & 1u) ^ 1u][tid] = 0u;  // the next tile's counters (last touched a whole tile ago)
            nl = c / RC_LINE; keep = c % RC_LINE;
            packed = (nl << 16) | keep;
            if (nl) {  // consumed after the tile sits in LDS
                if (CELLS) l0 = atomicAdd(a.cell_line_cur + (size_t)cur_hi * RS_BINS + (u32)tid, nl);
                else l0 = atomicAdd(a.line_cur + sw_cursor_slot((u32)tid), nl);
            }
        }
        u32 tot;
        const u32 ex = block_excl_scan_u32<T>(packed, s_ws, tot);
        const u32 f_total = (tot >> 16) * RC_LINE;  // records leaving with this tile
        u32 foff = 0;
        if (tid < RS_BINS) {
            foff = (ex >> 16) * RC_LINE;
            const u32 koff = ex & 0xFFFFu;
            s_pa_tb[tid] = make_uint2(foff + c0, f_total + koff + c0 - nl * RC_LINE);
            s_pt[tid] = (int)(nl * RC_LINE) - (int)c0;
            s_keep[tid] = keep; s_ko[tid] = koff;
            own_keep = keep;
        }
        __syncthreads();

        // ---- stage: [lines of bin 0 | lines of bin 1 | ... | rest of bin 0 | rest of bin 1 | ...] ----------------------
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            if ((cur_vmask >> r) & 1u) {
                const u32 dg = digit(key[r]);
                const u32 rank = (rk[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu;
                const uint2 pt2 = s_pa_tb[dg];
                const u32 pos = ((int)rank < s_pt[dg] ? pt2.x : pt2.y) + rank;
                rk[r >> 1] = (rk[r >> 1] & ~(0xFFFFu << ((r & 1) * 16))) | (pos << ((r & 1) * 16));
                s_stage[pos] = key[r];
            }
        }
        // the records carried in: in front of their bin's new ones (oldest first)
#pragma unroll
        for (int slot = tid; slot < RS_BINS * RC_LINE; slot += T) {
            const u32 b = (u32)slot / RC_LINE, i = (u32)slot % RC_LINE, cb = s_c0[b];
            if (i < cb) {
                const uint2 pt2 = s_pa_tb[b];
                const u32 dest = (s_pt[b] + (int)cb >= RC_LINE ? pt2.x : pt2.y) - cb + i;
                s_stage[dest] = s_carry[b][i];
            }
        }
        // the next tile's keys: in flight under this tile's stores (CELLS: the keys are still needed by the tile's next sub-pass)
        if (!RC_OVERLAP && RC_PREFETCH_LATE == 0 && last_sub) fetch_next(tile + tile_step);

        // ---- the claimed lines' positions --------------------------------------------------------------------------
        bool over = false;  // owner: its claim reaches beyond the bin's interior (at most one line per bin and launch)
        if (tid < RS_BINS) {
            s_glob[tid] = own_A + l0 * RC_LINE - foff;
            over = nl != 0 && l0 + nl > own_I;
            if (over) s_over = 1u;
            if (CELLS && over) *a.miss = 1u;  // the cell was laid out too small: the lines beyond it are dropped
        }
        // RC_OVERLAP: the prefetch goes out BEHIND the owners' wait for their claims (a returning atomic, older than these loads: the
        // compiler cannot count across fetch's branches and waits for everything in flight, the prefetch included -- and the barrier
        // below then holds every other wave for it as well)
        // (the claim is issued in one conditional block and read in another: for the compiler a path exists on which it is still in
        // flight, and it would wait wherever the store loop reuses the register -- one wait HERE, where only the claims can be
        // outstanding and their owners have just waited for them, says that nothing is)
        if (RC_OVERLAP && RC_WAIT_BEFORE_PREFETCH) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        if (RC_OVERLAP && RC_PREFETCH_LATE == 0 && last_sub) fetch_next(tile + tile_step);
        __syncthreads();

        // ---- stream the lines out: 8 consecutive threads write one aligned 128-byte line, TWO records (16 bytes) per thread --
        // the phase is store-ISSUE bound (s_memtime stamps: 17 % of a tile's time with 8-byte stores), and a line's records are
        // consecutive in the stage, so a thread takes an even / odd pair of one line: half the store, LDS-read and base-lookup
        // instructions.  (A line starts at an even stage slot and at a 128-byte aligned address: both halves of the pair
        // belong to one bin and the 16-byte store is aligned.)
        const bool any_over = s_over != 0u;
        // Every iteration of the store loop is conditional, so the compiler cannot know inside one that an earlier one has waited for
        // the loads still in flight (the owners' claims, the prefetch) and waits AGAIN in each -- s_waitcnt vmcnt(0), which also waits
        // for the stores issued so far: the six stores of a thread left one round trip apart.  One unconditional wait in front of the
        // loop (what the first iteration did anyway) tells it that nothing is pending: the stores go out back to back.
        if (RC_DRAIN_BEFORE_STORES || (CELLS && RC_DRAIN_CELLS)) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        // (CELLS: the payload is requested here, in front of the key stores -- the keys have left the registers it takes; behind the
        // stores its round trip stood between them and the payload's own stores)
        if constexpr (CELLS && RC_CELLS_VALS_EARLY) load_vals();
        u32 odg[HAS_VAL ? (RC_OUT_ITERS + 3) / 4 : 1];  // the written pairs' digits, one byte each (payload phase)
        const u32 otid = opaque((u32)tid);
#pragma unroll
        for (int k = 0; k < RC_OUT_ITERS; ++k) {
            const u32 li = 2u * ((u32)k * T + otid);
            if constexpr (HAS_VAL) { if (k % 4 == 0) odg[k / 4] = 0u; }
            if (li < f_total) {
                const ulonglong2 kk = *reinterpret_cast<const ulonglong2*>(&s_stage[li]);
                const u32 dg = digit(kk.x);
                if constexpr (HAS_VAL) odg[k / 4] |= dg << (8 * (k % 4));
                const u32 g = s_glob[dg] + li;
                if (!any_over || g < s_lim[dg]) {
                    if (SW_EXP != 1) *reinterpret_cast<ulonglong2*>(keys_out + g) = kk;
                }
            }
            if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);
        }
        // a line beyond the interior: its records fill fragment slots instead (rare: the owner moves them alone)
        u32 over_q = 0, over_at = 0;
        if (!CELLS && over) {
            const BinGeom g = bin_geom(a.bin_base, (u32)tid, n_out, a0);
            const u32 first = l0 > own_I ? l0 : own_I;  // first line index beyond the interior
            over_at = foff + (first - l0) * RC_LINE;
            const u32 cnt = (l0 + nl - first) * RC_LINE;
            over_q = atomicAdd(a.frag_cur + sw_cursor_slot((u32)tid), cnt);
            for (u32 i = 0; i < cnt; ++i) keys_out[frag_pos(g, over_q + i)] = s_stage[over_at + i];
        }
        if (RC_PREFETCH_LATE == 1 && last_sub) fetch_next(tile + tile_step);
        if constexpr (CELLS && !RC_CELLS_VALS_EARLY) load_vals();
        // what stays: back into the carry slots
#pragma unroll
        for (int slot = tid; slot < RS_BINS * RC_LINE; slot += T) {
            const u32 b = (u32)slot / RC_LINE, i = (u32)slot % RC_LINE;
            if (i < s_keep[b]) s_carry[b][i] = s_stage[f_total + s_ko[b] + i];
        }
        if constexpr (HAS_VAL) {
            __syncthreads();  // everyone is done with the keys in the stage
            V* stage_v = reinterpret_cast<V*>(s_stage);
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                if ((cur_vmask >> r) & 1u) stage_v[(rk[r >> 1] >> ((r & 1) * 16)) & 0xFFFFu] = val[r];
            }
#pragma unroll
            for (int slot = tid; slot < RS_BINS * RC_LINE; slot += T) {
                const u32 b = (u32)slot / RC_LINE, i = (u32)slot % RC_LINE, cb = s_c0[b];
                if (i < cb) {
                    const uint2 pt2 = s_pa_tb[b];
                    const u32 dest = (s_pt[b] + (int)cb >= RC_LINE ? pt2.x : pt2.y) - cb + i;
                    stage_v[dest] = (V)s_carry_v[b][i];
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < RC_OUT_ITERS; ++k) {
                const u32 li = 2u * ((u32)k * T + otid);
                if (li < f_total) {
                    const u32 dg = (odg[k / 4] >> (8 * (k % 4))) & 0xFFu;
                    const u32 g = s_glob[dg] + li;
                    if (!any_over || g < s_lim[dg]) {
                        if constexpr (sizeof(V) == 4) *reinterpret_cast<uint2*>(vals_out + g) = *reinterpret_cast<const uint2*>(&stage_v[li]);
                        else { vals_out[g] = stage_v[li]; vals_out[g + 1] = stage_v[li + 1]; }
                    }
                }
                if (k % 4 == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if (!CELLS && over) {
                const BinGeom g = bin_geom(a.bin_base, (u32)tid, n_out, a0);
                const u32 first = l0 > own_I ? l0 : own_I;
                const u32 cnt = (l0 + nl - first) * RC_LINE;
                for (u32 i = 0; i < cnt; ++i) vals_out[frag_pos(g, over_q + i)] = stage_v[over_at + i];
            }
#pragma unroll
            for (int slot = tid; slot < RS_BINS * RC_LINE; slot += T) {
                const u32 b = (u32)slot / RC_LINE, i = (u32)slot % RC_LINE;
                if (i < s_keep[b]) s_carry_v[b][i] = (u32)stage_v[f_total + s_ko[b] + i];
            }
        }
        if (any_over) {  // (uniform) the flag is per tile
            __syncthreads();
            if (tid == 0) s_over = 0u;
        }
        if (CELLS && !last_sub) __syncthreads();  // the next sub-pass reuses the tile's LDS bookkeeping
      }
    }

    // ---- what the workgroup still carries: into the fragment slots of its bins ------------------------------------------
    if (!CELLS || cur_hi != 0xFFFFFFFFu) flush_carry();
}

// =================================================================================================
// overlap scan: closed form of scan_impl (src/layer.rs:550-573)
//
// In a tree sorted by (index, id) the stack when entry i arrives is exactly the not-skipped
// entries j < i whose cell contains cell i.  For every depth dd <= depth(i)
// that occurs in the tree, those entries are the run of keys equal to
//     T(dd) = (key_i & level_mask(dd)) | dd
// that lies before i; runs are visited deepest first, so each search range ends where the
// previous one began.  Emission order inside an entry (stack bottom -> top) is ascending j.
// =================================================================================================
// Ancestor runs of entry i, deepest level first.  The kernels built on this are bound by the number
// of DEPENDENT load round trips a wave makes (lanes diverge, so every lane's chain is serialised), so
// the search is written to issue independent loads together: all gallop probes of a level at once
// (distances 1, 3, 7, 15, 31, 63 back from the end of the remaining range), a small bracket loaded
// whole instead of bisected, and the run of equal keys read four entries per step.
// visit(lo, hi, dd) receives each non-empty run [lo, hi) of keys equal to the level's target.
template <int KIND, class Visit>
__device__ __forceinline__ void walk_ancestors(const u64* __restrict__ keys, u32 i, u64 key, u32 present, Visit&& visit) {
    const u32 d = key_depth<KIND>(key);
    u32 todo = present & (d >= 31u ? 0xFFFFFFFFu : ((2u << d) - 1u));
    u32 end = i;  // every entry in [end, i) is a visited ancestor or irrelevant; search [0, end)
    while (todo != 0 && end > 0) {
        const u32 dd = 31u - (u32)__clz(todo);
        todo &= ~(1u << dd);
        const u64 target = (key & level_mask<KIND>(dd)) | (u64)dd;
        u32 hi = end;  // first index in [0, end) whose key is > target
        if (dd != d) {
            constexpr int G = 6;
            u32 pp[G];
            u64 pv[G];
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const u32 dist = (2u << k) - 1u;
                pp[k] = end >= dist ? end - dist : 0u;
                pv[k] = keys[pp[k]];
            }
            u32 lb = 0, hb = pp[G - 1];
            bool found = false;
#pragma unroll
            for (int k = 0; k < G; ++k) {
                if (!found && pv[k] <= target) {
                    found = true;
                    lb = pp[k] + 1;
                    hb = k == 0 ? end : pp[k - 1];
                }
            }
            if (!found) {  // farther than 63 entries back: classic gallop from there
                u32 step = 64;
                for (;;) {
                    if (step > hb) { lb = 0; break; }
                    const u32 p = hb - step;
                    if (keys[p] > target) { hb = p; step <<= 1; }
                    else { lb = p + 1; break; }
                }
            }
            while (hb - lb > 8) {
                const u32 mid = lb + ((hb - lb) >> 1);
                if (keys[mid] > target) hb = mid;
                else lb = mid + 1;
            }
            u32 le = 0;  // sorted: the keys <= target form a prefix of the bracket
#pragma unroll
            for (u32 q = 0; q < 8; ++q) {
                const u64 kq = (lb + q) < hb ? keys[lb + q] : ~0ull;
                le += kq <= target ? 1u : 0u;
            }
            hi = lb + le;
        }
        u32 j = hi;
        for (;;) {
            u32 eq = 0;
            bool run = true;
#pragma unroll
            for (u32 q = 0; q < 4; ++q) {
                const u64 kq = j > q ? keys[j - 1 - q] : ~target;
                run = run && kq == target;
                eq += run ? 1u : 0u;
            }
            j -= eq;
            if (eq < 4) break;
        }
        if (j < hi) visit(j, hi, dd);
        end = j;
    }
}

// Exact duplicate suppression at the source.  The reference emits (a, b) once per containment
// combo (cell of a inside-or-equal a cell of b) and removes the repeats with sort + dedup
// (src/layer.rs:473-474, 516-517).  With ids distinct, the orientation of a pair is fixed (a =
// the deeper object, or the larger id at equal depth), every cell of a lies in at most one cell
// of b, and each object's cells form a box of <= 2 cells per axis described by its BOX code.
// So the combo with the minimal-key cell of a is unique, and only that entry emits the pair:
// entry (c of a) keeps ancestor (C of b) iff no cell c' of a with key(c') < key(c) lies in b's
// box.  Unknown boxes (BOX_UNKNOWN) never suppress on their side, which can only leave extra
// duplicates for the (then mandatory) final dedup pass -- never lose a pair.
//
// Everything the rule needs can be read off the low bits of the cell coordinates: a neighbour cell
// c +/- 1 falls into a different depth-(d - sh) cell iff the low sh bits of c are all ones / all
// zeros, and the most significant bit in which c +/- 1 differs from c is the position of c's lowest
// zero / lowest one.  So per axis only the counts of trailing zeros and trailing ones of the
// coordinate are kept, taken straight from the Morton code (no decode).
template <int KIND> struct BoxCtx {
    // scalars only (per-axis arrays in a long-lived struct end up in scratch memory)
    u32 depth;
    u32 earlier;   // non-zero <=> some cell of the box sorts before this one (else nothing is ever suppressed)
    u32 two;       // axes on which the box has two cells (3-bit mask)
    u32 needpat;   // per axis (2-bit field): the ancestor's code that keeps the OTHER cell inside
    u32 table;     // bit O (3-bit axis set "other cell is inside the ancestor's box"): suppressed
    u32 thr;       // per axis (8-bit field): the other cell leaves a depth-(depth - sh) ancestor cell iff field >= sh

    __device__ __forceinline__ void init(u64 key, u32 code) {
        constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB, DB = KT<KIND>::DB;
        constexpr u64 LANE = DIM == 3 ? 0x1249249249249249ull : 0x5555555555555555ull;
        depth = key_depth<KIND>(key);
        earlier = 0; two = 0; needpat = 0; table = 0; thr = 0;
        if (code == BOX_UNKNOWN) return;  // unknown box: behave as a single-cell object (never suppressed)
        const u64 md = depth == 0 ? 0ull : ((key >> DB) >> (DIM * (AB - depth)));  // DIM * depth Morton bits
        const u64 valid = depth == 0 ? 0ull : ((1ull << (DIM * depth)) - 1ull);
        // per axis: own code, and the position of the most significant bit in which the OTHER cell of
        // the box differs from this one = trailing ones (other is one step up) / zeros (one step down)
        const u32 t0 = code & 3u, t1 = (code >> 2) & 3u, t2 = (code >> 4) & 3u;
        u32 f0 = 0, f1 = 0, f2 = 0;
        {
            const u64 src = t0 == 1u ? ((~md) & LANE & valid) : (md & LANE);
            const u32 c = src ? (u32)__builtin_ctzll(src) : DIM * depth;
            f0 = DIM == 3 ? (c * 171u) >> 9 : c >> 1;
        }
        {
            const u64 src = t1 == 1u ? ((~(md >> 1)) & LANE & (valid >> 1)) : ((md >> 1) & LANE);
            const u32 c = src ? (u32)__builtin_ctzll(src) : DIM * depth;
            f1 = DIM == 3 ? (c * 171u) >> 9 : c >> 1;
        }
        if (DIM == 3) {
            const u64 src = t2 == 1u ? ((~(md >> 2)) & LANE & (valid >> 2)) : ((md >> 2) & LANE);
            const u32 c = src ? (u32)__builtin_ctzll(src) : DIM * depth;
            f2 = (c * 171u) >> 9;
        }
        two = (t0 ? 1u : 0u) | (t1 ? 2u : 0u) | (t2 ? 4u : 0u);
        needpat = (t0 == 1u ? 1u : 2u) | ((t1 == 1u ? 1u : 2u) << 2) | ((t2 == 1u ? 1u : 2u) << 4);
        thr = f0 | (f1 << 8) | (f2 << 16);
        // Morton position of the deciding bit per axis, and whether moving to the other cell goes DOWN
        const int p0 = (int)(f0 * DIM + 0u), p1 = (int)(f1 * DIM + 1u), p2 = (int)(f2 * DIM + 2u);
        const bool dn0 = t0 == 2u, dn1 = t1 == 2u, dn2 = t2 == 2u;
        // diffset bit D (set of axes that differ from this cell): that cell exists and sorts earlier
        u32 ds = 0;
        if (t0 && dn0) ds |= 1u << 1;
        if (t1 && dn1) ds |= 1u << 2;
        if (t2 && dn2) ds |= 1u << 4;
        if (t0 && t1 && (p1 > p0 ? dn1 : dn0)) ds |= 1u << 3;
        if (t0 && t2 && (p2 > p0 ? dn2 : dn0)) ds |= 1u << 5;
        if (t1 && t2 && (p2 > p1 ? dn2 : dn1)) ds |= 1u << 6;
        if (t0 && t1 && t2) {
            const bool d = (p2 > p1 && p2 > p0) ? dn2 : (p1 > p0 ? dn1 : dn0);
            if (d) ds |= 1u << 7;
        }
        earlier = ds;
        // table[O] = some earlier cell differs from this one only on axes in O (subset closure of ds)
        u32 tb = ds;
        tb |= (tb & 0x55u) << 1;
        tb |= (tb & 0x33u) << 2;
        tb |= (tb & 0x0Fu) << 4;
        table = tb;
    }
    // axes on which the other cell of the box leaves the depth-dd ancestor cell (per run)
    __device__ __forceinline__ u32 crossing(u32 dd) const {
        const u32 sh = depth - dd;
        return ((thr & 0xFFu) >= sh ? 1u : 0u) | (((thr >> 8) & 0xFFu) >= sh ? 2u : 0u) | (((thr >> 16) & 0xFFu) >= sh ? 4u : 0u);
    }
    // ancestor entry with box code `code` in a run whose crossing mask is cr: is the pair emitted elsewhere?
    __device__ __forceinline__ bool suppressed(u32 cr, u32 code) const {
        const u32 cb = code == BOX_UNKNOWN ? 0u : code;       // unknown: the ancestor's own cell only
        const u32 x = cb ^ needpat;
        const u32 m = ~(x | (x >> 1)) & 0x15u;                // field == 0 <=> the ancestor's box covers the other cell
        const u32 match = (m & 1u) | ((m >> 1) & 2u) | ((m >> 2) & 4u);
        const u32 inside = (~cr | match) & two;
        return ((table >> inside) & 1u) != 0;
    }
};

// candidates of one run, highest index first, FOUR per step so their id / box loads are independent
template <class F> __device__ __forceinline__ void for_run_desc(u32 lo, u32 hi, F&& f) {
    u32 j = hi;
    while (j >= lo + 4) {
        j -= 4;
        f(j + 3); f(j + 2); f(j + 1); f(j);
    }
    while (j > lo) { --j; f(j); }
}

// general mode only: skipped(i) <=> some ancestor carries the same id (src/layer.rs:563-565)
template <int KIND>
__global__ __launch_bounds__(SCAN_THREADS) void scan_skipflags_kernel(ScanParams p) {
    const size_t gi = (size_t)blockIdx.x * SCAN_THREADS + threadIdx.x;
    if (gi >= p.m) return;
    const u32 i = (u32)gi;
    const u64 key = p.keys[i];
    const u32 id = p.ids[i];
    bool sk = false;
    walk_ancestors<KIND>(p.keys, i, key, p.depth_mask, [&](u32 lo, u32 hi, u32) {
        for_run_desc(lo, hi, [&](u32 j) { sk = sk || (p.ids[j] == id); });
    });
    p.skip[i] = sk ? 1 : 0;
}

template <int KIND, bool GENERAL>
__global__ __launch_bounds__(SCAN_THREADS) void scan_count_kernel(ScanParams p, u32* __restrict__ counts,
                                                                  u64* __restrict__ block_sums) {
    __shared__ u32 s_sum[SCAN_THREADS / 64];
    const size_t gi = (size_t)blockIdx.x * SCAN_THREADS + threadIdx.x;
    const u32 i = (u32)gi;
    u32 cnt = 0;
    if (gi < p.m) {
        const u64 key = p.keys[i];
        bool active = key_depth<KIND>(key) >= p.coarse_depth || coarse_emits(p, p.ids[i]);
        if (GENERAL) active = active && !p.skip[i];
        if (active) {
            if (GENERAL) {
                walk_ancestors<KIND>(p.keys, i, key, p.depth_mask, [&](u32 lo, u32 hi, u32) {
                    for_run_desc(lo, hi, [&](u32 j) { cnt += p.skip[j] ? 0u : 1u; });
                });
            } else if (p.box) {
                BoxCtx<KIND> bx;
                bx.init(key, p.box[i]);
                walk_ancestors<KIND>(p.keys, i, key, p.depth_mask, [&](u32 lo, u32 hi, u32 dd) {
                    if (bx.earlier == 0) { cnt += hi - lo; return; }
                    const u32 cr = bx.crossing(dd);
                    for_run_desc(lo, hi, [&](u32 j) { cnt += bx.suppressed(cr, p.box[j]) ? 0u : 1u; });
                });
            } else {
                walk_ancestors<KIND>(p.keys, i, key, p.depth_mask, [&](u32 lo, u32 hi, u32) { cnt += hi - lo; });
            }
        }
        counts[i] = cnt;
    }
    const u32 ws = wave_sum_u32(cnt);
    if (lane_id() == 0) s_sum[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 t = 0;
        for (int k = 0; k < SCAN_THREADS / 64; ++k) t += s_sum[k];
        block_sums[blockIdx.x] = t;
    }
}

template <int KIND, bool GENERAL>
__global__ __launch_bounds__(SCAN_THREADS) void scan_emit_kernel(ScanParams p, const u32* __restrict__ counts,
                                                                 const u64* __restrict__ block_offs,
                                                                 u64* __restrict__ out) {
    __shared__ u32 s_ws[SCAN_THREADS / 64];
    const size_t gi = (size_t)blockIdx.x * SCAN_THREADS + threadIdx.x;
    const u32 i = (u32)gi;
    const u32 cnt = gi < p.m ? counts[i] : 0u;
    u32 total;
    const u32 excl = block_excl_scan_u32<SCAN_THREADS>(cnt, s_ws, total);
    if (cnt == 0) return;
    const u64 key = p.keys[i];
    const u64 hi_id = (u64)p.ids[i] << 32;
    // ancestors are visited deepest-first / descending j; filling the slot backwards leaves them in
    // ascending j = the reference's stack order (src/layer.rs:566-570)
    size_t pos = (size_t)block_offs[blockIdx.x] + excl + cnt;
    if (!GENERAL && p.box) {
        BoxCtx<KIND> bx;
        bx.init(key, p.box[i]);
        walk_ancestors<KIND>(p.keys, i, key, p.depth_mask, [&](u32 lo, u32 hi, u32 dd) {
            const u32 cr = bx.crossing(dd);
            for_run_desc(lo, hi, [&](u32 j) {
                const u32 idj = p.ids[j];
                if (bx.earlier != 0 && bx.suppressed(cr, p.box[j])) return;
                --pos;
                out[pos] = hi_id | (u64)idj;
            });
        });
        return;
    }
    walk_ancestors<KIND>(p.keys, i, key, p.depth_mask, [&](u32 lo, u32 hi, u32) {
        for_run_desc(lo, hi, [&](u32 j) {
            const u32 idj = p.ids[j];
            if (GENERAL && p.skip[j]) return;
            --pos;
            out[pos] = hi_id | (u64)idj;
        });
    });
}

// Fused single-pass scan (default path when the result is sorted afterwards anyway, so the order
// of the raw stream is free): every entry walks its ancestors ONCE; the workgroup's pairs are
// staged in LDS, its output range is reserved with one atomicAdd on a global cursor, and the stage
// is flushed with coalesced stores.  A workgroup whose pairs do not fit the stage reserves first
// and walks a second time writing straight to HBM (rare: deep power-law stacks).
constexpr u32 FUSED_STAGE = 3072;  // pairs per workgroup staged in LDS (24 KB)
// Bounds::overlaps (src/geom.rs:112-119): inclusive; written with negated comparisons, so NaN overlaps
template <int DIM> __device__ __forceinline__ bool boxes_overlap(const float* __restrict__ a, const float* __restrict__ b) {
    bool ov = true;
#pragma unroll
    for (int ax = 0; ax < DIM; ++ax) {
        if (a[ax] > b[DIM + ax] || a[DIM + ax] < b[ax]) ov = false;
    }
    return ov;
}
// The example's sphere narrow phase (examples/main.rs:466-474): rows are (centre, radius);
// `dist > r0 + r1` rejects, so NaN keeps.  dist = sqrt(x*x + y*y [+ z*z]) summed left to right
// (cgmath ^0.17 InnerSpace::magnitude, a dependency that is not in the tree), no contraction.
template <int DIM> __device__ __forceinline__ bool spheres_touch(const float* __restrict__ a, const float* __restrict__ b) {
    float d2 = 0.f;
#pragma unroll
    for (int ax = 0; ax < DIM; ++ax) {
        const float o = b[ax] - a[ax];
        d2 = ax == 0 ? o * o : d2 + o * o;
    }
    return !(__fsqrt_rn(d2) > a[DIM] + b[DIM]);
}
// one narrow-phase row per id: FILT_AABB rows of 2*DIM floats, FILT_SPHERE rows of DIM+1
template <int DIM> __device__ __forceinline__ u32 filt_stride(int shape) { return shape == FILT_SPHERE ? DIM + 1 : 2 * DIM; }
template <int DIM> __device__ __forceinline__ bool narrow_keep(const float* __restrict__ a, const float* __restrict__ b, int shape) {
    return shape == FILT_SPHERE ? spheres_touch<DIM>(a, b) : boxes_overlap<DIM>(a, b);
}

template <int KIND, bool FILT>
__global__ __launch_bounds__(SCAN_THREADS) void scan_fused_kernel(ScanParams p, u64* __restrict__ out, u64 out_cap,
                                                                  DevScalars* sc) {
    constexpr int FDIM = (int)KT<KIND>::DIM;
    __shared__ u64 s_pairs[FUSED_STAGE];
    __shared__ u32 s_cursor;
    __shared__ u64 s_base;
    if (threadIdx.x == 0) s_cursor = 0;
    __syncthreads();
    const size_t gi = (size_t)blockIdx.x * SCAN_THREADS + threadIdx.x;
    const u32 i = (u32)gi;
    u64 key = 0, hi_id = 0;
    bool active = false;
    BoxCtx<KIND> bx;
    bx.earlier = 0;
    if (gi < p.m) {
        key = p.keys[i];
        active = key_depth<KIND>(key) >= p.coarse_depth || coarse_emits(p, p.ids[i]);
        if (active) {
            hi_id = (u64)p.ids[i] << 32;
            if (p.box) bx.init(key, p.box[i]);
        }
    }
    float mybox[2 * FDIM];
    const u32 fstride = FILT ? filt_stride<FDIM>(p.filt_shape) : 0;
    if (FILT && active) {
#pragma unroll
        for (int q = 0; q < 2 * FDIM; ++q)
            if ((u32)q < fstride) mybox[q] = p.filt_bounds[(size_t)(u32)(hi_id >> 32) * fstride + q];
    }
    u32 mine = 0;
    if (active) {
        walk_ancestors<KIND>(p.keys, i, key, p.depth_mask, [&](u32 lo, u32 hi, u32 dd) {
            const u32 cr = bx.crossing(dd);
            for_run_desc(lo, hi, [&](u32 j) {
                const u32 idj = p.ids[j];
                if (bx.earlier != 0 && bx.suppressed(cr, p.box[j])) return;
                if (FILT && !narrow_keep<FDIM>(mybox, p.filt_bounds + (size_t)idj * fstride, p.filt_shape)) return;
                const u32 slot = atomicAdd(&s_cursor, 1u);
                if (slot < FUSED_STAGE) s_pairs[slot] = hi_id | (u64)idj;
                ++mine;
            });
        });
    }
    __syncthreads();
    const u32 total = s_cursor;
    if (threadIdx.x == 0) s_base = total ? atomicAdd(&sc->total, (u64)total) : 0ull;
    __syncthreads();
    const u64 base = s_base;
    if (total == 0 || base + total > out_cap) return;  // capacity is checked by the host afterwards
    if (total <= FUSED_STAGE) {
        for (u32 k = threadIdx.x; k < total; k += SCAN_THREADS) out[base + k] = s_pairs[k];
        return;
    }
    // overflow: second walk, direct stores; slots again from an LDS cursor
    if (threadIdx.x == 0) s_cursor = 0;
    __syncthreads();
    if (active && mine) {
        walk_ancestors<KIND>(p.keys, i, key, p.depth_mask, [&](u32 lo, u32 hi, u32 dd) {
            const u32 cr = bx.crossing(dd);
            for_run_desc(lo, hi, [&](u32 j) {
                if (bx.earlier != 0 && bx.suppressed(cr, p.box[j])) return;
                const u32 idj = p.ids[j];
                if (FILT && !narrow_keep<FDIM>(mybox, p.filt_bounds + (size_t)idj * fstride, p.filt_shape)) return;
                const u32 slot = atomicAdd(&s_cursor, 1u);
                out[base + slot] = hi_id | (u64)idj;
            });
        });
    }
}

// =================================================================================================
// Expanding scan (default): the same pair set as scan_fused, enumerated from the ANCESTOR's side.
//
// In the sorted tree the entries contained in the cell of entry j (and the later entries of the
// same cell) are exactly the contiguous range (j, end_j) with end_j = first index whose key is
// >= (key_j & level_mask(d_j)) + one cell span at depth d_j: a key in that interval has j's Morton
// prefix and cannot have a smaller depth (it would sort before j).  So no per-level searches are
// needed: one forward gallop per entry gives end_j, and the raw stream is all (i, j), j < i < end_j.
// The ranges of the 64 entries a wave owns are concatenated (prefix sum) and the candidates are
// dealt to the lanes one per step, so lanes stay busy whatever the mix of range lengths; a
// candidate costs one 8-byte load of entry i's {id, suppression context} word.  Kept pairs are
// compacted with a ballot into a per-wave LDS stage and flushed with one atomic per ~500 pairs.
// Ranges of SX_BIG or more candidates (large objects) are cut into slices on a work list and
// spread over the whole grid by scan_expand_big_kernel.
// =================================================================================================
// Per-entry word: id << 32 | ctx.  ctx packs what the candidate test needs, laid out so that the test is THREE
// full-rate VALU instructions on whole words (a subtraction, a mask, a compare):
//   byte a = 0, 1, 2 (one per axis):  bits 0..4  G_a: where this entry is the UPPER of its box's two cells on axis a, g_a = depth -
//                                                (trailing zeros of the cell's coordinate) -- the box's other (lower) cell on that axis
//                                                leaves the ancestor cell of depth dd  <=>  dd >= g_a, and g_a >= 1; else 0
//                                     bit  5     always 0 (the guard of the byte-wise subtraction below); byte 0's bit 5 is
//                                                CTX_SKIPPED: set (with G_0 = 0), the entry takes no part as the deeper side -- the skip
//                                                rule's mark in the repeated-id modes, and the coarse entries of a shard scan
//                                     bits 6..7  t_a = this entry's box code on the axis (0 one cell, 1 lower, 2 upper of two)
//   byte 3:  0
// Which combo emits a pair: the reference emits (a, b) once per containment combo (a cell of `a` inside a cell of `b`), and only the
// combo of the MINIMAL-KEY cell of `a` inside `b`'s box is kept here.  The cells of `a` inside `b`'s box are a product of per-axis
// choices, and the Morton order of two cells of one box is decided by the axis whose differing bit is most significant -- on which the
// upper cell is the larger: so the minimal cell takes the lower cell wherever it is inside, and entry i is that cell  <=>  on every axis
// where i is the upper of two, the lower cell is NOT inside b's box  <=>  it leaves j's cell (dd >= g_a) and j is not itself the upper of
// b's two cells there (else b's lower cell covers it).  (Rounds 1 - 4 indexed an 8-bit table per entry by the 3-bit set of axes; the table's
// subset closure reduces to this conjunction -- tests/test_ctx_rule.py checks the two rules against each other and against brute force.)
// The ancestor (owner) side supplies ONE word: DJ byte a = 0xE0 | (t_a(j) == 2 ? 0 : dd).  Then t = DJ - ctx has no borrow between bytes
// (0xE0 >= 0xBF, the largest context byte), the code bits and 0xC0 of DJ cancel out of bit 5, and bit 5 of byte a  <=>  dd' >= G_a:
// keep  <=>  (t & 0x202020) == 0x202020.  A skipped / inactive entry (byte 0 = 0x20 | code) always fails: 0xE0 + dd' - 0x20 has bit 5 clear.
// a * K for a below 2^24 as ONE instruction with the constant as an immediate.  (What this saves is instructions, not issue slots:
// tools/micro/alu_rate.hip measures v_mul_lo_u32, v_mul_hi_u32 and the 64-bit shifts at the same 4.2 cycles per wave instruction
// as a 32-bit shift on gfx950 -- none of them is a reduced-rate instruction on this part, as rounds 4 - 5 first assumed; the
// gains booked to "24-bit multiplies" came from the shorter sequences: no constant in a register, no fix-ups of a division.)
template <u32 K> __device__ __forceinline__ u32 mul24k(u32 a) {  // a * K, K a compile-time constant below 2^24 (an immediate, no register)
    static_assert(K < (1u << 24), "24-bit constant");
    u32 r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "i"(K), "v"(a));
    return r;
}
constexpr u32 CTX_GUARD = 0x00202020u;
constexpr u32 CTX_SKIPPED = 1u << 5;
constexpr u32 CTX_INACTIVE = CTX_SKIPPED;
// an owner word no context passes (the null record behind a chunk's owners): byte 0 cannot borrow, byte 1 = 0xDF - (64 t + G) has bit 5 clear
constexpr u32 CTX_DJ_NONE = 0x00E0DFFFu;
__host__ __device__ constexpr u32 ctx_codebits(u32 c27) {  // c27 = t_x + 3 t_y + 9 t_z
    return ((c27 % 3u) << 6) | (((c27 % 9u) / 3u) << 14) | ((c27 / 9u) << 22);
}
// G bytes from the per-axis trailing-zero counts f_a of the cell's coordinates: (depth - f_a) on the axes whose code is 2
__device__ __forceinline__ u32 ctx_g_bytes(u32 codebits, u32 d, u32 f0, u32 f1, u32 f2) {
    // (24-bit multiplies with immediate constants: one instruction each)
    const u32 up = mul24k<0x1Fu>((codebits >> 7) & 0x010101u);            // 0x1F in the bytes of the upper-of-two axes
    return (mul24k<0x010101u>(d) - (f0 | (f1 << 8) | (f2 << 16))) & up;    // (f_a <= d < 32: no borrow)
}
template <int KIND> __device__ __forceinline__ u32 make_entry_ctx(u64 key, u32 code, u32 inactive_below) {
    BoxCtx<KIND> bx;
    bx.init(key, code);
    const u32 d = bx.depth;
    // the code always travels: every entry also acts as an ancestor, whatever its own role as the deeper side
    const u32 cb = code == BOX_UNKNOWN ? 0u : code;
    const u32 codebits = ((cb & 3u) << 6) | (((cb >> 2) & 3u) << 14) | (((cb >> 4) & 3u) << 22);
    if (d < inactive_below) return CTX_INACTIVE | codebits;
    if (code == BOX_UNKNOWN) return 0u;  // nothing is ever suppressed
    return ctx_g_bytes(codebits, d, bx.thr & 0xFFu, (bx.thr >> 8) & 0xFFu, (bx.thr >> 16) & 0xFFu) | codebits;
}
// The same word without BoxCtx (the kernels that build context words per entry -- entry_ctx, the tree finishes -- are VALU-bound):
// the code fields come from a 27-entry table, the G bytes from the trailing zeros of the three coordinate lanes.
struct CtxLut { u32 e[32]; };
constexpr CtxLut make_ctx_lut() {
    CtxLut l{};
    for (u32 c = 0; c < 32; ++c) l.e[c] = c < 27u ? ctx_codebits(c) : 0u;
    return l;
}
__constant__ CtxLut g_ctx_lut = make_ctx_lut();
constexpr int CTX_LUT_WORDS = 32;
// every thread of the workgroup calls this before the first entry_ctx_lut (ends with a barrier)
__device__ __forceinline__ void ctx_lut_load(u32* s_lut) {
    for (int k = (int)threadIdx.x; k < CTX_LUT_WORDS; k += (int)blockDim.x) s_lut[k] = g_ctx_lut.e[k];
    __syncthreads();
}
// c27: the box code as it travels in the ids (t_x + 3 t_y + 9 t_z, >= 27: unknown)
template <int KIND> __device__ __forceinline__ u32 entry_ctx_lut_wide(u64 key, u32 c27, u32 inactive_below, const u32* s_lut) {
    constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB, DB = KT<KIND>::DB;
    constexpr u64 LANE = DIM == 3 ? 0x1249249249249249ull : 0x5555555555555555ull;
    const u32 d = key_depth<KIND>(key);
    if (c27 >= 27u) return d < inactive_below ? CTX_INACTIVE : 0u;  // unknown box: own cell only, never suppressed
    const u32 codebits = s_lut[c27];
    if (d < inactive_below) return CTX_INACTIVE | codebits;
    const u64 md = d == 0 ? 0ull : ((key >> DB) >> (DIM * (AB - d)));  // DIM * depth Morton bits
    // per axis: trailing zeros of the coordinate (this is the upper cell: the other one is a step down); an empty lane: the whole coordinate
    const u64 s0 = md & LANE, s1 = (md >> 1) & LANE, s2 = DIM == 3 ? (md >> 2) & LANE : 0ull;
    const u32 c0 = s0 ? (u32)__builtin_ctzll(s0) : DIM * d, c1 = s1 ? (u32)__builtin_ctzll(s1) : DIM * d, c2 = s2 ? (u32)__builtin_ctzll(s2) : DIM * d;
    const u32 f0 = DIM == 3 ? mul24k<171u>(c0) >> 9 : c0 >> 1, f1 = DIM == 3 ? mul24k<171u>(c1) >> 9 : c1 >> 1, f2 = DIM == 3 ? mul24k<171u>(c2) >> 9 : 0u;
    return ctx_g_bytes(codebits, d, f0, f1, f2) | codebits;
}
// The same word where the entry's DIM * depth Morton bits fit 32 (depth <= 10 in 3-D, <= 16 in 2-D): the three axes' trailing
// runs come out of ONE 32-bit word -- per axis a mask, a find-first-bit and a minimum with DIM * depth (an empty lane).
template <int KIND> __device__ __forceinline__ u32 entry_ctx_lut_narrow(u64 key, u32 c27, u32 inactive_below, const u32* s_lut) {
    constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB, DB = KT<KIND>::DB;
    constexpr u32 L0 = DIM == 3 ? 0x49249249u : 0x55555555u, L1 = L0 << 1, L2 = DIM == 3 ? L0 << 2 : 0u;
    const u32 d = key_depth<KIND>(key);
    if (c27 >= 27u) return d < inactive_below ? CTX_INACTIVE : 0u;
    const u32 codebits = s_lut[c27];
    if (d < inactive_below) return CTX_INACTIVE | codebits;
    const u32 nb = DIM * d;                                      // <= 32 (the caller checks)
    const u32 md = (u32)(key >> (DB + DIM * (AB - d)));         // (depth 0: everything shifts out; nothing above bit nb - 1)
    const u32 s0 = md & L0, s1 = md & L1, s2 = md & L2;
    const u32 p0 = s0 ? (u32)__builtin_ctz(s0) : nb, p1 = s1 ? (u32)__builtin_ctz(s1) : nb,
              p2 = DIM == 3 ? s2 ? (u32)__builtin_ctz(s2) : nb : nb;
    const u32 f0 = DIM == 3 ? mul24k<171u>(p0) >> 9 : p0 >> 1, f1 = DIM == 3 ? mul24k<171u>(p1) >> 9 : p1 >> 1, f2 = DIM == 3 ? mul24k<171u>(p2) >> 9 : 0u;
    return ctx_g_bytes(codebits, d, f0, f1, f2) | codebits;
}
template <int KIND> __device__ __forceinline__ u32 entry_ctx_lut(u64 key, u32 c27, u32 inactive_below, const u32* s_lut) {
    constexpr u32 DIM = KT<KIND>::DIM;
    // (wave-uniform choice: one lane whose cell is deeper than 32 Morton bits sends the wave down the wide version)
    if (ballot64(DIM * key_depth<KIND>(key) > 32u) == 0ull) return entry_ctx_lut_narrow<KIND>(key, c27, inactive_below, s_lut);
    return entry_ctx_lut_wide<KIND>(key, c27, inactive_below, s_lut);
}
// owner-side word of an ancestor entry of depth dd whose own context word is ctx_j: 0xE0 | dd per byte, dd = 0 on the axes where j is
// the upper of its box's two cells (its lower cell covers a candidate's lower neighbour there: nothing passes, g_a >= 1)
__device__ __forceinline__ u32 ctx_owner_dj(u32 dd, u32 ctx_j) {
    const u32 up = mul24k<0x1Fu>((ctx_j >> 7) & 0x010101u);
    return (0x00E0E0E0u | mul24k<0x010101u>(dd)) & ~up;
}
// is candidate (entry i with context ctx_i, ancestor j described by dj) the emitting combo?
__device__ __forceinline__ bool ctx_keep(u32 ctx_i, u32 dj) { return ((dj - ctx_i) & CTX_GUARD) == CTX_GUARD; }

// active_lo / active_hi: entries of depth < inactive_below whose id lies in this range stay active all the same (shard
// scans: coarse-coarse pairs are emitted by the rank that owns the deeper entry's id); lo > hi: no such range
template <int KIND>
__global__ __launch_bounds__(256) void entry_ctx_kernel(const u64* __restrict__ keys, const u32* __restrict__ ids,
                                                        const uint8_t* __restrict__ box, size_t m, u32 inactive_below,
                                                        u64* __restrict__ idctx, u32 active_lo, u32 active_hi) {
    __shared__ u32 s_lut[CTX_LUT_WORDS];
    ctx_lut_load(s_lut);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const u32 code = box ? (u32)box[i] : (u32)BOX_UNKNOWN;
    const u32 id = ids[i];
    const u32 thr = (id >= active_lo && id <= active_hi) ? 0u : inactive_below;
    idctx[i] = ((u64)id << 32) | (u64)entry_ctx_lut<KIND>(keys[i], box_to_packed(code), thr, s_lut);
}

// SX_STAMPS (diagnostic builds only, tools/ab.sh build stamps "-DSX_STAMPS"): where a scanning wave's time goes, summed over the
// waves in s_memtime ticks -- read back with bp_debug_sx_stamps
#ifdef SX_STAMPS
__device__ unsigned long long g_sx_stamps[16];
__device__ unsigned long long g_sx_wave[8192][4];  // per wave of the last launch: loop ticks, steps << 32 | chunks, HW_ID | XCC_ID << 32, start tick
#if SX_STAMPS_LIGHT  // only the begin / end of a wave's chunk loop (the per-phase stamps slow the loop down)
#define SX_T(var) const u64 var = 0
#define SX_ACC(slot, v) sx_acc[slot] += (slot == 5 || slot == 6) ? (v) : 0
#else
#define SX_T(var) const u64 var = __builtin_readcyclecounter()
#define SX_ACC(slot, v) sx_acc[slot] += (v)
#endif
#else
#define SX_T(var)
#define SX_ACC(slot, v)
#endif
constexpr int SX_THREADS = 256;
constexpr int SX_WAVES = SX_THREADS / 64;
#ifndef SX_STAGE_DEF
#define SX_STAGE_DEF 512
#endif
constexpr u32 SX_STAGE = SX_STAGE_DEF;  // per-wave LDS stage: what is left of the previous chunk (< one block) + a chunk's pairs (a chunk that needs more drains in the middle)
#ifndef SX_BIG_DEF
#define SX_BIG_DEF 16384
#endif
constexpr u32 SX_BIG = SX_BIG_DEF;      // ranges at least this long go to the work list, in slices of this size
static_assert(SX_BLOCK == HOLE_BLOCK, "the first sort pass reads the scan's block structure");
constexpr u32 SX_CNT_UNKNOWN = 0x7FFu;  // range length field of a depth / count word: the producer could not bound the range

struct ExpandParams {
    const u64* keys;
    const u64* idctx;
    const unsigned short* dcnt;  // per entry: depth | range length << 5 (SX_CNT_UNKNOWN: not counted), or null (the range ends are searched)
    size_t m;
    u64* out;          // block-structured: block g = slots [g * SX_BLOCK, (g + 1) * SX_BLOCK)
    u64 out_blocks;    // capacity in blocks
    u32* blockfill;    // [out_blocks] zeroed by the host; records present in each block
    u64* shards;       // [SX_SHARDS] block cursors, [SX_SHARDS] pair counts, [SX_SHARDS] candidate counts (zeroed by the host)
    DevScalars* sc;
    u64* big_list;     // two words per item: j | slice << 32 (slice s covers candidates [s * SX_BIG, (s + 1) * SX_BIG) of j's range), the range's end
    u32 big_cap;
    u32 big_inline;    // no work-list kernel follows this launch: ranges up to SX_INLINE_MAX candidates stay with their wave, longer ones fail the scan
    u32 n_waves;       // waves of the main kernel's grid (chunk stride)
    const float* filt_bounds;
    int filt_shape;
    // digit histograms of the emitted pairs for the sort that follows (saves its separate read of the
    // stream): hist_passes <= SX_HIST_MAX rows of 256 u64 counters (zeroed by the host), row q counts
    // digit (pair >> hist_shift[q]) & 255
    u64* hist;
    int hist_passes;
    u32 hist_rows;     // bit q: row q is counted (a consumer that reads one digit only says so: every row costs LDS atomics per pair)
    int hist_shift[SX_HIST_MAX];
    // multi-GPU: instead, count the pairs per OWNER of the first id (owner_n >= 0: bounds[i] = first id of rank
    // i + 1, as PairOwnerDigit) into row 0 of hist -- the bucketing pass that follows then needs no histogram read
    int owner_n;
    u32 owner_bounds[15];
};

// first index > j whose key is >= upper (m if none); keys[j] < upper
__device__ __forceinline__ u32 range_end(const u64* __restrict__ keys, u32 m, u32 j, u64 upper) {
    constexpr int G = 7;
    u64 pv[G];
#pragma unroll
    for (int k = 0; k < G; ++k) {  // unconditional loads (clamped index) so that all probes are in flight together
        const u64 pos = (u64)j + (1u << k);
        const u64 v = keys[pos < m ? pos : (u64)m - 1u];
        pv[k] = pos < m ? v : ~0ull;
    }
    // every probe is consumed HERE: a probe left unread on some path stays "pending" for the compiler, which then
    // waits for ALL outstanding loads (the caller's cross-chunk prefetch included) wherever its register is reused
#pragma unroll
    for (int k = 0; k < G; ++k) {
        u32 a = (u32)pv[k], b = (u32)(pv[k] >> 32);
        asm volatile("" : "+v"(a), "+v"(b));
        pv[k] = ((u64)b << 32) | a;
    }
    u32 lo = j, hi = 0;
    bool found = false;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        if (!found) {
            if (pv[k] >= upper) { found = true; hi = (u32)min((u64)j + (1u << k), (u64)m); }
            else lo = j + (1u << k);
        }
    }
    if (!found) {
        u64 step = 128;
        for (;;) {
            const u64 pos = (u64)j + step;
            if (pos >= m) { hi = m; break; }
            if (keys[pos] >= upper) { hi = (u32)pos; break; }
            lo = (u32)pos;
            step <<= 1;
        }
    }
    while (hi - lo > 1) {  // keys[lo] < upper <= keys[hi] (hi == m: virtual +inf)
        const u32 mid = lo + ((hi - lo) >> 1);
        if (keys[mid] >= upper) hi = mid;
        else lo = mid;
    }
    return hi;
}

// The same for ONE range, by the whole wave (every lane calls it with the same arguments): first index > j whose key is >= upper
// (m if none); keys[j] < upper.  The next 64 keys, then a gallop round (lane k probes j + 4^k) and 64-ary rounds: 1 / ~5 dependent memory round trips
// for a range of a million entries, where the per-lane search above takes ~40 (and stalls the whole wave as long).
__device__ __forceinline__ u32 range_end_wave(const u64* __restrict__ keys, u32 m, u32 j, u64 upper) {
    const u32 lane = (u32)lane_id();
    {   // most ranges that leave the window end within the next 64 entries: one coalesced load decides those
        const u64 pos = (u64)j + 1u + lane;
        const bool in = pos < (u64)m;
        const u64 v = keys[in ? pos : (u64)m - 1u];
        const u64 ge = ballot64(!in || v >= upper);
        if (ge) return (u32)min((u64)j + 1u + (u64)__builtin_ctzll(ge), (u64)m);
        j += 64u;  // (keys[j] < upper still holds; j < m)
    }
    u64 lo = j, hi = m;  // keys[lo] < upper <= keys[hi] (hi == m: virtual +inf)
    {
        const u64 pos = (u64)j + (lane < 16u ? (1ull << (2u * lane)) : (1ull << 32));
        const bool in = pos < (u64)m;
        const u64 v = keys[in ? pos : (u64)m - 1u];
        const u64 ge = ballot64(!in || v >= upper) & 0xFFFFull;  // (lane 15 probes j + 2^30: beyond it only for m > 2^30)
        if (ge) {
            const int f = __builtin_ctzll(ge);
            hi = min((u64)j + (1ull << (2 * f)), (u64)m);
            if (f > 0) lo = (u64)j + (1ull << (2 * (f - 1)));
        } else {
            lo = (u64)j + (1ull << 30);
        }
    }
    while (hi - lo > 1) {  // wave-uniform
        const u64 span = hi - lo;
        // 64 probes strictly inside (lo, hi), non-decreasing in the lane (a short span: consecutive positions, the last ones clamped)
        const u64 pos = span <= 65 ? min(lo + 1 + lane, hi - 1) : lo + (span * (u64)(lane + 1)) / 65;
        const u64 v = keys[pos];
        const u64 ge = ballot64(v >= upper);
        const u32 plo = (u32)pos, phi = (u32)(pos >> 32);
        if (ge) {
            const int f = __builtin_ctzll(ge);
            hi = ((u64)(u32)__builtin_amdgcn_readlane((int)phi, f) << 32) | (u32)__builtin_amdgcn_readlane((int)plo, f);
            if (f > 0) lo = ((u64)(u32)__builtin_amdgcn_readlane((int)phi, f - 1) << 32) | (u32)__builtin_amdgcn_readlane((int)plo, f - 1);
        } else {
            lo = ((u64)(u32)__builtin_amdgcn_readlane((int)phi, 63) << 32) | (u32)__builtin_amdgcn_readlane((int)plo, 63);
        }
    }
    return (u32)hi;
}

template <int KIND> __device__ __forceinline__ u64 cell_upper(u64 key) {
    constexpr u32 DIM = KT<KIND>::DIM, TOP = KT<KIND>::DIM * KT<KIND>::AB + KT<KIND>::DB;
    const u32 d = key_depth<KIND>(key);
    if (d == 0) return ~0ull;  // the root cell holds everything
    return (key & level_mask<KIND>(d)) + (1ull << (TOP - DIM * d));
}

// Per-wave output stage.  One device-scope atomic on ONE address costs ~11 ns of serialised service
// (88 / us for the whole chip), which at one reservation per ~500 pairs was 3/4 of this kernel's
// time.  So the output is block-structured: a wave stages pairs in LDS (a linear buffer: pushes append, a full
// block leaves from the front and the few pairs behind it move down) and writes SX_BLOCK of them as one whole block; blocks are handed out by SX_SHARDS
// independent cursors (shard s owns blocks s, s + SX_SHARDS, ...), so the atomics spread over 64
// addresses.  Only a wave's last block is partial (blockfill says how much of it exists) and the
// shards end a few blocks apart; the first pass of the pair sort reads around those holes.
#ifndef SX_STATS_IN_VGPR
#define SX_STATS_IN_VGPR 1
#endif
#ifndef SX_EXP
#define SX_EXP 0  // timing experiments (results WRONG): 1 no pair stores, 2 no digit counts, 4 no stage writes, 8 no candidate steps
#endif
// Round 5: WHEN a block leaves.  vmcnt counts loads and stores alike and retires in order, so a wait for the next chunk's prefetched
// words also waits for every store issued behind those loads -- and a block written in the middle of a chunk left the chunk's
// back edge waiting for its stores' round trip (s_memtime stamps: 30 % of a wave's time at the back edge, more behind the
// returning atomic of a reservation).  Now the scanning wave only APPENDS while it works on a chunk; the full blocks leave at the
// top of the next chunk -- after the wait for that chunk's words, before the loads of the one after are issued -- so the stores
// a wait can see are a whole chunk old.  The reservation's atomic goes out when the last reserved block is taken and is read at
// the next drain.  (A chunk that would overflow the stage drains in the middle, with the old waits: rare by the stage's size.)
// LDS_STATS: the two statistics (pairs written, candidates examined) live in two LDS words per wave (stat), not in registers --
// the scanning kernel is short of both kinds
template <bool LDS_STATS = false>
struct WaveStageT {
    u64* buf;      // LDS, SX_STAGE slots
    u32 wp;        // pairs staged and not yet written (wave-uniform)
    u32 shard;
    u64 next_k;    // blocks reserved from the shard's cursor and not yet used: [next_k, next_k + k_left)
    u32 k_left;
    u32* hist;     // LDS, workgroup-wide: SX_HIST_MAX x 256 digit counters
    u64 examined;  // candidates this wave tested (wave-uniform)
    u64 total = 0; // pairs written so far
    u32 pend = 0;  // lane 0: what the reservation in flight returns (k_pend)
    u32 k_pend = 0;
    u64* stat = nullptr;  // LDS_STATS: [0] pairs written, [1] candidates examined
    // the lean block write of the scanning kernel (write_full): the output's capacity in blocks as 32 bits, and -- when exactly ONE
    // digit row is counted and it lies in the pairs' high dword (the cell route's) -- that row's shift and LDS offset
    u32 cap_blocks = 0;
    u32 fast_sh = 0, fast_row = 0, fast_on = 0;
    __device__ __forceinline__ void setup_lean(const ExpandParams& p) {
        cap_blocks = p.out_blocks > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)p.out_blocks;
        u32 n = 0, sh = 0, row = 0, hi = 0;
        for (int h = 0; h < SX_HIST_MAX; ++h)
            if (h < p.hist_passes && ((p.hist_rows >> h) & 1u)) { ++n; sh = (u32)p.hist_shift[h] & 31u; row = (u32)h; hi = p.hist_shift[h] >= 32 ? 1u : 0u; }
        // (wave-uniform, said explicitly: they live in scalar registers and the per-block branch on fast_on is a scalar branch)
        fast_on = (u32)__builtin_amdgcn_readfirstlane((int)((p.owner_n < 0 && n == 1u && hi == 1u) ? 1u : 0u));
        fast_sh = (u32)__builtin_amdgcn_readfirstlane((int)sh);
        fast_row = (u32)__builtin_amdgcn_readfirstlane((int)(row * 256u));
    }
    __device__ __forceinline__ void add_total(u32 n) {
        if (LDS_STATS) { if (lane_id() == 0) stat[0] += n; }
        else { total += n; if (SX_STATS_IN_VGPR) asm volatile("" : "+v"(total)); }
    }
    __device__ __forceinline__ void add_examined(u64 n) {
        if (LDS_STATS) { if (lane_id() == 0) stat[1] += n; }
        else { examined += n; if (SX_STATS_IN_VGPR) asm volatile("" : "+v"(examined)); }
    }
    __device__ __forceinline__ void reserve(const ExpandParams& p) {  // one returning atomic (~2 us round trip) buys SX_RESERVE blocks
        // (the address goes through a register the compiler cannot see through: with a provably uniform address LLVM's atomic
        // optimizer rewrites the add into mbcnt + atomic + readfirstlane -- and the readfirstlane WAITS for the atomic right here.
        // A shard's cursor counts blocks: the add is on its low dword -- below 2^32 blocks per shard, no carry -- and one register returns)
        u32 z = 0;
        asm volatile("" : "+v"(z));
        u32 k = 0;
        if (lane_id() == 0) k = atomicAdd(reinterpret_cast<u32*>(&p.shards[shard]) + z, (u32)SX_RESERVE);
        pend = k;
        k_pend = 1u;
    }
    // OWNER (multi-GPU): the pairs per owner of the first id are counted here, once per block -- a ballot per owner and
    // quarter block, one LDS add per owner (an LDS atomic per pushed pair on <= 16 counters serialised the wave's lanes
    // on a few addresses: the shard scan took 0.83 ms instead of 0.43)
    // from: the block's first slot in the stage.  AHEAD: the next reservation's atomic goes out when the last but one reserved block
    // is taken, and is read two blocks later -- by then it has long returned
    template <bool OWNER = false, bool AHEAD = false>
    __device__ __forceinline__ void write_block(const ExpandParams& p, u32 fill, u32 from = 0u) {
        __builtin_amdgcn_wave_barrier();
        if (k_left == 0) {
            if (!k_pend) reserve(p);
            next_k = (u64)(u32)__builtin_amdgcn_readfirstlane((int)pend);
            k_left = SX_RESERVE;
            k_pend = 0u;
        }
        const u64 g = next_k * SX_SHARDS + shard;
        ++next_k;
        --k_left;
        const bool fits = g < p.out_blocks;  // capacity is checked by the host afterwards (from the cursors)
        const u64* const src = buf + from;
        u64 pr[SX_BLOCK / 64];
#pragma unroll
        for (u32 q = 0; q < SX_BLOCK / 64; ++q) pr[q] = src[q * 64u + (u32)lane_id()];
#pragma unroll
        for (u32 q = 0; q < SX_BLOCK / 64; ++q)
            if (!(SX_EXP & 1) && q * 64u + (u32)lane_id() < fill && fits) p.out[g * SX_BLOCK + q * 64u + (u32)lane_id()] = pr[q];
        if (AHEAD && k_left == 1 && !k_pend) reserve(p);  // (a block ahead of its use)
        if (!OWNER) {
            // the digit counters for the sort that follows are fed here, 64 pairs per instruction (counted where a pair is
            // pushed, the same LDS adds ran on the ~45 % of the lanes whose candidate was kept, and cost twice the issue slots)
#pragma unroll
            for (int h = 0; h < SX_HIST_MAX; ++h) {
                if (!(SX_EXP & 2) && h < p.hist_passes && ((p.hist_rows >> h) & 1u)) {  // a counted digit never straddles the two dwords (scan_can_count)
                    const u32 sh = (u32)p.hist_shift[h] & 31u;
                    if (p.hist_shift[h] >= 32) {  // (a scalar branch: a select per pair otherwise)
#pragma unroll
                        for (u32 q = 0; q < SX_BLOCK / 64; ++q) {
                            const u32 dg = ((u32)(pr[q] >> 32) >> sh) & 255u;
                            if (q * 64u + (u32)lane_id() < fill) __hip_atomic_fetch_add(&hist[h * 256 + dg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    } else {
#pragma unroll
                        for (u32 q = 0; q < SX_BLOCK / 64; ++q) {
                            const u32 dg = ((u32)pr[q] >> sh) & 255u;
                            if (q * 64u + (u32)lane_id() < fill) __hip_atomic_fetch_add(&hist[h * 256 + dg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                }
            }
        } else {
#pragma unroll
            for (u32 q = 0; q < SX_BLOCK / 64; ++q) {
                const bool valid = q * 64u + (u32)lane_id() < fill;
                const u32 first = (u32)(pr[q] >> 32);
                u32 o = 0;
                for (int i = 0; i < p.owner_n; ++i) o += first >= p.owner_bounds[i] ? 1u : 0u;  // wave-uniform trip counts
                for (int r = 0; r <= p.owner_n; ++r) {
                    const u32 c = (u32)__popcll(ballot64(valid && o == (u32)r));
                    if (c && lane_id() == 0) __hip_atomic_fetch_add(&hist[r], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (fits && lane_id() == 0) p.blockfill[g] = fill;
        __builtin_amdgcn_wave_barrier();
    }
    // a FULL block, with as few instructions as it takes (the scanning kernel is bound by its instruction count, scalar and vector
    // alike; the general write_block spends ~60 scalar instructions and branches per block on fill masks, 64-bit block numbers and
    // four digit rows that are mostly off): no lane masks, 32-bit block numbers, the one counted digit row straight-line
    template <bool AHEAD>
    __device__ __forceinline__ void write_full(const ExpandParams& p, u32 from) {
        __builtin_amdgcn_wave_barrier();
        if (k_left == 0) {
            if (!k_pend) reserve(p);
            next_k = (u64)(u32)__builtin_amdgcn_readfirstlane((int)pend);
            k_left = SX_RESERVE;
            k_pend = 0u;
        }
        const u32 g = (u32)next_k * (u32)SX_SHARDS + shard;
        ++next_k;
        --k_left;
        const u64* const src = buf + from;
        u64 pr[SX_BLOCK / 64];
#pragma unroll
        for (u32 q = 0; q < SX_BLOCK / 64; ++q) pr[q] = src[q * 64u + (u32)lane_id()];
        if (g < cap_blocks) {  // capacity is checked by the host afterwards (from the cursors)
            u64* const dst = p.out + (size_t)g * SX_BLOCK;
#pragma unroll
            for (u32 q = 0; q < SX_BLOCK / 64; ++q)
                if (!(SX_EXP & 1)) dst[q * 64u + (u32)lane_id()] = pr[q];
            if (lane_id() == 0) p.blockfill[g] = SX_BLOCK;
        }
        if (AHEAD && k_left == 1 && !k_pend) reserve(p);  // (a block ahead of its use)
        if (fast_on) {
            if (!(SX_EXP & 2)) {
#pragma unroll
                for (u32 q = 0; q < SX_BLOCK / 64; ++q)
                    __hip_atomic_fetch_add(&hist[fast_row + (((u32)(pr[q] >> 32) >> fast_sh) & 255u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
#pragma unroll
            for (int h = 0; h < SX_HIST_MAX; ++h) {
                if (!(SX_EXP & 2) && h < p.hist_passes && ((p.hist_rows >> h) & 1u)) {  // a counted digit never straddles the two dwords (scan_can_count)
                    const u32 sh = (u32)p.hist_shift[h] & 31u;
                    const bool hi = p.hist_shift[h] >= 32;
#pragma unroll
                    for (u32 q = 0; q < SX_BLOCK / 64; ++q) {
                        const u32 dg = ((hi ? (u32)(pr[q] >> 32) : (u32)pr[q]) >> sh) & 255u;
                        __hip_atomic_fetch_add(&hist[h * 256 + dg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // every full block leaves; the (fewer than SX_BLOCK) pairs behind them move to the front
    template <bool OWNER, bool AHEAD = false>
    __device__ __forceinline__ void drain(const ExpandParams& p) {
        u32 from = 0;
        while (wp - from >= SX_BLOCK) {  // wave-uniform
            if constexpr (!OWNER && LDS_STATS) write_full<AHEAD>(p, from);   // (LDS_STATS: the scanning kernel, which called setup_lean)
            else write_block<OWNER, AHEAD>(p, SX_BLOCK, from);
            from += SX_BLOCK;
        }
        if (from == 0) return;
        add_total(from);
        const u32 over = wp - from;  // < SX_BLOCK
        wp = over;
        if (over) {
            const u32 l = (u32)lane_id();
            u64 a[SX_BLOCK / 64];
#pragma unroll
            for (u32 q = 0; q < SX_BLOCK / 64; ++q) a[q] = buf[from + q * 64u + l];  // (source and destination never overlap: from >= SX_BLOCK)
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (u32 q = 0; q < SX_BLOCK / 64; ++q)
                if (q * 64u + l < over) buf[q * 64u + l] = a[q];
            __builtin_amdgcn_wave_barrier();
        }
    }
    // one candidate per lane, written out as soon as a block is full (the work-list kernel, the region queries)
    template <bool OWNER = false>
    __device__ __forceinline__ void push(bool keep, u64 pair, const ExpandParams& p) {
        static_assert(SX_STAGE >= SX_BLOCK + 64, "stage too small for one candidate per lane");
        const u64 mask = ballot64(keep);
        const u32 n = (u32)__popcll(mask);
        if (n == 0) return;
        if (keep) buf[__builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, wp))] = pair;
        wp += n;
        if (wp >= SX_BLOCK) drain<OWNER>(p);
    }
    // two candidates per lane (<= 128 pairs per call), APPENDED: the caller drains at the top of its next chunk; only a stage
    // without room for another call drains here.  m0 / m1: the votes on keep0 / keep1; a pair is (first id = the candidate's,
    // second = its ancestor's), handed over as its two dwords: the stage's write position is folded into the (scalar) base, the
    // slot is v_mbcnt alone
    template <bool OWNER = false>
    __device__ __forceinline__ void push2(u64 m0, bool keep0, u32 lo0, u32 hi0, u64 m1, bool keep1, u32 lo1, u32 hi1, const ExpandParams& p) {
        static_assert(SX_STAGE >= SX_BLOCK + 128 && SX_STAGE >= 256, "stage too small for two candidates per lane");
        const u32 n0 = (u32)__popcll(m0), n = n0 + (u32)__popcll(m1);
        if (n == 0) return;
        u32* const d0 = reinterpret_cast<u32*>(buf + wp);
        u32* const d1 = reinterpret_cast<u32*>(buf + wp + n0);
        if (!(SX_EXP & 4) && keep0) {
            const u32 s = __builtin_amdgcn_mbcnt_hi((u32)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m0, 0u));
            d0[2u * s] = lo0; d0[2u * s + 1u] = hi0;
        }
        if (!(SX_EXP & 4) && keep1) {
            const u32 s = __builtin_amdgcn_mbcnt_hi((u32)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)m1, 0u));
            d1[2u * s] = lo1; d1[2u * s + 1u] = hi1;
        }
        wp += n;
        if (wp > SX_STAGE - 128u) drain<OWNER, true>(p);
    }
    template <bool OWNER = false>
    __device__ __forceinline__ void finish(const ExpandParams& p) {
        drain<OWNER>(p);
        if (wp) write_block<OWNER>(p, wp);
        add_total(wp);
        __builtin_amdgcn_wave_barrier();
        const u64 t = LDS_STATS ? uniform_u64(stat[0]) : total, x = LDS_STATS ? uniform_u64(stat[1]) : examined;
        if (t && lane_id() == 0) atomicAdd(&p.shards[SX_SHARDS + shard], t);
        if (x && lane_id() == 0) atomicAdd(&p.shards[2 * SX_SHARDS + shard], x);
    }
};
using WaveStage = WaveStageT<false>;

__host__ __device__ __forceinline__ int sx_hist_rows(const ExpandParams& p) { return p.owner_n >= 0 ? 1 : p.hist_passes; }
static size_t sx_hist_bytes(const ExpandParams& p) { return (size_t)sx_hist_rows(p) * 256 * sizeof(u32); }
__device__ __forceinline__ void sx_hist_clear(u32* s_hist, const ExpandParams& p) {
    for (int k = (int)threadIdx.x; k < sx_hist_rows(p) * 256; k += SX_THREADS) s_hist[k] = 0;
    __syncthreads();
}
__device__ __forceinline__ void sx_hist_flush(const u32* s_hist, const ExpandParams& p) {
    __syncthreads();
    for (int k = (int)threadIdx.x; k < sx_hist_rows(p) * 256; k += SX_THREADS) {
        const u32 c = s_hist[k];
        // digit counts: spread over HIST_COPIES copies (one address serves ~90 atomics per us: 1536 workgroups on ONE copy queue
        // for ~17 us at the end of the kernel); the per-owner counts of the shard scans stay in one row
        u64* const dst = p.owner_n >= 0 ? p.hist : p.hist + (size_t)(blockIdx.x % HIST_COPIES) * (SX_HIST_MAX * 256);
        if (c) atomicAdd(&dst[k], (u64)c);
    }
}

#ifndef SX_WAVES_PER_EU
#define SX_WAVES_PER_EU 6  // (round 5: the 512-pair stage leaves LDS for six workgroups per CU) round 4 measured at 10 M objects: 5 per SIMD 0.414 ms, 6 0.400, 7 0.390 (69 VGPRs, nothing spilled; round 3's 0.44 for 7 was with the stage cut by half a block)
#endif
#if SX_WAVES_PER_EU
// (the filtering variants need more registers than seven waves leave: they keep the range 5 .. 6.  OCC6: the same for the default
// variant, chosen at launch for mixed-scale scenes -- with ranges that leave the LDS windows, the seventh wave per SIMD costs the
// scan 2.5 % (12.5 M power-law objects: 1.15 against 1.12 ms), on the uniform scene it gains as much)
#ifndef SX_FILT_WAVES
#define SX_FILT_WAVES 3  // FILT: the window's rows (12 KB per workgroup in 3-D) and the queue leave LDS for three workgroups per CU (3, 4, 5 measured equal)
#endif
#define SX_OCC_ATTR __attribute__((amdgpu_waves_per_eu(FILT ? SX_FILT_WAVES : OCC6 ? 5 : SX_WAVES_PER_EU, FILT ? SX_FILT_WAVES + 1 : OCC6 ? 6 : SX_WAVES_PER_EU)))
#else
#define SX_OCC_ATTR
#endif
constexpr u32 SX_WIN = 128;  // entries of a chunk's LDS windows: the wave's 64 and the 64 that follow
constexpr u32 SX_INLINE_MAX = 4u * SX_BIG;  // longest range a wave walks itself when no work-list kernel follows (ExpandParams::big_inline)
static_assert(SX_WIN == SCAN_PAD_ENTRIES, "the padding the layer allocates behind the context / count words");
constexpr u32 SX_REC = 66;   // owner records per wave: 64 owners, the null record behind them, 16 bytes of padding
// GEN = 0: ids distinct, duplicates suppressed at the source (the default).
// GEN = 1: ids may repeat, MARK pass: every candidate whose id equals its ancestor's gets CTX_SKIPPED (nothing is emitted).
// GEN = 2: ids may repeat, EMIT pass: every combo of two unskipped entries (the reference's raw stream as a set).
// PRECNT: the tree finish left every entry's depth and the LENGTH of its range in p.dcnt (round 5): no key is loaded, nothing is
//         searched -- a chunk is two loads of {id, context} words and one of the 16-bit depth / count words.  Without it the range
//         ends are bisected in an LDS window of the chunk's keys (rounds 1 - 4; still what the other tree finishes' output takes).
//
// Everything a chunk waits for is either fetched one chunk ahead or served from LDS: (1) the chunk's global loads are issued a whole
// chunk before they are used (register prefetch across the persistent loop); (2) a candidate's {id, context} word comes from a
// 128-entry LDS window (nearly every range ends within the 64 entries after its owner) -- only a candidate beyond the window is a
// global load, and only chunks that have such a range run the loop that looks for them.
template <int KIND, bool FILT, bool OWNER, int GEN = 0, bool OCC6 = false, bool PRECNT = false>
__global__ __launch_bounds__(SX_THREADS) SX_OCC_ATTR void scan_expand_kernel(ExpandParams p) {
    constexpr int FDIM = (int)KT<KIND>::DIM;
#ifndef SX_SUPER_WORDS
#define SX_SUPER_WORDS 16  // (a chunk has ~320 candidates on the uniform scene; every byte of LDS counts towards the 7th wave per SIMD)
#endif
    constexpr u32 SUPER = 64 * SX_SUPER_WORDS;  // candidates covered by one fill of the boundary masks
    __shared__ u64 s_ends[SX_WAVES][SX_SUPER_WORDS];   // bit q of word b: some owner's range ends before candidate 64 b + q
    // owners with a non-empty range, compacted: {lane + 1 - first candidate (a candidate's entry offset is k + x), id, DJ, -}, and behind
    // them ONE null record {-total, -, CTX_DJ_NONE, -}: the lanes beyond a chunk's last candidate find it by the same count of ended
    // ranges as any other lane, land inside the window and fail the test -- no lane of a step needs an "is there a candidate" mask.
    // Without PRECNT the same storage holds the wave's key window while the range ends are searched (a union, so the compiler knows
    // the two views alias)
    union RecWin { uint4 rec[SX_REC]; u64 win[SX_WIN]; };
    __shared__ RecWin s_recwin[SX_WAVES];
    __shared__ u64 s_cwin[SX_WAVES][SX_WIN];  // {id, context} words of the entries [base, base + SX_WIN)
    // FILT: the narrow-phase rows of the window's entries -- gathered by id ONCE per chunk (128 gathers, a whole chunk ahead of their
    // use) instead of once per candidate that passes the cell test (~310 per chunk on the uniform scene, + 64 owners): a
    // candidate's row and its owner's are then LDS reads, like their {id, context} words
    __shared__ float s_rows[FILT ? SX_WAVES : 1][FILT ? SX_WIN : 1][2 * FDIM];
    // FILT: the candidates that passed the cell test wait here -- {window slot of the candidate | the owner's << 8} -- until 64 of them
    // can take the exact test together: a step's two slots pass it to ~29 % of their lanes, and the rows' LDS reads cost the same
    // whether one lane asks or all of them (the test inside the step put this kernel on the LDS pipe)
    constexpr u32 SX_QCAP = 192;  // (at most 63 left over + 128 new per step)
    __shared__ u32 s_q[FILT ? SX_WAVES : 1][FILT ? SX_QCAP : 1];
    [[maybe_unused]] u32 qn = 0;  // entries waiting (wave-uniform)
    __shared__ u64 s_stage[SX_WAVES][SX_STAGE];
    // the digit counters: as many rows as this launch counts (dynamic LDS, sx_hist_bytes) -- with the two rows of the usual
    // pair sort a workgroup needs 22.6 KB and seven of them share a CU
    extern __shared__ u32 s_hist[];
    sx_hist_clear(s_hist, p);
    // the wave's index and everything derived from it (chunk base, loop bounds, the stage's counters) are wave-uniform: said
    // explicitly, they live in SGPRs and the loops branch on scalar conditions instead of exec masks
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    uint4* const s_rec_w = s_recwin[w].rec;
    u64* const win = s_recwin[w].win;
    u64* const cwin = s_cwin[w];
    const u32 m = (u32)p.m;
    const u32 gw = blockIdx.x * SX_WAVES + (u32)w;  // persistent wave: chunks gw, gw + n_waves, ...
    __shared__ u64 s_stat[SX_WAVES][2];
    WaveStageT<true> st{s_stage[w], 0u, gw % SX_SHARDS, 0ull, 0u, s_hist, 0ull};
    st.stat = s_stat[w];
    st.setup_lean(p);
    if (lane < 2) s_stat[w][lane] = 0ull;
    // a chunk's global loads (clamped addresses: every load is unconditional, validity is decided at use)
    struct Pre { u64 k0, k1, w0, w1; u32 dc; };
    // (m < 2^32 - 1, checked by the host: the chunk loop's state is 32-bit -- scalar registers are what this kernel runs out of)
    auto prefetch = [&](u32 base) -> Pre {
        Pre r{};
        if constexpr (PRECNT) {
            // ONE shape of loads, whatever the chunk: both arrays are readable SX_WIN entries beyond the tree (the layer allocates them
            // so), and what lies there is replaced at use.  An unsigned 32-bit lane index on a uniform base: the loads take their base
            // from scalar registers -- no clamped indices, no 64-bit address pairs held in VGPRs
            // (the lane index goes through an opaque register: hoisted out of the loop, "array + lane" becomes a 64-bit address pair per
            // array that lives in VGPRs for the whole kernel -- and one of them was spilled to scratch, with its reload's wait in here)
            u32 ul = (u32)lane;
            asm volatile("" : "+v"(ul));
            const u64* __restrict__ c = p.idctx + base;
            r.w0 = c[ul]; r.w1 = c[64u + ul];
            r.dc = (p.dcnt + base)[ul];
        } else if (m - base >= SX_WIN) {  // wave-uniform: every chunk but the last one or two -- no clamping, scalar base + lane offset
            const u64* __restrict__ k = p.keys + base;
            const u64* __restrict__ c = p.idctx + base;
            r.k0 = k[lane]; r.k1 = k[64 + lane]; r.w0 = c[lane]; r.w1 = c[64 + lane];
        } else {
            const u32 last = m - 1u;
            const u32 a = min(base + (u32)lane, last), b = min(base + 64u + (u32)lane, last);  // (clamped: lanes beyond the tree are masked at use)
            r.k0 = p.keys[a];
            r.k1 = p.keys[b];
            r.w0 = p.idctx[a];
            r.w1 = p.idctx[b];
        }
        return r;
    };
    u32 base = gw * 64u;
    const u32 chunk_step = (u32)p.n_waves * 64u;
#ifdef SX_STAMPS
    u64 sx_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const u64 sx_begin = __builtin_readcyclecounter();
#endif
    Pre cur_pre{};
    if (base < m) cur_pre = prefetch(base);
    // FILT: the rows of a chunk's window entries are gathered a whole chunk ahead, like its words -- which therefore run TWO chunks ahead
    // (the gathers need the ids): nxt_pre = the words of the next chunk, nrow0 / nrow1 = the rows of this one
    [[maybe_unused]] Pre nxt_pre{};
    [[maybe_unused]] float nrow0[FILT ? 2 * FDIM : 1], nrow1[FILT ? 2 * FDIM : 1];
    auto gather_rows = [&](const Pre& pw, u32 b0, float* r0, float* r1) {
        if constexpr (FILT) {
            // (a slot behind the tree reads row 0; the id of a real entry is below the table's length: the host checks)
            const u32 jj = b0 + (u32)lane;
            const bool v0 = jj < m, v1 = v0 && m - jj > 64u;
            const u32 fs = filt_stride<FDIM>(p.filt_shape);
            const float* const r0p = p.filt_bounds + (size_t)(v0 ? (u32)(pw.w0 >> 32) : 0u) * fs;
            const float* const r1p = p.filt_bounds + (size_t)(v1 ? (u32)(pw.w1 >> 32) : 0u) * fs;
            if (p.filt_shape == FILT_SPHERE) {  // (wave-uniform)
#pragma unroll
                for (int k = 0; k < 2 * FDIM; ++k) { r0[k] = k <= FDIM ? r0p[k] : 0.f; r1[k] = k <= FDIM ? r1p[k] : 0.f; }
            } else {
#pragma unroll
                for (int k = 0; k < 2 * FDIM; ++k) { r0[k] = r0p[k]; r1[k] = r1p[k]; }
            }
        }
    };
    if constexpr (FILT) {
        if (base < m) {
            gather_rows(cur_pre, base, nrow0, nrow1);  // (the wave's first chunk: its words are waited for here, once)
            if (m - base > chunk_step) nxt_pre = prefetch(base + chunk_step);
        }
    }
    for (bool more = base < m; more; base += chunk_step) {  // wave-uniform (the last increment may wrap: `more` alone decides)
        SX_T(t0);
        const Pre pre = cur_pre;
        more = m - base > chunk_step;  // (base < m)
        if (GEN != 1) {
            // the blocks the LAST chunk filled leave here: behind the wait for this chunk's words, before the next chunk's loads are
            // issued (the compiler must not move either across: the loads are of read-only data and would otherwise be free to go up)
            if constexpr (PRECNT) {
                u32 a0 = (u32)pre.w0, a1 = (u32)pre.w1, a2 = pre.dc;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2) : : "memory");
            } else {
                u32 a0 = (u32)pre.w0, a1 = (u32)pre.w1, a3 = (u32)pre.k0, a4 = (u32)pre.k1;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a3), "+v"(a4) : : "memory");
            }
            st.template drain<OWNER, true>(p);
            asm volatile("" ::: "memory");
        }
        const u32 j = base + (u32)lane;
        const bool in0 = j < m;
        // FILT: this chunk's rows (gathered a chunk ago) go to the LDS window; then the next chunk's rows and the words of the chunk
        // behind it are requested
        if constexpr (FILT) {
#pragma unroll
            for (int k = 0; k < 2 * FDIM; ++k) { s_rows[w][lane][k] = nrow0[k]; s_rows[w][64 + lane][k] = nrow1[k]; }
            if (more) {
                cur_pre = nxt_pre;
                gather_rows(cur_pre, base + chunk_step, nrow0, nrow1);
                if (m - base - chunk_step > chunk_step) nxt_pre = prefetch(base + 2u * chunk_step);
            }
        } else {
            if (more) cur_pre = prefetch(base + chunk_step);  // the NEXT chunk's loads fly while this chunk is processed
        }
        if (PRECNT && m - base < SX_WIN) {  // (wave-uniform, the tree's last chunks) what was loaded from behind the tree: an inert word
            cwin[lane] = in0 ? pre.w0 : (u64)CTX_SKIPPED;
            cwin[64 + lane] = (in0 && m - j > 64u) ? pre.w1 : (u64)CTX_SKIPPED;
        } else {
            cwin[lane] = pre.w0;
            cwin[64 + lane] = pre.w1;
        }
        u32 c = 0, dd = 0;
        u32 end = 0;
        SX_T(t1);
        SX_ACC(0, t1 - t0);
        SX_ACC(5, 1);
        if constexpr (PRECNT) {
            dd = pre.dc & 31u;
            const u32 cc = pre.dc >> 5;
            // (<= m when the finish ran; clamped all the same: a tree sort whose deferred check fails leaves unwritten words behind, and the
            // scan that read them must stay inside the arrays -- its output is void, the host repairs the tree and scans again)
            end = min(j + 1u + cc, m);
            // a range the finish could not bound (an entry coarser than its group: never on the compact route): searched in HBM
            if (ballot64(in0 && cc == SX_CNT_UNKNOWN) != 0ull) {
                if (in0 && cc == SX_CNT_UNKNOWN) {
                    const u64 key = p.keys[j];
                    end = range_end(p.keys, m, j, cell_upper<KIND>(key));
                }
            }
        } else {
            const bool in1 = in0 && m - j > 64u;
            const u64 key = in0 ? pre.k0 : ~0ull;
            win[lane] = key;
            win[64 + lane] = in1 ? pre.k1 : ~0ull;
            __builtin_amdgcn_wave_barrier();
            // in the window: base + its slot there; SX_WIN + base: not in the window
            if (in0) {
                dd = key_depth<KIND>(key);
                const u64 upper = cell_upper<KIND>(key);
                u32 lo = (u32)lane, hi = SX_WIN;  // win[lo] < upper <= win[hi] (hi == SX_WIN: not in the window)
#pragma unroll
                for (int it = 0; it < 7; ++it) {
                    const u32 mid = (lo + hi) >> 1;  // <= 127: the read is unconditional (no branch per step)
                    const u64 v = win[mid];
                    const bool ge = (mid > lo) & (v >= upper);
                    hi = ge ? mid : hi;
                    lo = ge ? lo : mid;  // mid == lo leaves lo unchanged as well
                }
                end = (u32)base + hi;
            }
            // Ranges that leave the window (rare on a uniform scene, one in most chunks of a mixed-scale one): their end is somewhere
            // in HBM.  Up to four are searched by the whole wave, one after the other (~5 memory round trips each; the owner's key is
            // still in the key window); more: every lane for itself (~40 dependent loads each, but all lanes at once).
            if (m - base <= SX_WIN) {  // (wave-uniform) the window reaches the end of the tree
                if (in0 && end == (u32)base + SX_WIN) end = m;
            } else {
                u64 fm = ballot64(in0 && end == (u32)base + SX_WIN);
                if (fm != 0) {
                    if (__popcll(fm) <= 4) {
                        while (fm) {
                            const int l = __builtin_ctzll(fm);
                            fm &= fm - 1;
                            const u32 e = range_end_wave(p.keys, m, (u32)base + SX_WIN - 1u, cell_upper<KIND>(win[l]));
                            if (lane == l) end = e;
                        }
                    } else if (in0 && end == (u32)base + SX_WIN) {
                        end = range_end(p.keys, m, (u32)base + SX_WIN - 1u, cell_upper<KIND>(key));
                    }
                }
            }
        }
        if (in0) {
            c = end - j - 1u;
            if (GEN == 2 && ((u32)pre.w0 & CTX_SKIPPED)) c = 0;  // a skipped entry never enters the stack
            if (c >= SX_BIG) {  // the work list: one item per slice, each with the range's end (the second kernel searches nothing)
                const u32 ns = (c + SX_BIG - 1u) / SX_BIG;
                const u32 at = atomicAdd(&p.sc->big_count, ns);  // (counted either way: the host launches the work-list kernel when the last scan had such ranges)
                if (p.big_inline) {
                    // the last scan had no range this long, so no work-list kernel was enqueued behind this launch: a moderately
                    // long range stays in this wave's own stepping loop (a few hundred steps, once -- the next frame has the
                    // kernel again), a longer one fails the scan and the host runs the per-entry walk for this frame
                    if (c > SX_INLINE_MAX) { atomicOr(&p.sc->err, (u32)KERR_BIGLIST_OVERFLOW); c = 0; }
                } else
                if ((u64)at + ns <= p.big_cap) {
                    for (u32 sl = 0; sl < ns; ++sl) {
                        p.big_list[2 * (size_t)(at + sl)] = (u64)j | ((u64)sl << 32);
                        p.big_list[2 * (size_t)(at + sl) + 1] = (u64)end;
                    }
                } else {
                    atomicOr(&p.sc->err, (u32)KERR_BIGLIST_OVERFLOW);
                }
                if (!p.big_inline) c = 0;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the key window is dead from here on (s_rec is written below)
        const u64 nz = ballot64(c > 0);
        SX_T(t2);
        SX_ACC(1, t2 - t1);
        if (nz == 0) continue;  // wave-uniform
        const u32 incl = wave_incl_scan_u32(c);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        // The candidates of the 64 ranges, concatenated, are dealt to the lanes 128 per step.  Which
        // owner a candidate belongs to = how many ranges end before it: every owner sets ONE
        // bit (at its range end) in a row of 64-bit masks, and a step needs its two mask words, two
        // v_mbcnt pairs and one 16-byte LDS read per candidate -- no per-candidate search.
        st.add_examined(total);  // (a statistic: in LDS -- not worth two registers of either kind in this kernel)
        // a range leaves the word window: its last candidate is entry base + lane + c (wave-uniform: which stepping loop runs)
        const bool any_far = ballot64(c > 0 && (u32)lane + c >= SX_WIN) != 0ull;
        {
            const u32 n_own = (u32)__popcll(nz);
            if (c > 0)
                s_rec_w[__builtin_amdgcn_mbcnt_hi((u32)(nz >> 32), __builtin_amdgcn_mbcnt_lo((u32)nz, 0u))] =
                    make_uint4((u32)lane + 1u + c - incl, (u32)(pre.w0 >> 32), ctx_owner_dj(dd, (u32)pre.w0), (u32)lane);  // (.w: the owner's slot in the window -- FILT reads its row there)
            if (lane == 0) s_rec_w[n_own] = make_uint4(0u - total, 0u, CTX_DJ_NONE, 0u);
            if constexpr (FILT) __builtin_amdgcn_wave_barrier();  // (the rows written at the chunk's top)
        }
        u32 owners_done = 0;  // owners whose range ends before the current step (wave-uniform)
        SX_T(t3);
        SX_ACC(2, t3 - t2);
        for (u32 ss = 0; ss < total; ss += SUPER) {  // wave-uniform; one iteration unless the ranges are long
            if (lane < (int)SX_SUPER_WORDS) s_ends[w][lane] = 0ull;
            __builtin_amdgcn_wave_barrier();
            const u32 rel = incl - 1u - ss;  // this owner's LAST candidate is ss + rel (bit rel: an owner's range ends there)
            if (c > 0 && incl > ss && rel < SUPER)
                __hip_atomic_fetch_or(&s_ends[w][rel >> 6], 1ull << (rel & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_wave_barrier();
            // lane q keeps mask word q: a step takes its two words with v_readlane, no LDS round trip in the loop
            const u64 my_ends = s_ends[w][lane & (SX_SUPER_WORDS - 1)];
            const u32 ends_lo = (u32)my_ends, ends_hi = (u32)(my_ends >> 32);
            const u32 left = total - ss;
            const u32 n_steps = left >= SUPER ? SUPER / 128u : (left + 127u) >> 7;
            u32 kl = ss + (u32)lane;  // slot 0's candidate of the current step (slot 1: + 64)
            // One step = 128 candidates, two per lane (slot 0: candidate 128 b + lane, slot 1: + 64): the two lookup chains
            // (owner record -> {id, context} word -> test) are independent, so their LDS latencies overlap, and the
            // per-step bookkeeping is paid once per 128 candidates.
            // FILT: a candidate that passes the cell test is QUEUED (its window slot and its owner's); whenever 64 wait they take the exact
            // test together, rows from the LDS window -- a far candidate (the far loop only) is gathered from HBM and tested there and
            // then.  (Tried before, DESIGN section 9: rows gathered per candidate with the test a step or a group of steps behind its
            // loads -- bound by the throughput of 87 M random gathers; the test inside the step on LDS rows -- bound by the LDS pipe.)
            auto run_steps = [&](auto far_tag, auto nf_tag) {
                constexpr bool FAR = decltype(far_tag)::value;
                constexpr int NF = decltype(nf_tag)::value;  // floats per narrow-phase row (0: not FILT)
                // a far candidate (the far loop only): its row is gathered from HBM there and then
                [[maybe_unused]] auto narrow_far = [&](bool k, u32 cand_id, u32 own_slot) -> bool {
                    if constexpr (NF == 0) return k;
                    else {
                        bool r = false;
                        if (k) {
                            const float* const ow = &s_rows[FILT ? w : 0][own_slot][0];
                            const float* const g = p.filt_bounds + (size_t)cand_id * NF;
                            float a[NF], o[NF];
#pragma unroll
                            for (int q = 0; q < NF; ++q) { a[q] = g[q]; o[q] = ow[q]; }
                            if constexpr (NF == FDIM + 1) r = spheres_touch<FDIM>(a, o);
                            else r = boxes_overlap<FDIM>(a, o);
                        }
                        return r;
                    }
                };
                // the last `cnt` (<= 64) queue entries take the exact test, one per lane; what passes goes to the stage
                [[maybe_unused]] auto qpop = [&](u32 cnt) {
                    if constexpr (NF != 0) {
                        const u32* const qq = s_q[FILT ? w : 0];
                        qn -= cnt;
                        const bool on = (u32)lane < cnt;
                        const u32 e = on ? qq[qn + (u32)lane] : 0u;
                        const u32 cs = e & 0xFFu, os = e >> 8;
                        const float* const cw = &s_rows[FILT ? w : 0][cs][0];
                        const float* const ow = &s_rows[FILT ? w : 0][os][0];
                        float a[NF], o[NF];
#pragma unroll
                        for (int q = 0; q < NF; ++q) { a[q] = cw[q]; o[q] = ow[q]; }
                        const u32 cid = (u32)(cwin[cs] >> 32), oid = (u32)(cwin[os] >> 32);
                        bool r = on;
                        if constexpr (NF == FDIM + 1) r = r & spheres_touch<FDIM>(a, o);
                        else {  // boxes_overlap without its early outs (NaN compares false, so NaN overlaps: src/geom.rs:112-119)
#pragma unroll
                            for (int ax = 0; ax < FDIM; ++ax) r = r & !(a[ax] > o[FDIM + ax]) & !(a[FDIM + ax] < o[ax]);
                        }
                        st.template push2<OWNER>(ballot64(r), r, oid, cid, 0ull, false, 0u, 0u, p);
                    }
                };
                [[maybe_unused]] auto qpush = [&](bool k, u32 val) {
                    if constexpr (NF != 0) {
                        const u64 mk = ballot64(k);
                        if (k) s_q[FILT ? w : 0][qn + __builtin_amdgcn_mbcnt_hi((u32)(mk >> 32), __builtin_amdgcn_mbcnt_lo((u32)mk, 0u))] = val;
                        qn += (u32)__popcll(mk);
                    }
                };
                auto one = [&](u32 b) {
                    const u64 e0 = ((u64)(u32)__builtin_amdgcn_readlane((int)ends_hi, (int)(2u * b)) << 32) |
                                   (u64)(u32)__builtin_amdgcn_readlane((int)ends_lo, (int)(2u * b));
                    const u64 e1 = ((u64)(u32)__builtin_amdgcn_readlane((int)ends_hi, (int)(2u * b + 1u)) << 32) |
                                   (u64)(u32)__builtin_amdgcn_readlane((int)ends_lo, (int)(2u * b + 1u));
                    // owner of a candidate = how many ranges ended strictly before it: the bits below the lane (v_mbcnt) on top of
                    // the running total, which is folded into the (scalar) record pointer
                    const uint4* const rb0 = s_rec_w + owners_done;
                    const u32 n0 = (u32)__popcll(e0);
                    const uint4* const rb1 = rb0 + n0;
                    owners_done += n0 + (u32)__popcll(e1);
                    const uint4 r0 = rb0[__builtin_amdgcn_mbcnt_hi((u32)(e0 >> 32), __builtin_amdgcn_mbcnt_lo((u32)e0, 0u))];
                    const uint4 r1 = rb1[__builtin_amdgcn_mbcnt_hi((u32)(e1 >> 32), __builtin_amdgcn_mbcnt_lo((u32)e1, 0u))];
                    asm volatile("" : : "v"(r0.w), "v"(r1.w));  // (all four dwords: ds_read_b128 takes 4 LDS cycles, the ds_read_b96 of three takes 8)
                    const u32 off0 = kl + r0.x, off1 = kl + 64u + r1.x;   // candidate = entry base + off
                    const bool act0 = GEN == 1 && kl < total, act1 = GEN == 1 && kl + 64u < total;  // (only the marking pass asks)
                    kl += 128u;
                    u32 w0lo, w0hi, w1lo, w1hi;
                    if constexpr (FAR) {
                        // the common case must stay a ds_read: left to itself the compiler merges the LDS and the global load
                        // into ONE flat load of a selected address, whose wait (vmcnt(0)) also waits for the next chunk's prefetch
                        {
                            const u64 t0 = cwin[off0 < SX_WIN ? off0 : SX_WIN - 1u], t1 = cwin[off1 < SX_WIN ? off1 : SX_WIN - 1u];
                            w0lo = (u32)t0; w0hi = (u32)(t0 >> 32); w1lo = (u32)t1; w1hi = (u32)(t1 >> 32);
                            asm volatile("" : "+v"(w0lo), "+v"(w0hi), "+v"(w1lo), "+v"(w1hi));
                        }
                        const bool far0 = off0 >= SX_WIN, far1 = off1 >= SX_WIN;  // (a lane behind the last candidate is inside the window)
                        if (__any(far0 || far1)) {  // those lanes read HBM: both slots' loads first, ONE wait for the two
                            u64 t0 = 0, t1 = 0;
                            if (far0) t0 = p.idctx[base + off0];
                            if (far1) t1 = p.idctx[base + off1];
                            u32 a0 = (u32)t0, a1 = (u32)(t0 >> 32), b0 = (u32)t1, b1 = (u32)(t1 >> 32);
                            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));  // consumed HERE: else the wait lands at the join, on every path
                            if (far0) { w0lo = a0; w0hi = a1; }
                            if (far1) { w1lo = b0; w1hi = b1; }
                        }
                    } else {
                        const u64 t0 = cwin[off0], t1 = cwin[off1];
                        w0lo = (u32)t0; w0hi = (u32)(t0 >> 32); w1lo = (u32)t1; w1hi = (u32)(t1 >> 32);
                    }
                    if (GEN == 1) {  // mark: a candidate whose id equals its ancestor's is dropped by the skip rule
                        if (act0 && w0hi == r0.y && !(w0lo & CTX_SKIPPED)) atomicOr(const_cast<u64*>(p.idctx) + base + off0, (u64)CTX_SKIPPED);
                        if (act1 && w1hi == r1.y && !(w1lo & CTX_SKIPPED)) atomicOr(const_cast<u64*>(p.idctx) + base + off1, (u64)CTX_SKIPPED);
                        return;
                    }
                    // no short circuit: the test runs on every lane and the ballots come straight out of the compares
                    const bool c0 = ctx_keep(w0lo, r0.z), c1 = ctx_keep(w1lo, r1.z);
                    if (GEN == 0 && !FILT) {
                        st.template push2<OWNER>(ballot64(c0), c0, r0.y, w0hi, ballot64(c1), c1, r1.y, w1hi, p);
                        SX_ACC(6, 1);
                        return;
                    }
                    bool keep0 = c0, keep1 = c1;
                    if (GEN == 2) {  // (implied by the test: a skipped entry's context never passes; kept explicit)
                        keep0 = keep0 && !(w0lo & CTX_SKIPPED);
                        keep1 = keep1 && !(w1lo & CTX_SKIPPED);
                    }
                    if constexpr (NF != 0) {
                        bool far0 = false, far1 = false;
                        if constexpr (FAR) { far0 = off0 >= SX_WIN; far1 = off1 >= SX_WIN; }
                        qpush(keep0 & !far0, off0 | (r0.w << 8));
                        qpush(keep1 & !far1, off1 | (r1.w << 8));
                        while (qn >= 64u) qpop(64u);  // (wave-uniform)
                        if constexpr (FAR) {
                            keep0 = narrow_far(keep0 & far0, w0hi, r0.w);
                            keep1 = narrow_far(keep1 & far1, w1hi, r1.w);
                            if (__any(keep0 | keep1)) st.template push2<OWNER>(ballot64(keep0), keep0, r0.y, w0hi, ballot64(keep1), keep1, r1.y, w1hi, p);
                        }
                        SX_ACC(6, 1);
                        return;
                    }
                    SX_T(tp0);
                    st.template push2<OWNER>(ballot64(keep0), keep0, r0.y, w0hi, ballot64(keep1), keep1, r1.y, w1hi, p);
                    SX_T(tp1);
                    SX_ACC(4, tp1 - tp0);
                    SX_ACC(6, 1);
                };
                for (u32 b = 0; b < n_steps; ++b) one(b);
                if constexpr (NF != 0) {  // (the window these entries point into is overwritten at the next chunk's top)
                    if (qn) qpop(qn);
                }
            };
            if (SX_EXP & 8) continue;
            if constexpr (!FILT) {
                if (any_far) run_steps(std::true_type{}, std::integral_constant<int, 0>{});
                else run_steps(std::false_type{}, std::integral_constant<int, 0>{});
            } else if (p.filt_shape == FILT_SPHERE) {
                if (any_far) run_steps(std::true_type{}, std::integral_constant<int, FDIM + 1>{});
                else run_steps(std::false_type{}, std::integral_constant<int, FDIM + 1>{});
            } else {
                if (any_far) run_steps(std::true_type{}, std::integral_constant<int, 2 * FDIM>{});
                else run_steps(std::false_type{}, std::integral_constant<int, 2 * FDIM>{});
            }
        }
        __builtin_amdgcn_wave_barrier();
        SX_T(t4);
        SX_ACC(3, t4 - t3);
    }
#ifdef SX_STAMPS
    sx_acc[7] = __builtin_readcyclecounter() - sx_begin;
    if (lane == 0) {
        for (int q = 0; q < 8; ++q) atomicAdd(&g_sx_stamps[q], sx_acc[q]);
        atomicMax(&g_sx_stamps[8], sx_acc[7]);                                  // the slowest wave's chunk loop
        atomicMin(&g_sx_stamps[9], ~sx_acc[7]);                                 // (complement of) the fastest wave's
        atomicMax(&g_sx_stamps[10], sx_acc[5]);                                 // most chunks of one wave
        if (gw < 8192u) {
            g_sx_wave[gw][0] = sx_acc[7];
            g_sx_wave[gw][1] = (sx_acc[6] << 32) | sx_acc[5];
            g_sx_wave[gw][2] = (u64)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) | ((u64)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32);
            g_sx_wave[gw][3] = sx_begin;
        }
    }
#endif
    if (GEN == 1) return;
    st.template finish<OWNER>(p);
    sx_hist_flush(s_hist, p);
}

// work-list slices of the long ranges: one slice per workgroup iteration
template <int KIND, bool FILT, bool OWNER, int GEN = 0>
__global__ __launch_bounds__(SX_THREADS) void scan_expand_big_kernel(ExpandParams p) {
    constexpr int FDIM = (int)KT<KIND>::DIM;
    __shared__ u64 s_stage[SX_WAVES][SX_STAGE];
    __shared__ u32 s_hist[SX_HIST_MAX * 256];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 m = (u32)p.m;
    // an overflowed list has unwritten slots (a lane that did not fit reserved its slices anyway) and the host
    // discards this scan's output: do nothing
    const u32 n_items = p.sc->big_count <= p.big_cap ? p.sc->big_count : 0u;
    if (blockIdx.x >= n_items) return;  // workgroup-uniform: nothing on the list for this workgroup
    sx_hist_clear(s_hist, p);
    WaveStage st{s_stage[w], 0u, (p.n_waves + blockIdx.x * SX_WAVES + (u32)w) % SX_SHARDS, 0ull, 0u, s_hist, 0ull};
    for (u32 it = blockIdx.x; it < n_items; it += gridDim.x) {
        const u64 item = p.big_list[2 * (size_t)it];
        const u32 j = (u32)item, slice = (u32)(item >> 32);
        const u32 end = min((u32)p.big_list[2 * (size_t)it + 1], m);  // (found by the wave that listed the range)
        const u64 key = p.keys[j];
        const u64 wj = p.idctx[j];
        const u32 idj = (u32)(wj >> 32), dj = ctx_owner_dj(key_depth<KIND>(key), (u32)wj);
        const u64 first = (u64)j + 1u + (u64)slice * SX_BIG;
        const u64 last = min(first + SX_BIG, (u64)end);
        if (w == 0 && last > first) st.examined += last - first;
        // (four words per thread and step, all four loads issued before the first is used: with one load per step a slice was 64
        // dependent round trips per wave -- 2.2 TB/s on the 12.5 M power-law scene's 100 M listed candidates)
        constexpr u32 UNR = 4;
        for (u64 i0 = first; i0 < last; i0 += UNR * SX_THREADS) {  // workgroup-uniform trip count
            u64 wv[UNR];
#pragma unroll
            for (u32 q = 0; q < UNR; ++q) {
                const u64 i = i0 + q * SX_THREADS + threadIdx.x;
                wv[q] = i < last ? p.idctx[i] : (u64)CTX_SKIPPED;   // (a skipped word: fails the test, is never marked)
            }
#pragma unroll
            for (u32 q = 0; q < UNR; ++q) {
                const u64 i = i0 + q * SX_THREADS + threadIdx.x;
                const u64 wi = wv[q];
                const bool in = i < last;
                if (GEN == 1) {
                    if (in && (u32)(wi >> 32) == idj && !((u32)wi & CTX_SKIPPED)) atomicOr(const_cast<u64*>(p.idctx) + i, (u64)CTX_SKIPPED);
                    continue;
                }
                bool keep = in && ctx_keep((u32)wi, dj);
                if (GEN == 2) keep = keep && !((u32)wi & CTX_SKIPPED);
                if (FILT && keep)
                    keep = narrow_keep<FDIM>(p.filt_bounds + (size_t)(u32)(wi >> 32) * filt_stride<FDIM>(p.filt_shape),
                                             p.filt_bounds + (size_t)idj * filt_stride<FDIM>(p.filt_shape), p.filt_shape);
                if (i0 + q * SX_THREADS < last) st.template push<OWNER>(keep, (wi & 0xFFFFFFFF00000000ull) | (u64)idj, p);  // (workgroup-uniform)
            }
        }
    }
    if (GEN == 1) return;
    st.template finish<OWNER>(p);
    sx_hist_flush(s_hist, p);
}

// =================================================================================================
// dedup: Vec::dedup on the sorted pair list (src/layer.rs:474, 517)
// =================================================================================================
__global__ __launch_bounds__(UQ_THREADS) void unique_count_kernel(const u64* __restrict__ a, size_t n,
                                                                  u64* __restrict__ block_sums) {
    __shared__ u32 s_sum[UQ_THREADS / 64];
    const size_t base = (size_t)blockIdx.x * UQ_TILE;
    u32 cnt = 0;
#pragma unroll
    for (int k = 0; k < UQ_ITEMS; ++k) {
        const size_t i = base + (size_t)k * UQ_THREADS + threadIdx.x;
        if (i < n) cnt += (i == 0 || a[i] != a[i - 1]) ? 1u : 0u;
    }
    const u32 ws = wave_sum_u32(cnt);
    if (lane_id() == 0) s_sum[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 t = 0;
        for (int k = 0; k < UQ_THREADS / 64; ++k) t += s_sum[k];
        block_sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(UQ_THREADS) void unique_emit_kernel(const u64* __restrict__ a, size_t n,
                                                                 const u64* __restrict__ block_offs,
                                                                 uint2* __restrict__ out) {
    __shared__ u32 s_ws[UQ_THREADS / 64];
    const size_t base = (size_t)blockIdx.x * UQ_TILE;
    size_t run = (size_t)block_offs[blockIdx.x];
    // rounds of UQ_THREADS consecutive elements keep both the reads and the compacted writes coalesced
    for (int k = 0; k < UQ_ITEMS; ++k) {
        const size_t i = base + (size_t)k * UQ_THREADS + threadIdx.x;
        u64 v = 0;
        u32 keep = 0;
        if (i < n) {
            v = a[i];
            keep = (i == 0 || v != a[i - 1]) ? 1u : 0u;
        }
        u32 total;
        const u32 excl = block_excl_scan_u32<UQ_THREADS>(keep, s_ws, total);
        if (keep) out[run + excl] = make_uint2((u32)(v >> 32), (u32)v);  // memory order: id, id_
        run += total;
    }
}

// DUPS = false: the pairs are distinct; output in the ABI layout (id, id_).
// DUPS = true: pairs may repeat (no suppression at the source): equal pairs are ordered by input position and the
// output stays packed (id << 32 | id_), sorted, for the unique_count / unique_emit pass that follows.
template <bool DUPS>
__global__ __launch_bounds__(SEG_THREADS) void segment_finish_kernel(const u64* __restrict__ in, size_t n,
                                                                      uint2* __restrict__ out, DevScalars* sc) {
    constexpr int SEG_CHUNK = SEGP_CHUNK, SEG_HALO = SEGP_HALO;
    constexpr int WIN = SEG_HALO + SEG_CHUNK + SEG_HALO;
    constexpr int GROUPS = WIN / 64;
    __shared__ u64 s_win[WIN];
    __shared__ u64 s_heads[GROUPS];  // per 64 window elements: bit l set <=> element starts a segment
    const size_t base = (size_t)blockIdx.x * SEG_CHUNK;
    const size_t win_lo = base > (size_t)SEG_HALO ? base - SEG_HALO : 0;
    const size_t hi_want = base + SEG_CHUNK + SEG_HALO;
    const size_t win_hi = hi_want < n ? hi_want : n;
    const u32 wlen = (u32)(win_hi - win_lo);
    for (u32 k = threadIdx.x; k < (u32)WIN; k += SEG_THREADS) s_win[k] = k < wlen ? in[win_lo + k] : ~0ull;
    __syncthreads();
    // head flags by ballot: wave w covers groups w, w + 4, ...
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    for (int g = w; g < GROUPS; g += SEG_THREADS / 64) {
        const u32 x = (u32)g * 64u + (u32)lane;
        const bool head = x < wlen && (x == 0 || (u32)(s_win[x] >> 32) != (u32)(s_win[x - 1] >> 32));
        const u64 m = ballot64(head);
        if (lane == 0) s_heads[g] = m;
    }
    __syncthreads();
    const u32 off = (u32)(base - win_lo);
    const u32 cend = (u32)(((base + SEG_CHUNK) < n ? (base + SEG_CHUNK) : n) - win_lo);
    const u32 ngroups = (wlen + 63u) / 64u;
    bool too_long = false;
    for (u32 x = off + threadIdx.x; x < cend; x += SEG_THREADS) {
        const u64 v = s_win[x];
        const u32 g = x >> 6, l = x & 63u;
        const u64 m = s_heads[g];
        const u64 le = l == 63u ? ~0ull : ((2ull << l) - 1ull);
        // segment start: nearest head at or before x
        u32 start;
        const u64 below = m & le;
        if (below) {
            start = g * 64u + 63u - (u32)__clzll((long long)below);
        } else {
            int gg = (int)g - 1;
            while (gg >= 0 && s_heads[gg] == 0) --gg;
            start = gg < 0 ? 0u : (u32)gg * 64u + 63u - (u32)__clzll((long long)s_heads[gg]);
        }
        // segment end: nearest head after x
        u32 end;
        const u64 above = m & ~le;
        if (above) {
            end = g * 64u + (u32)__builtin_ctzll(above);
        } else {
            u32 gg = g + 1;
            while (gg < ngroups && s_heads[gg] == 0) ++gg;
            end = gg < ngroups ? gg * 64u + (u32)__builtin_ctzll(s_heads[gg]) : wlen;
        }
        const u32 vb = (u32)v;
        if ((start == 0 && win_lo > 0) || (end == wlen && win_hi < n)) {
            // the segment may leave the window (rare): bounds by bisection on the whole array, rank straight from HBM
            const u32 fid = (u32)(v >> 32);
            size_t lo = 0, hi = win_lo + x;  // first element with first id >= fid
            while (lo < hi) { const size_t mid = lo + ((hi - lo) >> 1); if ((u32)(in[mid] >> 32) < fid) lo = mid + 1; else hi = mid; }
            const size_t gs = lo;
            lo = win_lo + x + 1; hi = n;     // first element with first id > fid
            while (lo < hi) { const size_t mid = lo + ((hi - lo) >> 1); if ((u32)(in[mid] >> 32) <= fid) lo = mid + 1; else hi = mid; }
            const size_t ge = lo;
            if (ge - gs > (size_t)SEG_SLOW_MAX) { too_long = true; continue; }
            u32 rank = 0;
            const size_t self = win_lo + x;
            for (size_t y = gs; y < ge; ++y) {
                const u32 vy = (u32)in[y];
                rank += (vy < vb || (DUPS && vy == vb && y < self)) ? 1u : 0u;
            }
            if (DUPS) reinterpret_cast<u64*>(out)[gs + rank] = v; else out[gs + rank] = make_uint2(fid, vb);
            continue;
        }
        u32 rank = 0;
        const u32* s_lo = reinterpret_cast<const u32*>(s_win);  // low dword = second id (first ids are equal)
#pragma unroll 4  // segments hold ~9 pairs: the compiler's 16-wide unrolling mostly runs its remainder loops
        for (u32 y = start; y < end; ++y) {  // independent LDS reads
            const u32 vy = s_lo[2 * y];
            rank += (vy < vb || (DUPS && vy == vb && y < x)) ? 1u : 0u;
        }
        if (DUPS) reinterpret_cast<u64*>(out)[win_lo + start + rank] = v;
        else out[win_lo + start + rank] = make_uint2((u32)(v >> 32), (u32)v);
    }
    if (__any(too_long) && lane == 0) atomicOr(&sc->err, KERR_SEGMENT_TOO_LONG);
}

template <bool PACKED>
__global__ __launch_bounds__(SEG_THREADS) void tree_segment_finish_kernel(const u64* __restrict__ keys_in,
                                                                           const u32* __restrict__ ids_in, size_t n, int seg_shift,
                                                                           u64* __restrict__ keys_out, u32* __restrict__ ids_out,
                                                                           uint8_t* __restrict__ box_out, DevScalars* sc) {
    constexpr int SEG_CHUNK = SEGT_CHUNK, SEG_HALO = SEGT_HALO;
    constexpr int WIN = SEG_HALO + SEG_CHUNK + SEG_HALO;
    constexpr int GROUPS = WIN / 64;
    __shared__ u64 s_key[WIN];
    __shared__ u32 s_id[WIN];
    __shared__ u64 s_heads[GROUPS];
    const size_t base = (size_t)blockIdx.x * SEG_CHUNK;
    const size_t win_lo = base > (size_t)SEG_HALO ? base - SEG_HALO : 0;
    const size_t hi_want = base + SEG_CHUNK + SEG_HALO;
    const size_t win_hi = hi_want < n ? hi_want : n;
    const u32 wlen = (u32)(win_hi - win_lo);
    for (u32 k = threadIdx.x; k < (u32)WIN; k += SEG_THREADS) {
        s_key[k] = k < wlen ? keys_in[win_lo + k] : ~0ull;
        s_id[k] = k < wlen ? ids_in[win_lo + k] : 0u;
    }
    __syncthreads();
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    for (int g = w; g < GROUPS; g += SEG_THREADS / 64) {
        const u32 x = (u32)g * 64u + (u32)lane;
        const bool head = x < wlen && (x == 0 || (s_key[x] >> seg_shift) != (s_key[x - 1] >> seg_shift));
        const u64 m = ballot64(head);
        if (lane == 0) s_heads[g] = m;
    }
    __syncthreads();
    const u32 off = (u32)(base - win_lo);
    const u32 cend = (u32)(((base + SEG_CHUNK) < n ? (base + SEG_CHUNK) : n) - win_lo);
    const u32 ngroups = (wlen + 63u) / 64u;
    bool too_long = false;
    for (u32 x = off + threadIdx.x; x < cend; x += SEG_THREADS) {
        const u64 kx = s_key[x];
        const u32 px = s_id[x];
        const u32 ix = PACKED ? (px & BOX_ID_MASK) : px;
        const u32 g = x >> 6, l = x & 63u;
        const u64 m = s_heads[g];
        const u64 le = l == 63u ? ~0ull : ((2ull << l) - 1ull);
        u32 start;
        const u64 below = m & le;
        if (below) {
            start = g * 64u + 63u - (u32)__clzll((long long)below);
        } else {
            int gg = (int)g - 1;
            while (gg >= 0 && s_heads[gg] == 0) --gg;
            start = gg < 0 ? 0u : (u32)gg * 64u + 63u - (u32)__clzll((long long)s_heads[gg]);
        }
        u32 end;
        const u64 above = m & ~le;
        if (above) {
            end = g * 64u + (u32)__builtin_ctzll(above);
        } else {
            u32 gg = g + 1;
            while (gg < ngroups && s_heads[gg] == 0) ++gg;
            end = gg < ngroups ? gg * 64u + (u32)__builtin_ctzll(s_heads[gg]) : wlen;
        }
        if ((start == 0 && win_lo > 0) || (end == wlen && win_hi < n)) {
            // the group may leave the window (rare): bounds by bisection on the whole array, rank straight from HBM
            const u64 gk = kx >> seg_shift;
            const size_t self = win_lo + x;
            size_t lo = 0, hi = self;
            while (lo < hi) { const size_t mid = lo + ((hi - lo) >> 1); if ((keys_in[mid] >> seg_shift) < gk) lo = mid + 1; else hi = mid; }
            const size_t gs = lo;
            lo = self + 1; hi = n;
            while (lo < hi) { const size_t mid = lo + ((hi - lo) >> 1); if ((keys_in[mid] >> seg_shift) <= gk) lo = mid + 1; else hi = mid; }
            const size_t ge = lo;
            if (ge - gs > (size_t)SEG_SLOW_MAX) { too_long = true; continue; }
            u32 rank = 0;
            for (size_t y = gs; y < ge; ++y) {
                const u64 ky = keys_in[y];
                const u32 iy = PACKED ? (ids_in[y] & BOX_ID_MASK) : ids_in[y];
                const bool less = ky < kx || (ky == kx && (iy < ix || (iy == ix && y < self)));
                rank += less ? 1u : 0u;
            }
            const size_t dst = gs + rank;
            keys_out[dst] = kx;
            ids_out[dst] = ix;
            if (PACKED) box_out[dst] = (uint8_t)box_from_packed(px >> BOX_ID_BITS);
            continue;
        }
        u32 rank = 0;
        for (u32 y = start; y < end; ++y) {  // order: key, then id, then input position (exact duplicates)
            const u64 ky = s_key[y];
            const u32 iy = PACKED ? (s_id[y] & BOX_ID_MASK) : s_id[y];
            const bool less = ky < kx || (ky == kx && (iy < ix || (iy == ix && y < x)));
            rank += less ? 1u : 0u;
        }
        const size_t dst = win_lo + start + rank;
        keys_out[dst] = kx;
        ids_out[dst] = ix;
        if (PACKED) box_out[dst] = (uint8_t)box_from_packed(px >> BOX_ID_BITS);
    }
    if (__any(too_long) && lane == 0) atomicOr(&sc->err, KERR_SEGMENT_TOO_LONG);
}

// =================================================================================================
// Grouped finish of the TREE sort: the radix passes sort the top bits only (key >> gshift); the entries of one
// group (equal key >> gshift; TGRP_CAP fit) are contiguous and one workgroup orders them in LDS: counting sort
// by the next 8 key bits, rank by (key, id, input position) inside the bin, coalesced write-out with the box
// code split off the id payload.  A larger group takes several rounds over bin ranges that fit (re-reading the
// group, which is L2-resident).  tgroup_stats reduces the group sizes so that the host can choose between this
// path and the classic one (one more pass + tree_segment_finish) for the NEXT sort.
// =================================================================================================
#ifndef TGRP_THREADS_MACRO
#define TGRP_THREADS_MACRO 256
#endif
#ifndef TGRP_CAP_MACRO
#define TGRP_CAP_MACRO 2048
#endif
constexpr int TGRP_THREADS = TGRP_THREADS_MACRO, TGRP_CAP = TGRP_CAP_MACRO, TGRP_ITEMS = TGRP_CAP / TGRP_THREADS;
// one bin per thread, at most 256: many small workgroups in flight hide the per-group latency chain (bounds, load,
// three barriers) better than few large ones
constexpr int TGRP_BIN_BITS = TGRP_THREADS >= 256 ? 8 : (TGRP_THREADS == 128 ? 7 : 6), TGRP_BINS = 1 << TGRP_BIN_BITS;
static_assert(TGRP_THREADS >= TGRP_BINS && TGRP_CAP % TGRP_THREADS == 0, "tree group geometry");
constexpr u32 TGRP_SLOW_MAX = 32768;  // input positions travel as u16 + rounds stay few; longer groups: classic sort
// bounds[t] = first index whose (key >> gshift) is >= g0 + t, for t = 0 .. n_groups; and, in the same launch, the
// statistics of the group sizes bounds[t + 1] - bounds[t]: stats[0] = largest group, stats[1] = records in groups above
// `cap` (both zero beforehand: they live in the scalars the caller has just reset).
#ifndef GBS_ARY
#define GBS_ARY 4
#endif
// one WAVE per 63 groups (its 64 lanes find 64 consecutive bounds, the last one again by the next wave): at 10 M objects the
// 39 K bounds are 620 workgroups instead of 153 of 256 threads -- the searches are chains of dependent, uncoalesced loads, and
// what they need is every CU's load path, not threads per CU
__global__ __launch_bounds__(64) void group_bounds_stats_kernel(const u64* __restrict__ keys, u32 n, u32 n_groups, u64 g0, int gshift,
                                                                u32* __restrict__ bounds, u32 cap, u32* __restrict__ stats,
                                                                const u32* __restrict__ dev_sizes = nullptr) {
    if (dev_sizes) n = dev_sizes[SIZES_N];  // (n was an upper bound)
    const u32 t = blockIdx.x * 63u + threadIdx.x;
    u32 a = n;
    if (t <= n_groups) {
        // first index whose group is >= key.  GBS_ARY-ary: the probes of a step are independent loads (one memory latency
        // per step); measured at 1 M / 10 M objects: binary 9.0 / 18.5 us (256-thread workgroups), 16-ary 12 / -- (the probes
        // are uncoalesced: four times the requests)
        const u64 key = g0 + (u64)t;
        u32 b = n;
        a = 0;
        constexpr int P = GBS_ARY - 1;
        while (GBS_ARY > 2 && b - a > (u32)GBS_ARY) {
            const u32 span = b - a;
            u64 pv[P > 0 ? P : 1];
            u32 pp[P > 0 ? P : 1];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                pp[q] = a + (u32)(((u64)span * (u32)(q + 1)) / (u32)GBS_ARY);  // a < pp[q] < b, non-decreasing in q
                pv[q] = keys[pp[q]];
            }
            u32 na = a, nb = b;
#pragma unroll
            for (int q = 0; q < P; ++q) {
                if ((pv[q] >> gshift) < key) na = pp[q] + 1u;          // everything up to pp[q] is below the key
                else if (pp[q] < nb) nb = pp[q];                       // the first probe at or above the key bounds the answer
            }
            a = na; b = nb;
        }
        while (a < b) {
            const u32 mid = a + ((b - a) >> 1);
            if ((keys[mid] >> gshift) < key) a = mid + 1; else b = mid;
        }
        bounds[t] = a;
    }
    if (!stats) return;
    const u32 nxt = (u32)__shfl_down((int)a, 1, 64);
    u32 c = 0;
    if (threadIdx.x < 63u && t < n_groups) c = nxt - a;
    u32 mx = c, over = c > cap ? c : 0u;
    for (int o = 32; o; o >>= 1) { mx = max(mx, (u32)__shfl_xor((int)mx, o, 64)); over += (u32)__shfl_xor((int)over, o, 64); }
    if (threadIdx.x == 0) {
        if (mx) atomicMax(&stats[0], mx);
        if (over) atomicAdd(&stats[1], over);
    }
}
static hipError_t launch_group_bounds_stats(const u64* keys, u32 n, u32 n_groups, u64 g0, int gshift, u32* bounds, u32 cap, u32* stats,
                                            hipStream_t stream, const u32* dev_sizes = nullptr) {
    group_bounds_stats_kernel<<<(n_groups + 1 + 62) / 63, 64, 0, stream>>>(keys, n, n_groups, g0, gshift, bounds, cap, stats, dev_sizes);
    return hipGetLastError();
}
// CTX_KIND >= 0 (with PACKED): also write the scan's per-entry {id, suppression context} word (entry_ctx_kernel's output
// for inactive_below = 0) while key and box code are at hand -- the separate streaming kernel is then not needed
template <bool PACKED, int CTX_KIND>
__global__ __launch_bounds__(TGRP_THREADS) void tgroup_finish_kernel(const u64* __restrict__ keys_in, const u32* __restrict__ ids_in,
                                                                     const u32* __restrict__ bounds, int gshift,
                                                                     u64* __restrict__ keys_out, u32* __restrict__ ids_out,
                                                                     uint8_t* __restrict__ box_out, u64* __restrict__ idctx_out,
                                                                     DevScalars* sc, const u32* __restrict__ stats, u32 abort_over) {
    __shared__ u64 s_key[TGRP_CAP];
    __shared__ u32 s_id[TGRP_CAP];             // payload as it travels (box code packed in when PACKED)
    __shared__ unsigned short s_pos[TGRP_CAP]; // input position inside the group: last tie-break (exact duplicates)
    __shared__ u32 s_cnt[TGRP_BINS], s_off[TGRP_BINS + 1], s_cur[TGRP_BINS];
    __shared__ u32 s_ws[TGRP_THREADS / 64];
    const u32 g = blockIdx.x, tid = threadIdx.x;
    // the scene changed under the plan (most entries now sit in oversize groups): rounds over every such group would cost
    // far more than the full-width sort the host runs when it sees this flag
    if (stats && stats[1] > abort_over) {
        if (g == 0 && tid == 0) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
        return;
    }
    const u32 lo = bounds[g], hi = bounds[g + 1];
    const u32 ng = hi - lo;
    if (ng == 0) return;  // workgroup-uniform
    constexpr u32 NB = (u32)TGRP_BINS, BM = NB - 1u;
    const int bshift = gshift - TGRP_BIN_BITS;  // the host guarantees gshift >= TGRP_BIN_BITS
    auto idof = [](u32 p) { return PACKED ? (p & BOX_ID_MASK) : p; };
    auto less3 = [&](u64 ky, u32 iy, u32 py, u64 kx, u32 ix, u32 px) { return ky < kx || (ky == kx && (iy < ix || (iy == ix && py < px))); };
    if (tid < NB) s_cnt[tid] = 0;
    __syncthreads();
    if (ng <= (u32)TGRP_CAP) {
        u64 k[TGRP_ITEMS];
        u32 p[TGRP_ITEMS];
#pragma unroll
        for (int i = 0; i < TGRP_ITEMS; ++i) {
            const u32 x = (u32)i * TGRP_THREADS + tid;
            k[i] = x < ng ? keys_in[(size_t)lo + x] : 0ull;
            p[i] = x < ng ? ids_in[(size_t)lo + x] : 0u;
        }
#pragma unroll
        for (int i = 0; i < TGRP_ITEMS; ++i) {
            const u32 x = (u32)i * TGRP_THREADS + tid;
            if (x < ng) __hip_atomic_fetch_add(&s_cnt[(u32)(k[i] >> bshift) & BM], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        const u32 mine = tid < NB ? s_cnt[tid] : 0u;
        u32 total;
        const u32 excl = block_excl_scan_u32<TGRP_THREADS>(mine, s_ws, total);
        if (tid < NB) { s_off[tid] = excl; s_cur[tid] = excl; }
        if (tid == 0) s_off[NB] = ng;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TGRP_ITEMS; ++i) {
            const u32 x = (u32)i * TGRP_THREADS + tid;
            if (x < ng) {
                const u32 pos = __hip_atomic_fetch_add(&s_cur[(u32)(k[i] >> bshift) & BM], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                s_key[pos] = k[i];
                s_id[pos] = p[i];
                s_pos[pos] = (unsigned short)x;
            }
        }
        __syncthreads();
        u32 dst[TGRP_ITEMS];
#pragma unroll
        for (int i = 0; i < TGRP_ITEMS; ++i) {
            const u32 x = (u32)i * TGRP_THREADS + tid;
            dst[i] = 0;
            if (x < ng) {
                const u32 b = (u32)(k[i] >> bshift) & BM;
                const u32 start = s_off[b], end = s_off[b + 1];
                const u32 ix = idof(p[i]);
                u32 rank = 0;
                for (u32 y = start; y < end; ++y) rank += less3(s_key[y], idof(s_id[y]), (u32)s_pos[y], k[i], ix, x) ? 1u : 0u;
                dst[i] = start + rank;
            }
        }
        __syncthreads();  // every rank is known: the buffers now take the group in final order
#pragma unroll
        for (int i = 0; i < TGRP_ITEMS; ++i) {
            const u32 x = (u32)i * TGRP_THREADS + tid;
            if (x < ng) { s_key[dst[i]] = k[i]; s_id[dst[i]] = p[i]; }
        }
        __syncthreads();
        for (u32 y = tid; y < ng; y += TGRP_THREADS) {
            const u32 px = s_id[y];
            const u64 ky = s_key[y];
            keys_out[(size_t)lo + y] = ky;
            ids_out[(size_t)lo + y] = idof(px);
            if (PACKED) {
                const u32 code = box_from_packed(px >> BOX_ID_BITS);
                box_out[(size_t)lo + y] = (uint8_t)code;
                if constexpr (CTX_KIND >= 0) idctx_out[(size_t)lo + y] = ((u64)idof(px) << 32) | (u64)make_entry_ctx<CTX_KIND>(ky, code, 0u);
            }
        }
        return;
    }
    // ---- a group that does not fit: rounds over bin ranges that do ------------------------------------------
    if (ng > TGRP_SLOW_MAX) {
        if (tid == 0) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
        return;
    }
    for (u32 x = tid; x < ng; x += TGRP_THREADS)
        __hip_atomic_fetch_add(&s_cnt[(u32)(keys_in[(size_t)lo + x] >> bshift) & BM], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const u32 mine_s = tid < NB ? s_cnt[tid] : 0u;
    u32 total_s;
    const u32 excl_s = block_excl_scan_u32<TGRP_THREADS>(mine_s, s_ws, total_s);
    if (tid < NB) s_off[tid] = excl_s;
    if (tid == 0) s_off[NB] = ng;
    if (mine_s > (u32)TGRP_CAP) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);  // one bin alone does not fit: the host re-sorts
    __syncthreads();
    u32 b0 = 0;
    while (b0 < NB) {  // workgroup-uniform
        u32 b1 = b0 + 1u;
        while (b1 < NB && s_off[b1 + 1u] - s_off[b0] <= (u32)TGRP_CAP) ++b1;
        const u32 base = s_off[b0];
        const u32 cnt = min(s_off[b1] - base, (u32)TGRP_CAP);
        if (tid >= b0 && tid < b1) s_cur[tid] = s_off[tid] - base;
        __syncthreads();
        for (u32 x = tid; x < ng; x += TGRP_THREADS) {
            const u64 kk = keys_in[(size_t)lo + x];
            const u32 b = (u32)(kk >> bshift) & BM;
            if (b >= b0 && b < b1) {
                const u32 pos = __hip_atomic_fetch_add(&s_cur[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (pos < (u32)TGRP_CAP) { s_key[pos] = kk; s_id[pos] = ids_in[(size_t)lo + x]; s_pos[pos] = (unsigned short)x; }
            }
        }
        __syncthreads();
        // ranks go straight to the output (this path is rare; its stores are scattered inside the round's range)
        for (u32 x = tid; x < ng; x += TGRP_THREADS) {
            const u64 kk = keys_in[(size_t)lo + x];
            const u32 b = (u32)(kk >> bshift) & BM;
            if (b >= b0 && b < b1) {
                const u32 px = ids_in[(size_t)lo + x], ix = idof(px);
                const u32 start = s_off[b] - base, end = min(s_off[b + 1u] - base, (u32)TGRP_CAP);
                u32 rank = 0;
                for (u32 y = start; y < end; ++y) rank += less3(s_key[y], idof(s_id[y]), (u32)s_pos[y], kk, ix, x) ? 1u : 0u;
                const size_t d = (size_t)lo + base + start + rank;
                if (start + rank < cnt) {
                    keys_out[d] = kk;
                    ids_out[d] = ix;
                    if (PACKED) {
                        const u32 code = box_from_packed(px >> BOX_ID_BITS);
                        box_out[d] = (uint8_t)code;
                        if constexpr (CTX_KIND >= 0) idctx_out[d] = ((u64)ix << 32) | (u64)make_entry_ctx<CTX_KIND>(kk, code, 0u);
                    }
                }
            }
        }
        __syncthreads();
        b0 = b1;
    }
}

// Composite-key variant of the grouped finish, for the per-frame fast path (ids distinct, box codes packed into the
// ids): inside a group the top bits of the key are equal, and inside a bin of the counting sort the next TG2_BIN_BITS are
// as well -- what is left of the key (rem_bits) and the id (id_bits) fit ONE 64-bit word, rem << id_bits | id, whose
// plain unsigned order is the (key, id) order.  The rank loop then costs one 8-byte LDS read and one 64-bit compare per
// element of the bin (the generic kernel above: three reads and a three-level compare -- its rank loops were 3/4 of its
// instructions), the arrival ticket of the counting pass doubles as the scatter slot (one returning LDS atomic per
// entry instead of a count pass and a cursor pass), and 1024 bins keep entries of different cells apart down to three
// levels below the group.  Equal composites cannot occur: ids are distinct and an object's cells have distinct keys.
// A group beyond the capacity sets KERR_SEGMENT_TOO_LONG (the host sorts full-width and leaves this plan).
constexpr int TG2_BIN_BITS = 10, TG2_BINS = 1 << TG2_BIN_BITS, TG2_BINS_PER_THREAD = TG2_BINS / TGRP_THREADS;
static_assert(TGRP_THREADS == 256 && TG2_BINS_PER_THREAD == 4, "tgroup_finish2: four bins per thread in the prefix scan");
template <int CTX_KIND, bool GAPPED = false>
__global__ __launch_bounds__(TGRP_THREADS) void tgroup_finish2_kernel(const u64* __restrict__ keys_in, const u32* __restrict__ ids_in,
                                                                      const u32* __restrict__ bounds, int gshift, int id_bits,
                                                                      u64* __restrict__ keys_out, u32* __restrict__ ids_out,
                                                                      uint8_t* __restrict__ box_out, u64* __restrict__ idctx_out,
                                                                      DevScalars* sc, const u32* __restrict__ stats, u32 abort_over,
                                                                      const uint4* __restrict__ cell_src = nullptr) {
    __shared__ u64 s_comp[TGRP_CAP];  // composites grouped by bin while ranking, then the keys in final order
    // counts / offsets of the bins while ranking; afterwards the same storage takes the payloads in final order
    union Aux { struct { u32 cnt[TG2_BINS]; u32 off[TG2_BINS + 4]; } b; u32 id[TGRP_CAP]; };
    __shared__ __attribute__((aligned(16))) Aux s_aux;
    __shared__ u32 s_ws[TGRP_THREADS / 64];
    __shared__ u32 s_lut[CTX_KIND >= 0 ? CTX_LUT_WORDS : 1];
    static_assert(sizeof(Aux) >= TGRP_CAP * sizeof(u32), "payload view must fit");
    const u32 g = blockIdx.x, tid = threadIdx.x;
    // (every load of the prologue is requested before the first is used -- the context table, the abort statistic, the group's source
    // word: a workgroup lives a few microseconds, and each dependent round trip at its start is most of one)
    constexpr int LUT_ITERS = CTX_KIND >= 0 ? (CTX_LUT_WORDS + TGRP_THREADS - 1) / TGRP_THREADS : 1;
    u32 lutv[LUT_ITERS];
#pragma unroll
    for (int q = 0; q < LUT_ITERS; ++q) lutv[q] = (CTX_KIND >= 0 && (u32)q * TGRP_THREADS + tid < (u32)CTX_LUT_WORDS) ? g_ctx_lut.e[(u32)q * TGRP_THREADS + tid] : 0u;
    uint4 si = make_uint4(0u, 0u, 0u, 0u);
    if (GAPPED) si = cell_src[g];
    const u32 over_stat = stats ? stats[1] : 0u;
    if (GAPPED) asm volatile("" : "+s"(si.x), "+s"(si.y), "+s"(si.z), "+s"(si.w));
    if (over_stat > abort_over) {
        if (g == 0 && tid == 0) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
        return;
    }
    // GAPPED: the entries are the result of a cell sort -- one 16-byte word per group says where its two pieces lie, where it
    // goes in the dense output and how long it is (launch_cell_count; group_finish_kernel reads the same word)
    u32 lo, ng, src_a, src_nf, src_b = 0;
    if (GAPPED) {
        src_a = si.x; src_b = si.y; lo = si.z; ng = si.w >> 16; src_nf = si.w & 0xFFFFu;
    } else {
        lo = bounds[g];
        ng = bounds[g + 1] - lo;
        src_a = lo; src_nf = ng;
    }
    if (ng == 0) return;  // workgroup-uniform
    auto src = [&](u32 x) -> size_t { return !GAPPED || x < src_nf ? (size_t)src_a + x : (size_t)src_b + (x - src_nf); };
    if (ng > (u32)TGRP_CAP) {
        if (tid == 0) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
        return;
    }
    const int bshift = gshift - TG2_BIN_BITS;  // the host guarantees 0 <= bshift and bshift + id_bits <= 64
    const u64 rem_mask = bshift >= 64 ? ~0ull : ((1ull << bshift) - 1ull);
    const u32 id_mask = id_bits >= 32 ? 0xFFFFFFFFu : ((1u << id_bits) - 1u);
    reinterpret_cast<uint4*>(s_aux.b.cnt)[tid] = make_uint4(0u, 0u, 0u, 0u);
    u64 k[TGRP_ITEMS];
    u32 p[TGRP_ITEMS];
#pragma unroll
    for (int i = 0; i < TGRP_ITEMS; ++i) {
        const u32 x = (u32)i * TGRP_THREADS + tid;
        k[i] = x < ng ? keys_in[src(x)] : 0ull;
        p[i] = x < ng ? ids_in[src(x)] : 0u;
    }
    if constexpr (CTX_KIND >= 0) {  // the table's LDS copy (visible after the barriers below)
#pragma unroll
        for (int q = 0; q < LUT_ITERS; ++q) {
            asm volatile("" : "+v"(lutv[q]));
            if ((u32)q * TGRP_THREADS + tid < (u32)CTX_LUT_WORDS) s_lut[(u32)q * TGRP_THREADS + tid] = lutv[q];
        }
    }
    __syncthreads();
    u32 slot[TGRP_ITEMS];  // arrival ticket inside the bin, then the scatter slot, then the final position
#pragma unroll
    for (int i = 0; i < TGRP_ITEMS; ++i) {
        const u32 x = (u32)i * TGRP_THREADS + tid;
        slot[i] = 0;
        if (x < ng)
            slot[i] = __hip_atomic_fetch_add(&s_aux.b.cnt[(u32)(k[i] >> bshift) & (TG2_BINS - 1u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    {   // exclusive prefix over the 1024 bin counts: four consecutive bins per thread
        const uint4 c = reinterpret_cast<const uint4*>(s_aux.b.cnt)[tid];
        u32 total;
        const u32 excl = block_excl_scan_u32<TGRP_THREADS>(c.x + c.y + c.z + c.w, s_ws, total);
        reinterpret_cast<uint4*>(s_aux.b.off)[tid] = make_uint4(excl, excl + c.x, excl + c.x + c.y, excl + c.x + c.y + c.z);
        if (tid == 0) s_aux.b.off[TG2_BINS] = ng;
    }
    __syncthreads();
    u64 comp[TGRP_ITEMS];
#pragma unroll
    for (int i = 0; i < TGRP_ITEMS; ++i) {
        const u32 x = (u32)i * TGRP_THREADS + tid;
        comp[i] = 0;
        if (x < ng) {
            comp[i] = ((k[i] & rem_mask) << id_bits) | (u64)(p[i] & id_mask);
            slot[i] += s_aux.b.off[(u32)(k[i] >> bshift) & (TG2_BINS - 1u)];
            s_comp[slot[i]] = comp[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TGRP_ITEMS; ++i) {
        const u32 x = (u32)i * TGRP_THREADS + tid;
        if (x < ng) {
            const u32 b = (u32)(k[i] >> bshift) & (TG2_BINS - 1u);
            const u32 start = s_aux.b.off[b], end = s_aux.b.off[b + 1];
            u32 rank = 0;
#pragma unroll 2
            for (u32 y = start; y < end; ++y) rank += s_comp[y] < comp[i] ? 1u : 0u;
            slot[i] = start + rank;
        }
    }
    __syncthreads();  // every rank is known: the buffers now take the group in final order (bin bookkeeping is dead)
#pragma unroll
    for (int i = 0; i < TGRP_ITEMS; ++i) {
        const u32 x = (u32)i * TGRP_THREADS + tid;
        if (x < ng) { s_comp[slot[i]] = k[i]; s_aux.id[slot[i]] = p[i]; }
    }
    __syncthreads();
    for (u32 y = tid; y < ng; y += TGRP_THREADS) {
        const u32 px = s_aux.id[y];
        const u64 ky = s_comp[y];
        const u32 c27 = px >> BOX_ID_BITS;
        keys_out[(size_t)lo + y] = ky;
        ids_out[(size_t)lo + y] = px & BOX_ID_MASK;
        box_out[(size_t)lo + y] = (uint8_t)box_from_packed(c27);
        if constexpr (CTX_KIND >= 0) idctx_out[(size_t)lo + y] = ((u64)(px & BOX_ID_MASK) << 32) | (u64)entry_ctx_lut<CTX_KIND>(ky, c27, 0u, s_lut);
    }
}

// ---- compact tree records (TreeCompact, bp_kernels.h) ---------------------------------------------------------------------
// composite -> index value, id, packed box code.  The tail's digits are taken three at a time through a table of BASE^3 entries
// (lut3: bits 0 .. 3 DIM - 1 = the three levels' Morton bits, bits 12 .. 13 = how many of the digits are levels the entry has);
// lut3 == nullptr: digit by digit
template <int DIM> struct CompactLut3 {
    static constexpr u32 BASE = (1u << DIM) + 1u, N = BASE * BASE * BASE, WORDS = (N + 1u) / 2u;
    unsigned short e[WORDS * 2];
};
template <int DIM> constexpr CompactLut3<DIM> make_compact_lut3() {
    CompactLut3<DIM> l{};
    constexpr u32 BASE = CompactLut3<DIM>::BASE;
    for (u32 v = 0; v < CompactLut3<DIM>::N; ++v) {
        const u32 d0 = v / (BASE * BASE), d1 = (v / BASE) % BASE, d2 = v % BASE;  // most significant digit = the shallowest level
        const u32 m = ((d0 ? d0 - 1u : 0u) << (2 * DIM)) | ((d1 ? d1 - 1u : 0u) << DIM) | (d2 ? d2 - 1u : 0u);
        l.e[v] = (unsigned short)(m | (((d0 ? 1u : 0u) + (d1 ? 1u : 0u) + (d2 ? 1u : 0u)) << 12));
    }
    return l;
}
__constant__ CompactLut3<3> g_compact_lut3_3d = make_compact_lut3<3>();
__constant__ CompactLut3<2> g_compact_lut3_2d = make_compact_lut3<2>();
// NSTEP: table reads the tail takes (1: T <= 3, 2: T <= 6 -- compile-time, so that the per-entry code has no branches on the
// format), 0: any T, by table when lut3 is given, else digit by digit
template <int KIND, int NSTEP = 0> __device__ __forceinline__ void tree_expand(u64 comp, const TreeCompact& f, u64& key, u32& id, u32& c27,
                                                                               const unsigned short* lut3 = nullptr, u32* nlev_out = nullptr) {
    constexpr u32 DIM = KT<KIND>::DIM, BASE = (1u << DIM) + 1u, N3 = BASE * BASE * BASE;
    const u64 kp = comp >> f.low;
    u32 t = (u32)kp & (u32)((1ull << f.tbits) - 1ull);  // (tbits <= 32: tree_compact_format keeps the tail a 32-bit number)
    u64 mort = 0;   // Morton bits of the levels dmin + 1 .. dmax, level dmax lowest
    u32 nlev = 0;
    if constexpr (NSTEP == 1) {
        const u32 e0 = lut3[t];
        mort = e0 & 0xFFFu;
        nlev = e0 >> 12;
    } else if constexpr (NSTEP == 2) {
        // t < BASE^6 (531441 in 3-D, 15625 in 2-D): the quotient by BASE^3 from ONE full-rate 24-bit multiply -- ceil(2^32 / 729) has 23
        // bits and errs by less than 1 / 729 below 2^20; ceil(2^24 / 125) by less than 1 / 125 below 2^14 (a 32-bit division by a
        // constant is a v_mul_hi_u32 plus shifts and fix-ups: four to six instructions)
        const u32 q = DIM == 3 ? (u32)(((u64)(t & 0xFFFFFu) * 5891589ull) >> 32) : (mul24k<134218u>(t & 0x3FFFu) >> 24);
        const u32 e0 = lut3[t - mul24k<N3>(q)], e1 = lut3[q];
        mort = (e0 & 0xFFFu) | ((e1 & 0xFFFu) << (3 * DIM));
        nlev = (e0 >> 12) + (e1 >> 12);
    } else if (lut3) {
        for (int sh = 0, left = f.T; left > 0; left -= 3, sh += 3 * (int)DIM) {  // three levels per table read, deepest first
            const u32 q = t / N3, e = lut3[t - q * N3];
            mort |= (u64)(e & 0xFFFu) << sh;
            nlev += e >> 12;
            t = q;
        }
    } else {
        for (int k = 0; k < f.T; ++k) {
            const u32 q = t / BASE, d = t - q * BASE;
            if (d) { mort |= (u64)(d - 1u) << (DIM * k); ++nlev; }
            t = q;
        }
    }
    const u32 depth = f.dmin + nlev;
    if (nlev_out) *nlev_out = nlev;
    key = ((kp >> f.tbits) << f.pshift) | (mort << (f.pshift - (int)DIM * f.T)) | (u64)depth;
    id = ((u32)(comp >> 5) & ((1u << f.idbits) - 1u)) + f.id_base;
    c27 = (u32)comp & 31u;
}
template <int KIND>
__global__ __launch_bounds__(256) void tree_expand_kernel(u64* __restrict__ words, u32* __restrict__ ids_out, uint8_t* __restrict__ box_out, size_t n,
                                                          TreeCompact f) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u64 key; u32 id, c27;
        tree_expand<KIND>(words[i], f, key, id, c27);
        words[i] = key;
        ids_out[i] = id;
        box_out[i] = (uint8_t)box_from_packed(c27);
    }
}
// Grouped finish of a cell sort of composite words: tgroup_finish2_kernel<.., GAPPED> with ONE word per record -- one load per
// entry, the bin and the rank straight from the word (inside a group the bits above comp_gshift are equal, inside a bin the next
// TG2_BIN_BITS as well: the plain unsigned order of the words is the (index, id) order), expansion to index value / id / box byte /
// context word while writing out.
#ifndef TGC_WAVES_PER_EU
#define TGC_WAVES_PER_EU 7
#endif
constexpr int TGC_CAP = 2044;  // (words per group)
// LDS: the words (+ two words of padding: the straight-line reads of a bin's first three), the bins' offsets -- alive to the end since round 5:
// the range lengths are ranks of cell bounds, found through the bins -- and the write-out's tables: 22.1 KB, seven workgroups per CU
// (round 4 measured seven against eight: 0.228 / 0.230 ms)
static_assert(TGC_CAP <= TGRP_CAP, "tgroup_finishc: group capacity");
// WRITE_IDS = false: ids_out / box_out are not written -- the caller derives them from the context words when something asks
// (launch_ids_from_ctx): two of the write-out's four stores and five of its 21 bytes per entry
// dcnt_out (round 5): per entry, depth | range length << 5 -- how many entries behind it lie in its cell, i.e. the candidates the scan
// pairs it with as the ancestor (src/layer.rs:550-573: the entries that find it on the stack).  In composite space an entry of depth
// dmin + k covers the words [kp, kp + BASE^(T - k)) << low; every entry of the tree is at least as deep as its group's cell (the group
// shift lies inside the prefix), so the range ends inside the group and its end is the RANK of that bound in the group: the bins'
// offsets give the bin's start, and the words of one bin (about one) are compared directly -- no search, and the scan loads no key
template <int CTX_KIND, int NSTEP, bool WRITE_IDS = true>
__global__ __launch_bounds__(TGRP_THREADS) __attribute__((amdgpu_waves_per_eu(TGC_WAVES_PER_EU, TGC_WAVES_PER_EU))) void tgroup_finishc_kernel(const u64* __restrict__ words_in, int gshift, TreeCompact fmt,
                                                                      u64* __restrict__ keys_out, u32* __restrict__ ids_out,
                                                                      uint8_t* __restrict__ box_out, u64* __restrict__ idctx_out,
                                                                      unsigned short* __restrict__ dcnt_out,
                                                                      DevScalars* sc, const u32* __restrict__ stats, u32 abort_over,
                                                                      const uint4* __restrict__ cell_src) {
    static_assert(CTX_KIND >= 0 && CTX_KIND <= 2, "the compact route always writes the context words (CTX_KIND = the index kind)");
    constexpr int XK = CTX_KIND;
    // s_comp: the bins' arrival counters, then the words grouped by bin, then the words in final order (the counters' last read is
    // the prefix scan's, two barriers before the first word is placed).  s_off: the bins' offsets (+ the block scan's scratch behind them)
    constexpr int XDIM = (int)KT<XK>::DIM;
    constexpr u32 XBASE = (1u << XDIM) + 1u;
    __shared__ __attribute__((aligned(16))) u64 s_comp[TGC_CAP + 2];
    u32* const s_cnt = reinterpret_cast<u32*>(s_comp);
    __shared__ __attribute__((aligned(16))) u32 s_off[TG2_BINS + 4];
    u32* const s_ws = s_off + TG2_BINS;   // (four words: the block scan's scratch, then s_off[TG2_BINS] = the group's size)
    __shared__ u32 s_lut[CTX_LUT_WORDS];
    __shared__ u32 s_pw[16];              // XBASE^k: the composite span of a cell k levels above the deepest
    __shared__ __attribute__((aligned(4))) unsigned short s_lut3[CompactLut3<XDIM>::WORDS * 2];
    static_assert((size_t)TGC_CAP * 8 >= (size_t)TG2_BINS * 4, "the counters fit the words' storage");
    static_assert((size_t)(TGC_CAP + 2) * 8 + (TG2_BINS + 4) * 4 + CTX_LUT_WORDS * 4 + 64 + CompactLut3<XDIM>::WORDS * 4 <= 163840 / 7, "tgroup_finishc: seven workgroups per CU");
    const u32 g = blockIdx.x, tid = threadIdx.x;
    // (the group's source word and the abort statistic are requested TOGETHER: a workgroup lives ~6 us, and every dependent load at
    // its start is most of a microsecond in which it holds its LDS and does nothing)
    uint4 si = cell_src[g];
    const u32 over_stat = stats ? stats[1] : 0u;
    asm volatile("" : "+s"(si.x), "+s"(si.y), "+s"(si.z), "+s"(si.w));  // (else the source word's load sinks below the abort test: one more round trip)
    if (over_stat > abort_over) {
        if (g == 0 && tid == 0) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
        return;
    }
    const u32 src_a = si.x, src_b = si.y, lo = si.z, ng = si.w >> 16, src_nf = si.w & 0xFFFFu;
    if (ng == 0) return;  // workgroup-uniform
    if (ng > (u32)TGC_CAP) {
        if (tid == 0) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
        return;
    }
    const int bshift = gshift - TG2_BIN_BITS;  // the host guarantees bshift >= 0
    reinterpret_cast<uint4*>(s_cnt)[tid] = make_uint4(0u, 0u, 0u, 0u);
    // Every load of the prologue is issued before the first is used -- the write-out's tables (constant memory behind the L2) and the
    // group's words: left in source order, each table's LDS store waited for its load in front of the next one's issue, and a
    // workgroup that lives ~6 us spent three more round trips at its start
    constexpr u32 L3_ITERS = (CompactLut3<XDIM>::WORDS + TGRP_THREADS - 1) / TGRP_THREADS;
    const u32* const src3 = reinterpret_cast<const u32*>(XDIM == 3 ? (const void*)g_compact_lut3_3d.e : (const void*)g_compact_lut3_2d.e);
    u32 lutv = 0, l3v[L3_ITERS];
    if (tid < (u32)CTX_LUT_WORDS) lutv = g_ctx_lut.e[tid];
#pragma unroll
    for (u32 k = 0; k < L3_ITERS; ++k) {
        const u32 q = k * TGRP_THREADS + tid;
        l3v[k] = q < CompactLut3<XDIM>::WORDS ? src3[q] : 0u;
    }
    u64 c[TGRP_ITEMS];
#pragma unroll
    for (int i = 0; i < TGRP_ITEMS; ++i) {
        const u32 x = (u32)i * TGRP_THREADS + tid;
        c[i] = x < ng ? words_in[x < src_nf ? (size_t)src_a + x : (size_t)src_b + (x - src_nf)] : 0ull;
    }
    asm volatile("" : "+v"(lutv));
#pragma unroll
    for (u32 k = 0; k < L3_ITERS; ++k) asm volatile("" : "+v"(l3v[k]));
    // the tables' LDS copies (their own storage; visible after the barriers below)
    if (tid < (u32)CTX_LUT_WORDS) s_lut[tid] = lutv;
    if (tid < 16u) {
        u32 pw = 1u;
        for (u32 k = 0; k < tid; ++k) pw *= XBASE;  // (wraps beyond what a 32-bit tail holds: never indexed there)
        s_pw[tid] = pw;
    }
#pragma unroll
    for (u32 k = 0; k < L3_ITERS; ++k) {
        const u32 q = k * TGRP_THREADS + tid;
        if (q < CompactLut3<XDIM>::WORDS) reinterpret_cast<u32*>(s_lut3)[q] = l3v[k];
    }
    __syncthreads();
    u32 slot[TGRP_ITEMS];  // arrival ticket inside the bin, then the scatter slot, then the final position
#pragma unroll
    for (int i = 0; i < TGRP_ITEMS; ++i) {
        const u32 x = (u32)i * TGRP_THREADS + tid;
        slot[i] = 0;
        if (x < ng)
            slot[i] = __hip_atomic_fetch_add(&s_cnt[(u32)(c[i] >> bshift) & (TG2_BINS - 1u)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    {   // exclusive prefix over the 1024 bin counts: four consecutive bins per thread
        const uint4 n4 = reinterpret_cast<const uint4*>(s_cnt)[tid];
        u32 total;
        const u32 excl = block_excl_scan_u32<TGRP_THREADS>(n4.x + n4.y + n4.z + n4.w, s_ws, total);
        reinterpret_cast<uint4*>(s_off)[tid] = make_uint4(excl, excl + n4.x, excl + n4.x + n4.y, excl + n4.x + n4.y + n4.z);
        if (tid == 0) s_off[TG2_BINS] = ng;   // (the scan's scratch is free: its last read lies before the scan's final barrier)
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TGRP_ITEMS; ++i) {
        const u32 x = (u32)i * TGRP_THREADS + tid;
        if (x < ng) {
            slot[i] += s_off[(u32)(c[i] >> bshift) & (TG2_BINS - 1u)];
            s_comp[slot[i]] = c[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TGRP_ITEMS; ++i) {
        const u32 x = (u32)i * TGRP_THREADS + tid;
        if (x < ng) {
            const u32 b = (u32)(c[i] >> bshift) & (TG2_BINS - 1u);
            const u32 start = s_off[b], len = s_off[b + 1] - start;
            // about one word per bin: the first three are read whatever the bin's length and masked by it (straight-line code),
            // the loop only runs for the few longer bins (group_finish_kernel has the same shape, and the same reason for it)
            const u64* const q = s_comp + start;
            const u64 q0 = q[0], q1 = q[1], q2 = q[2];
            u32 rank = ((u32)(len > 0u) & (u32)(q0 < c[i])) + ((u32)(len > 1u) & (u32)(q1 < c[i])) + ((u32)(len > 2u) & (u32)(q2 < c[i]));
            if (len > 3u) {
#pragma unroll 1
                for (u32 y = 3u; y < len; ++y) rank += q[y] < c[i] ? 1u : 0u;
            }
            slot[i] = start + rank;
        }
    }
    __syncthreads();  // every rank is known: the buffer now takes the group in final order
#pragma unroll
    for (int i = 0; i < TGRP_ITEMS; ++i) {
        const u32 x = (u32)i * TGRP_THREADS + tid;
        if (x < ng) s_comp[slot[i]] = c[i];
    }
    __syncthreads();
    for (u32 y = tid; y < ng; y += TGRP_THREADS) {
        u64 ky; u32 id, c27, nlev;
        const u64 wy = s_comp[y];
        tree_expand<XK, NSTEP>(wy, fmt, ky, id, c27, s_lut3, &nlev);
        keys_out[(size_t)lo + y] = ky;
        if constexpr (WRITE_IDS) {
            ids_out[(size_t)lo + y] = id;
            box_out[(size_t)lo + y] = (uint8_t)box_from_packed(c27);
        }
        idctx_out[(size_t)lo + y] = ((u64)id << 32) | (u64)entry_ctx_lut<CTX_KIND>(ky, c27, 0u, s_lut);
        // the range's end = the rank of the cell's upper bound among the group's words (all of them sorted by now)
        const u64 ub = ((wy >> fmt.low) + (u64)s_pw[(u32)fmt.T - nlev]) << fmt.low;
        const u32 b = (u32)(ub >> bshift) & (TG2_BINS - 1u);
        const u32 start = s_off[b], len = s_off[b + 1] - start;
        const u64* const q = s_comp + start;
        const u64 q0 = q[0], q1 = q[1], q2 = q[2];
        u32 end = start + ((u32)(len > 0u) & (u32)(q0 < ub)) + ((u32)(len > 1u) & (u32)(q1 < ub)) + ((u32)(len > 2u) & (u32)(q2 < ub));
        if (len > 3u) {
#pragma unroll 1
            for (u32 z = 3u; z < len; ++z) end += q[z] < ub ? 1u : 0u;
        }
        if (((ub ^ wy) >> gshift) != 0ull) end = ng;  // (the bound lies in the next group: everything behind y is inside)
        dcnt_out[(size_t)lo + y] = (unsigned short)((fmt.dmin + nlev) | ((end - y - 1u) << 5));
    }
}

template <int DIM>
__global__ __launch_bounds__(256) void filter_pairs_aabb_kernel(const u64* __restrict__ in, size_t n, const float* __restrict__ bounds, int shape,
                                                                u64* __restrict__ out, DevScalars* sc) {
    constexpr int ITEMS = 4;
    __shared__ u64 s_keep[256 * ITEMS];
    __shared__ u32 s_cursor;
    __shared__ u64 s_base;
    if (threadIdx.x == 0) s_cursor = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * 256 * ITEMS;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const size_t i = base + (size_t)k * 256 + threadIdx.x;
        if (i < n) {
            const u64 v = in[i];
            const float* a = bounds + (size_t)(u32)(v >> 32) * filt_stride<DIM>(shape);
            const float* b = bounds + (size_t)(u32)v * filt_stride<DIM>(shape);
            if (narrow_keep<DIM>(a, b, shape)) s_keep[atomicAdd(&s_cursor, 1u)] = v;
        }
    }
    __syncthreads();
    const u32 total = s_cursor;
    if (threadIdx.x == 0) s_base = total ? atomicAdd(&sc->total, (u64)total) : 0ull;
    __syncthreads();
    for (u32 k = threadIdx.x; k < total; k += 256) out[s_base + k] = s_keep[k];
}

__global__ __launch_bounds__(256) void pack_pairs_kernel(const uint2* __restrict__ in, size_t n, u64* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint2 v = in[i];
        out[i] = ((u64)v.x << 32) | (u64)v.y;
    }
}
__global__ __launch_bounds__(256) void unpack_pairs_kernel(const u64* __restrict__ in, size_t n, uint2* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u64 v = in[i];
        out[i] = make_uint2((u32)(v >> 32), (u32)v);
    }
}

inline unsigned grid_for(size_t n, int threads, unsigned cap) {
    size_t b = (n + threads - 1) / threads;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

}  // namespace

// =================================================================================================
// launchers
// =================================================================================================
#define BP_DISPATCH_KIND(kind, EXPR)                 \
    switch (kind) {                                  \
        case 0: { constexpr int K = 0; EXPR; } break;  \
        case 1: { constexpr int K = 1; EXPR; } break;  \
        default: { constexpr int K = 2; EXPR; } break; \
    }

hipError_t launch_encode_count(int kind, SysBounds sb, const float* bounds, const uint32_t* ids, size_t n,
                               uint32_t min_depth, int have_prev_id, uint32_t prev_id, void* recs,
                               unsigned long long* block_sums, BlockAux* aux, DevScalars* sc, hipStream_t stream,
                               unsigned long long* zero_a, size_t n_a, unsigned long long* zero_b, size_t n_b) {
    if (n == 0) return hipSuccess;
    const unsigned g = (unsigned)enc_blocks(n);
    BP_DISPATCH_KIND(kind, (encode_count_kernel<K><<<g, ENC_THREADS, 0, stream>>>(sb, bounds, ids, n, min_depth, have_prev_id,
                                                                                  prev_id, (uint4*)recs, block_sums, aux, sc,
                                                                                  (u64*)zero_a, zero_a ? (u32)n_a : 0u, (u64*)zero_b,
                                                                                  zero_b ? (u32)n_b : 0u)));
    return hipGetLastError();
}

hipError_t launch_encode_emit(int kind, SysBounds sb, const float* bounds, const uint32_t* ids, size_t n,
                              uint32_t min_depth, const void* recs, const unsigned long long* block_offs,
                              uint64_t* keys_out, uint32_t* ids_out, uint8_t* box_out, hipStream_t stream,
                              const RadixPlan* hist_plan, unsigned long long* hist, int pack_box, unsigned long long* cell_hist,
                              uint32_t cell_depth, DevScalars* sc_guard, size_t capacity, const TreeCompact* compact) {
    if (n == 0) return hipSuccess;
    if (compact && cell_hist) return hipErrorInvalidValue;
    const unsigned g = (unsigned)enc_blocks(n);
    DigitCount dc;
    const KindInfo ki = kind_info(kind);
    dc.cell_hist = nullptr; dc.cell_shift = 0; dc.cell_depth = 0; dc.n_cells = 0; dc.depth_mask = (1u << ki.depth_bits) - 1u;
    if (cell_hist && cell_depth > 0 && ki.dim * cell_depth <= 12 && (1u << (ki.dim * cell_depth)) <= (uint32_t)ENC_CELLS_MAX) {
        dc.cell_hist = (u64*)cell_hist;
        dc.cell_depth = cell_depth;
        dc.n_cells = 1u << (ki.dim * cell_depth);
        dc.cell_shift = (int)(ki.dim * ki.axis_bits + ki.depth_bits - ki.dim * cell_depth);
    }
    dc.hist = (u64*)hist; dc.passes = 0;
    for (int q = 0; q < SX_HIST_MAX; ++q) dc.shift[q] = 0;
    if (hist_plan && hist && hist_plan->n_passes <= SX_HIST_MAX) {
        bool key_digits_only = true;
        for (int q = 0; q < hist_plan->n_passes; ++q) key_digits_only = key_digits_only && !hist_plan->from_val[q];
        if (key_digits_only) {
            dc.passes = hist_plan->n_passes;
            for (int q = 0; q < hist_plan->n_passes; ++q) dc.shift[q] = hist_plan->shift[q];
        }
    }
    if (compact) {
        // (a digit at shift s of the word is the index value's at s + pshift - tbits - low, for shifts inside the word's prefix part)
        const int ksd = compact->pshift - compact->tbits - compact->low;
        for (int q = 0; q < dc.passes; ++q)
            if (dc.shift[q] < compact->tbits + compact->low) return hipErrorInvalidValue;  // (the counted digits lie in the groups' bits)
        BP_DISPATCH_KIND(kind, (encode_emit_compact_kernel<K><<<g, ENC_THREADS, 0, stream>>>(sb, bounds, ids, n, min_depth, (const uint4*)recs, block_offs,
                                                                                         (u64*)keys_out, dc, sc_guard, (u64)capacity, *compact, ksd)));
    } else if (dc.cell_hist) {
        BP_DISPATCH_KIND(kind, (encode_emit_kernel<K, true><<<g, ENC_THREADS, 0, stream>>>(sb, bounds, ids, n, min_depth, (const uint4*)recs,
                                                                                       block_offs, (u64*)keys_out, ids_out, box_out, dc, pack_box, sc_guard, (u64)capacity)));
    } else {
        BP_DISPATCH_KIND(kind, (encode_emit_kernel<K, false><<<g, ENC_THREADS, 0, stream>>>(sb, bounds, ids, n, min_depth, (const uint4*)recs,
                                                                                        block_offs, (u64*)keys_out, ids_out, box_out, dc, pack_box, sc_guard, (u64)capacity)));
    }
    return hipGetLastError();
}

// workgroups of a grid that places at most per_cu of them on a CU
static unsigned resident_cap(int per_cu) {
    static int cus = 0;
    if (!cus) {
        int dev = 0, v = 0;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return (unsigned)(cus * per_cu);
}
// resident workgroups of a persistent kernel = CUs x occupancy (queried once per kernel)
template <typename Kern>
static unsigned resident_blocks(Kern kernel, int threads, size_t dyn_lds = 0) {
    int dev = 0, cus = 256, per_cu = 4;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, dyn_lds) == hipSuccess && nb > 0) per_cu = nb;
    (void)hipGetLastError();
    return (unsigned)(cus * per_cu);
}

hipError_t launch_publish(const void* src, size_t bytes, const Published& pub, hipStream_t stream, const void* src2, size_t bytes2, size_t dst_off2) {
    if (!pub.dst || !pub.flag || ((bytes | bytes2 | dst_off2) & 3u) != 0 || bytes > (1u << 20) || bytes2 > (1u << 20)) return hipErrorInvalidValue;
    if (src2 && dst_off2 < bytes) return hipErrorInvalidValue;
    publish_kernel<<<1, 256, 0, stream>>>((const u32*)src, (u32)(bytes / 4), pub, (const u32*)src2, src2 ? (u32)(bytes2 / 4) : 0u, (u32)(dst_off2 / 4));
    return hipGetLastError();
}
hipError_t launch_scan_block_sums(unsigned long long* sums, size_t n, const BlockAux* aux, DevScalars* sc,
                                  hipStream_t stream, const DevScalars* init, unsigned long long* zero_a, size_t n_a,
                                  unsigned long long* zero_b, size_t n_b, const Published* pub) {
    if (n_a >= 0xFFFFFFFFull || n_b >= 0xFFFFFFFFull || (init && !aux)) return hipErrorInvalidValue;
    DevScalars iv = {};
    if (init) iv = *init;
    scan_block_sums_kernel<<<1, SBS_THREADS, 0, stream>>>(sums, n, aux, sc, iv, init ? 1 : 0, (u64*)zero_a, zero_a ? (u32)n_a : 0u,
                                                          (u64*)zero_b, zero_b ? (u32)n_b : 0u, pub ? *pub : Published{});
    return hipGetLastError();
}

hipError_t launch_tree_reduce(int kind, const uint64_t* keys, const uint32_t* ids, size_t m, DevScalars* sc,
                              hipStream_t stream) {
    if (m == 0) return hipSuccess;
    const unsigned g = grid_for(m, 256, 1024);
    BP_DISPATCH_KIND(kind, (tree_reduce_kernel<K><<<g, 256, 0, stream>>>((const u64*)keys, ids, m, sc)));
    return hipGetLastError();
}

RadixPlan make_radix_plan(unsigned long long key_varying, int use_ids, unsigned id_varying) {
    RadixPlan plan;
    plan.n_passes = 0;
    for (int i = 0; i < RS_MAX_PASSES; ++i) plan.shift[i] = plan.from_val[i] = 0;
    if (use_ids) {
        unsigned v = id_varying;
        while (v && plan.n_passes < RS_MAX_PASSES) {
            const int s = __builtin_ctz(v);
            plan.shift[plan.n_passes] = s;
            plan.from_val[plan.n_passes] = 1;
            ++plan.n_passes;
            v = s + 8 >= 32 ? 0u : (v >> (s + 8)) << (s + 8);
        }
    }
    unsigned long long v = key_varying;
    while (v && plan.n_passes < RS_MAX_PASSES) {
        const int s = __builtin_ctzll(v);
        plan.shift[plan.n_passes] = s;
        plan.from_val[plan.n_passes] = 0;
        ++plan.n_passes;
        v = s + 8 >= 64 ? 0ull : (v >> (s + 8)) << (s + 8);
    }
    return plan;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }


// tile geometry of radix_onesweep_kernel: 512 threads x 16 records (8192-record tiles, 2 workgroups per CU)
#ifndef RS_THREADS_MACRO
#define RS_THREADS_MACRO 512
#endif
#ifndef RS_ITEMS_MACRO
#define RS_ITEMS_MACRO 16
#endif
constexpr int RS_THREADS_DEF = RS_THREADS_MACRO, RS_ITEMS_DEF = RS_ITEMS_MACRO;
static size_t rs_tile_size() { return (size_t)RS_THREADS_DEF * RS_ITEMS_DEF; }
static size_t rs_num_tiles(size_t n) { return (n + rs_tile_size() - 1) / rs_tile_size(); }

// Workspace of one sort: a header of FIXED size (laid out for RS_MAX_PASSES passes whatever the plan, so that the region
// behind it only ever holds launch-tagged status words, even when sorts of different pass counts share the workspace):
// [hist: 12 x 256 u64][bin_base: 12 x 256 u64][tile tickets: 12 u32, padded][line cursors | fragment cursors of the
// claiming first pass] -- then [status: passes x tiles x 256 u64]
static size_t rs_header_bytes() {
    return 2 * (size_t)RS_MAX_PASSES * RS_BINS * sizeof(u64) + align_up((size_t)RS_MAX_PASSES * RS_TICKETS * sizeof(u32), 256) +
           align_up(2 * SW_CURSOR_WORDS * sizeof(u32), 256);
}
size_t radix_workspace_bytes(size_t n, int n_passes) {
    const size_t np = (size_t)(n_passes > 0 ? n_passes : 1);
    return align_up(rs_header_bytes() + np * rs_num_tiles(n) * RS_BINS * sizeof(u64), 256);
}

// the claiming first pass (radix_claim_kernel): one persistent workgroup per CU
template <typename V, bool CELLS = false, int TAG = 0>
static hipError_t launch_claim(const ClaimArgs& a, hipStream_t stream) {
#ifndef RS_GEOMETRY_EXPERIMENT
    static_assert(RC_TILE == RS_THREADS_DEF * RS_ITEMS_DEF, "the pass kernels share the tile size (holes, workspace)");
#endif
    static const unsigned resident = resident_blocks(radix_claim_kernel<V, CELLS, TAG>, RC_THREADS);
    const unsigned grid = a.n_tiles < resident ? a.n_tiles : resident;
    radix_claim_kernel<V, CELLS, TAG><<<grid, RC_THREADS, 0, stream>>>(a);
    return hipGetLastError();
}

// ---- cell route (CellRoute, bp_kernels.h) --------------------------------------------------------------------------------
// ---- cell route (CellRoute, bp_kernels.h) --------------------------------------------------------------------------------
// (+ 25 % + 64: between two frames of the moving-scene bench -- every object re-placed by up to half its size -- the worst of 39 K
// cells grows by 18 - 22 %; the gaps cost address space in a buffer sized for the scan's raw output anyway, not bandwidth)
__device__ __forceinline__ u32 cell_capacity(u32 cnt) { return cnt ? (cnt + cnt / 4u + 64u + (u32)(RC_LINE - 1)) & ~(u32)(RC_LINE - 1) : (u32)RC_LINE; }
// A walk over the 65536 cells with an exclusive prefix, in ONE launch of CELL_WGS workgroups: every workgroup loads its 1024
// cells (consecutive threads on consecutive cells, all loads of a thread issued before anything is computed), publishes their
// sum in a status word tagged with the launch (no zero fill between launches) and adds up the sums of the workgroups before
// it -- all CELL_WGS workgroups are resident, so the wait is a few hundred nanoseconds.  (One workgroup walking all cells took
// 25 - 115 us: a single CU's memory pipe.)
constexpr int CELL_WGS = 64, CELL_THREADS = 256, CELL_PER_THREAD = (int)CELL_COUNT / (CELL_WGS * CELL_THREADS);
static_assert(CELL_PER_THREAD == 4, "cell walk geometry");
constexpr u64 CELL_SUM_MASK = (1ull << 40) - 1ull;
// (the wait for an earlier workgroup's sum is bounded like the radix look-back's: it cannot run out while workgroups are dispatched
// in order, and must not hang the device should that ever not hold -- a wait that does run out poisons the prefix with 2^39, which
// every later workgroup then runs into as well: the last one's total fails every caller's check)
template <class Load, class Value, class Emit>
__device__ __forceinline__ u64 cells_walk(u64* status, u32 tag, Load&& load, Value&& value, Emit&& emit) {
    __shared__ u32 s_ws[CELL_THREADS / 64];
    __shared__ u64 s_before;
    const u32 t = threadIdx.x, b = blockIdx.x;
    // (a thread takes FOUR CONSECUTIVE cells: one block scan of the threads' sums instead of one per row of cells -- each is two
    // barriers in a kernel that is nothing but a chain of latencies)
    uint4 raw[CELL_PER_THREAD];
    u32 v[CELL_PER_THREAD], ex[CELL_PER_THREAD];
    const u32 c0 = b * 1024u + t * (u32)CELL_PER_THREAD;
#pragma unroll
    for (int k = 0; k < CELL_PER_THREAD; ++k) raw[k] = load(c0 + (u32)k);
    __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler pairs every cell's loads with their use: a wait per cell)
    u32 tsum = 0;
#pragma unroll
    for (int k = 0; k < CELL_PER_THREAD; ++k) {
        v[k] = value(c0 + (u32)k, raw[k]);
        ex[k] = tsum;
        tsum += v[k];
    }
    u32 wg_total = 0;
    const u32 tex = block_excl_scan_u32<CELL_THREADS>(tsum, s_ws, wg_total);
#pragma unroll
    for (int k = 0; k < CELL_PER_THREAD; ++k) ex[k] += tex;
    if (t == 0u) __hip_atomic_store(&status[b], ((u64)tag << 40) | (u64)wg_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < 64u) {  // wave 0: the sums of the workgroups before this one
        u64 mine = 0;
        if (t < b) {
            u32 spins = 0;
            for (;;) {
                const u64 w = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((u32)(w >> 40) == tag) { mine = w & CELL_SUM_MASK; break; }
                if (++spins > LOOKBACK_SPIN_LIMIT) { mine = 1ull << 39; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        for (int o = 32; o; o >>= 1)
            mine += ((u64)(u32)__shfl_xor((int)(u32)(mine >> 32), o, 64) << 32) | (u32)__shfl_xor((int)(u32)mine, o, 64);
        if (t == 0u) s_before = mine;
    }
    __syncthreads();
    const u64 before = s_before;
#pragma unroll
    for (int k = 0; k < CELL_PER_THREAD; ++k) emit(c0 + (u32)k, (u32)before + ex[k], v[k], raw[k]);
    return before + wg_total;  // (the last workgroup: the grand total)
}
__device__ __forceinline__ void cell_layout_body(const u32* __restrict__ prev_bounds, u32 prev_g0, u32 prev_groups,
                                                 u32* __restrict__ cell_base, u32* __restrict__ line_cur,
                                                 u32* __restrict__ frag_cur, u32* __restrict__ miss, u64 limit,
                                                 u64* status, u32 tag) {
    // (capacities are clamped: prev_bounds may be anything when the caller's bookkeeping was wrong -- the total then fails the
    // limit and *miss is set: the sort's kernels drop what does not fit its cell, nothing is written out of bounds as long as
    // limit <= the buffer, because cell_base[CELL_COUNT] is what every range is clipped to)
    const u64 total = cells_walk(
        status, tag,
        [&](u32 c) {
            const u32 g = c - prev_g0;  // (wraps for c < prev_g0: then g >= prev_groups)
            const u32 gi = g < prev_groups ? g : 0u;  // (the load is unconditional: prev_bounds has at least two words)
            return make_uint4(prev_bounds[gi], prev_bounds[gi + 1u], g < prev_groups ? 1u : 0u, 0u);
        },
        [&](u32, uint4 r) { return cell_capacity(min(r.z ? r.y - r.x : 0u, 1u << 16)); },
        [&](u32 c, u32 at, u32, uint4) {
            cell_base[c] = at;
            line_cur[c] = 0u;
            frag_cur[c] = 0u;
        });
    if (blockIdx.x == CELL_WGS - 1 && threadIdx.x == 0u) {
        const bool fits = total <= limit && total < 0xFFFFFFFFull;
        cell_base[CELL_COUNT] = (u32)(fits ? total : 0ull);
        *miss = fits ? 0u : 1u;
    }
}
__global__ __launch_bounds__(CELL_THREADS) void cell_layout_kernel(const u32* __restrict__ prev_bounds, u32 prev_g0, u32 prev_groups,
                                                                   u32* __restrict__ cell_base, u32* __restrict__ line_cur,
                                                                   u32* __restrict__ frag_cur, u32* __restrict__ miss, u64 limit,
                                                                   u64* status, u32 tag) {
    cell_layout_body(prev_bounds, prev_g0, prev_groups, cell_base, line_cur, frag_cur, miss, limit, status, tag);
}
// the sort's first kernel and the layout of its cells in one launch (the two are independent of each other and both tiny: a
// launch costs ~5 us whatever it does): CELL_WGS workgroups, the first n_passes of which also do radix_prepare_kernel's work
static_assert(CELL_THREADS == RS_BINS, "the fused launch runs both bodies with one workgroup shape");
__global__ __launch_bounds__(RS_BINS) void radix_prepare_layout_kernel(int n_passes, const u64* __restrict__ copies, int n_copies, u64* __restrict__ hist,
                                                                       u64* __restrict__ bin_base, u32* __restrict__ counters, DevScalars* sc,
                                                                       DevScalars sc_init, int set_sc, u32* __restrict__ cursors, ScanFinalize fin,
                                                                       CellRoute cr) {
    if ((int)blockIdx.x < n_passes) radix_prepare_body(copies, n_copies, hist, bin_base, counters, sc, sc_init, set_sc, cursors, fin);  // (workgroup-uniform)
    cell_layout_body(cr.layout_prev_bounds, cr.layout_g0, cr.layout_groups, (u32*)cr.cell_base, cr.line_cur, cr.frag_cur, cr.miss, (u64)cr.layout_limit,
                     (u64*)cr.status, cr.tag);
}
hipError_t launch_cell_layout(const uint32_t* prev_bounds, uint32_t prev_g0, uint32_t prev_groups, uint32_t* cell_base, uint32_t* line_cur,
                              uint32_t* frag_cur, uint32_t* miss, size_t limit, unsigned long long* status, uint32_t tag, hipStream_t stream) {
    if (prev_groups == 0) return hipErrorInvalidValue;
    cell_layout_kernel<<<CELL_WGS, CELL_THREADS, 0, stream>>>(prev_bounds, prev_g0, prev_groups, cell_base, line_cur, frag_cur, miss, (u64)limit,
                                                              (u64*)status, tag);
    return hipGetLastError();
}
__global__ __launch_bounds__(CELL_THREADS) void cell_count_kernel(CellRoute cr, u32 g0, u32 n_groups, u32* __restrict__ bounds, u32 cap,
                                                                  u32* __restrict__ stats, const u32* __restrict__ dev_sizes, u64* status, u32 tag,
                                                                  u32* err_word, u32 err_bit, u32 want_total) {
    __shared__ u32 s_mx[CELL_THREADS / 64], s_ov[CELL_THREADS / 64], s_bad[CELL_THREADS / 64];
    const u32* __restrict__ const lc = cr.line_cur;
    const u32* __restrict__ const fc = cr.frag_cur;
    const u32* __restrict__ const cb = cr.cell_base;
    uint4* __restrict__ const srcs = cr.src;
    const u32 t = threadIdx.x;
    const u32 limit = cb[CELL_COUNT];  // (every cell's range is clipped to the layout's total: a failed layout leaves 0)
    u32 mx = 0, over = 0, bad = 0;
    const u64 total = cells_walk(
        status, tag,
        [&](u32 c) { return make_uint4(lc[c], fc[c], cb[c], cb[c + 1u]); },  // {lines, fragment records, start, end}
        [&](u32 c, uint4 r) {
            const u32 cnt = r.x * (u32)RC_LINE + r.y;
            const bool in_range = c >= g0 && c - g0 < n_groups;
            // the two fronts of the cell ran into each other / no finish kernel could take it / a first id outside the range the
            // caller vouched for / the layout failed
            bad |= (r.w < r.z || cnt > r.w - r.z || r.w > limit || cnt >= 65536u || (cnt != 0u && !in_range)) ? 1u : 0u;
            mx = max(mx, cnt);
            over += cnt > cap ? cnt : 0u;
            return cnt;
        },
        [&](u32 c, u32 at, u32 cnt, uint4 r) {
            if (c >= g0 && c - g0 <= n_groups) bounds[c - g0] = at;  // (bounds[n_groups]: the total -- the cells behind the groups are empty)
            // everything the finish kernel needs of group c - g0 in one 16-byte word: where its whole lines start, where the
            // records claimed downwards start, its offset in the dense output, its size and how much of it is whole lines
            // (a cell that is not sound -- overflowed, or laid out by a failed layout -- is handed on as EMPTY: the finish kernel runs
            // before the host has seen the miss, and must not read where the cell claims to be)
            if (c >= g0 && c - g0 < n_groups) {
                const bool sound = r.w >= r.z && cnt <= r.w - r.z && r.w <= limit && cnt < 65536u;
                srcs[c - g0] = sound ? make_uint4(r.z, r.w - r.y, at, ((cnt & 0xFFFFu) << 16) | ((cnt - r.y) & 0xFFFFu)) : make_uint4(0u, 0u, at, 0u);
            }
        });
    for (int o = 32; o; o >>= 1) {
        mx = max(mx, (u32)__shfl_xor((int)mx, o, 64)); over += (u32)__shfl_xor((int)over, o, 64); bad |= (u32)__shfl_xor((int)bad, o, 64);
    }
    if (lane_id() == 0) { s_mx[t >> 6] = mx; s_ov[t >> 6] = over; s_bad[t >> 6] = bad; }
    __syncthreads();
    if (t == 0u) {
        for (int k = 1; k < CELL_THREADS / 64; ++k) { mx = max(mx, s_mx[k]); over += s_ov[k]; bad |= s_bad[k]; }
        if (stats) {
            if (mx) atomicMax(&stats[0], mx);
            if (over) atomicAdd(&stats[1], over);
        }
        if (blockIdx.x == CELL_WGS - 1) {
            if (g0 + n_groups == CELL_COUNT) bounds[n_groups] = (u32)total;
            if (total >= 0xFFFFFFFFull || (dev_sizes && (u32)total != dev_sizes[SIZES_N]) || (want_total && (u32)total != want_total)) bad = 1u;
        }
        if (bad) *cr.miss = 1u;
        // (a caller that reads an error word instead of the miss word: set there too, also for what the layout and the sort flagged)
        if (err_word && (bad || (blockIdx.x == 0 && *cr.miss))) atomicOr(err_word, err_bit);
    }
}
hipError_t launch_cell_count(const CellRoute& cells, uint32_t g0, uint32_t n_groups, uint32_t* bounds, uint32_t cap, uint32_t* stats,
                             const uint32_t* dev_sizes, hipStream_t stream, uint32_t* err_word, uint32_t err_bit, uint32_t want_total) {
    if ((size_t)g0 + n_groups > CELL_COUNT) return hipErrorInvalidValue;
    cell_count_kernel<<<CELL_WGS, CELL_THREADS, 0, stream>>>(cells, g0, n_groups, bounds, cap, stats, dev_sizes, (u64*)cells.status, cells.tag + 1u,
                                                             err_word, err_bit, want_total);
    return hipGetLastError();
}

template <typename K, typename V, typename DigitFn>
static hipError_t launch_onesweep(const K* ki, K* ko, const V* vi, V* vo, size_t n, DigitFn fn, const u64* bb, u64* st,
                                  u32* tc, DevScalars* sc, hipStream_t stream, const uint8_t* bi = nullptr, uint8_t* bo = nullptr,
                                  const u32* blockfill = nullptr, size_t n_out = 0, bool stable = true, u32 epoch = 0, bool few = false,
                                  const u32* dev_sizes = nullptr) {
    // n: extent of the input (tiles are cut from it); n_out: records that exist (== n without holes)
    const unsigned tiles = (unsigned)rs_num_tiles(n);
    if (!blockfill) n_out = n;
    if constexpr (!std::is_same<DigitFn, ShiftDigit>::value) {  // the bucketing passes (few buckets); radix digits never take this route
        if (few) {
            radix_onesweep_kernel<K, V, RS_THREADS_DEF, RS_ITEMS_DEF, DigitFn, true, true><<<tiles, RS_THREADS_DEF, 0, stream>>>(
                ki, ko, vi, vo, n, fn, bb, st, tc, sc, bi, bo, blockfill, n_out, epoch, dev_sizes);
            return hipGetLastError();
        }
    }
    if (stable)
        radix_onesweep_kernel<K, V, RS_THREADS_DEF, RS_ITEMS_DEF, DigitFn, true><<<tiles, RS_THREADS_DEF, 0, stream>>>(
            ki, ko, vi, vo, n, fn, bb, st, tc, sc, bi, bo, blockfill, n_out, epoch, dev_sizes);
    else
        radix_onesweep_kernel<K, V, RS_THREADS_DEF, RS_ITEMS_DEF, DigitFn, false><<<tiles, RS_THREADS_DEF, 0, stream>>>(
            ki, ko, vi, vo, n, fn, bb, st, tc, sc, bi, bo, blockfill, n_out, epoch, dev_sizes);
    return hipGetLastError();
}

bool radix_claim_applies(size_t extent) {
    const char* const claim_env = getenv("BP_RADIX_CLAIM");
    const char* const claim_min_env = getenv("BP_RADIX_CLAIM_MIN_TILES");
    const bool claim_off = claim_env != nullptr && atoi(claim_env) == 0;
    const size_t claim_min_tiles = claim_min_env ? (size_t)atoll(claim_min_env) : 256;
    return !claim_off && rs_num_tiles(extent) >= claim_min_tiles;
}

hipError_t radix_sort(uint64_t* keys0, uint64_t* keys1, uint32_t* vals0, uint32_t* vals1, size_t n,
                      const RadixPlan& plan, void* workspace, size_t workspace_bytes, DevScalars* sc,
                      int* result_buf, hipStream_t stream, void (*mark)(void*, int), void* mark_ctx,
                      int class_hist, int class_sort, const uint8_t* box_in, uint8_t* box_out,
                      const uint32_t* blockfill, size_t extent, const unsigned long long* ready_hist, int ready_copies,
                      int unstable_first, uint32_t* epoch_io, const DevScalars* sc_init, const uint32_t* dev_sizes, const ScanFinalize* finalize,
                      const CellRoute* cells) {
    if (dev_sizes && (!ready_hist || vals0)) return hipErrorInvalidValue;  // (the counting kernel and the payload passes take host sizes)
    if (finalize && (!dev_sizes || finalize->sizes != dev_sizes)) return hipErrorInvalidValue;
    static const bool no_unstable = getenv("BP_NO_UNSTABLE_PASS") != nullptr;  // A/B switch
    if (no_unstable) unstable_first = 0;
    *result_buf = 0;
    const int np = plan.n_passes;
    if (n == 0 || np == 0) return hipSuccess;
    if (!blockfill) extent = n;
    if (extent < n || (blockfill && vals0)) return hipErrorInvalidValue;
    if (extent >= 0xFFFFFFFFull) return hipErrorInvalidValue;
    if (radix_workspace_bytes(extent, np) > workspace_bytes) return hipErrorInvalidValue;
    unsigned char* ws = (unsigned char*)workspace;
    if (np > RS_MAX_PASSES) return hipErrorInvalidValue;
    u64* hist = (u64*)ws;
    u64* bin_base = hist + (size_t)RS_MAX_PASSES * RS_BINS;
    u32* counters = (u32*)(bin_base + (size_t)RS_MAX_PASSES * RS_BINS);
    u32* cursors = (u32*)((unsigned char*)counters + align_up((size_t)RS_MAX_PASSES * RS_TICKETS * sizeof(u32), 256));
    u64* status = (u64*)(ws + rs_header_bytes());
    const size_t tiles = rs_num_tiles(extent);
    hipError_t e;
    // the first pass of a sort that a finishing kernel completes: radix_claim_kernel (line-aligned stores, no chain) when the
    // input is large enough for its persistent workgroups to amortise their final partial lines; BP_RADIX_CLAIM=0: off
    // (the two switches are read per sort, so that the tests can force the kernel onto small inputs)
    const bool claim = radix_claim_applies(extent) && unstable_first && !plan.from_val[0] && !(np == 1 && box_out);
    if (cells && (np != 2 || !claim || !ready_hist || plan.from_val[1] || plan.shift[1] != plan.shift[0] + 8 || (vals0 && !cells->out_vals) ||
                  box_out || (dev_sizes && vals0)))
        return hipErrorInvalidValue;

    if (mark) mark(mark_ctx, class_hist);
    // status words: tagged per launch when the caller keeps an epoch counter for this workspace (epoch_io; the words of
    // earlier launches then need no zero-fill), zero-filled otherwise
    u32 epoch0 = 0;
    if (epoch_io) {
        if (*epoch_io + (u32)np >= ST_EPOCH_MASK) {  // tags wrap: start over on a clean workspace
            e = hipMemsetAsync(workspace, 0, workspace_bytes, stream);
            if (e != hipSuccess) return e;
            *epoch_io = 0;
        }
        epoch0 = *epoch_io + 1u;
        *epoch_io += (u32)np;
    } else {
        e = hipMemsetAsync(workspace, 0, radix_workspace_bytes(extent, np), stream);
        if (e != hipSuccess) return e;
    }
    DevScalars init = {};
    if (sc_init) init = *sc_init;
    if (cells && cells->layout_prev_bounds && !ready_hist) return hipErrorInvalidValue;  // (the folded layout lives in the ready-histogram kernel)
    if (ready_hist) {  // the producer of the records already counted the digits of this plan
        if (cells && cells->layout_prev_bounds)
            radix_prepare_layout_kernel<<<CELL_WGS, RS_BINS, 0, stream>>>(np, (const u64*)ready_hist, ready_copies, hist, bin_base, counters, sc, init,
                                                                          sc_init ? 1 : 0, claim ? cursors : nullptr,
                                                                          finalize ? *finalize : ScanFinalize{}, *cells);
        else
            radix_prepare_kernel<<<np, RS_BINS, 0, stream>>>((const u64*)ready_hist, ready_copies, hist, bin_base, counters, sc, init, sc_init ? 1 : 0,
                                                             claim ? cursors : nullptr, finalize ? *finalize : ScanFinalize{});
    } else {
        if (epoch_io) {  // the counting kernel adds into the histogram
            e = hipMemsetAsync(hist, 0, (size_t)np * RS_BINS * sizeof(u64), stream);
            if (e != hipSuccess) return e;
        }
        static const int hist_blocks_env = getenv("BP_HIST_BLOCKS") ? atoi(getenv("BP_HIST_BLOCKS")) : 0;  // measurement knob
        radix_hist_kernel<<<grid_for(extent, HIST_THREADS * 4, hist_blocks_env > 0 ? (unsigned)hist_blocks_env : (unsigned)HIST_MAX_BLOCKS), HIST_THREADS, 0, stream>>>(
            (const u64*)keys0, vals0, extent, plan, hist, blockfill);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        radix_prepare_kernel<<<np, RS_BINS, 0, stream>>>(nullptr, 0, hist, bin_base, counters, sc, init, sc_init ? 1 : 0,
                                                         claim ? cursors : nullptr);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (mark) mark(mark_ctx, -1);

    u64* kb[2] = {(u64*)keys0, (u64*)keys1};
    u32* vb[2] = {vals0, vals1};
    int cur = 0;
    int first_key_pass = -1;
    for (int p = 0; p < np && first_key_pass < 0; ++p)
        if (!plan.from_val[p]) first_key_pass = p;
    if (box_in && first_key_pass != 0) return hipErrorInvalidValue;                   // must be gathered before any permutation
    if (box_out && (first_key_pass < 0 || plan.from_val[np - 1])) return hipErrorInvalidValue;  // no key pass to split them off
    if (cells) {
        // the claiming pass by the HIGH digit, then the claiming pass that spreads every high bin over its cells
        const bool holes = blockfill != nullptr;
        ClaimArgs a{};
        a.keys_in = kb[0]; a.keys_out = kb[1];
        a.vals_in = vals0 ? vb[0] : nullptr; a.vals_out = vals0 ? vb[1] : nullptr;
        a.n = holes ? extent : n; a.n_out = n;
        a.shift = plan.shift[1]; a.bin_base = bin_base + RS_BINS;
        a.line_cur = cursors; a.frag_cur = cursors + SW_CURSOR_WORDS;
        a.blockfill = blockfill; a.box_in = box_in;
        a.n_tiles = (u32)rs_num_tiles(a.n);
        a.dev_sizes = dev_sizes;
        if (mark) mark(mark_ctx, class_sort);
        const bool tree_words = !vals0 && cells->out_keys != nullptr;  // (the tree's compact records: a kernel name of their own)
        e = vals0 ? launch_claim<u32>(a, stream) : tree_words ? launch_claim<NoVal, false, 1>(a, stream) : launch_claim<NoVal>(a, stream);
        if (mark) mark(mark_ctx, -1);
        if (e != hipSuccess) return e;
        // the gapped result: in keys0 (key-only sorts) or in the buffers the caller set aside for it
        u64* const gk = cells->out_keys ? (u64*)cells->out_keys : kb[0];
        ClaimArgs b{};
        b.keys_in = kb[1]; b.keys_out = gk;
        b.vals_in = vals0 ? vb[1] : nullptr; b.vals_out = vals0 ? cells->out_vals : nullptr;
        b.n = n; b.n_out = n;
        b.shift = plan.shift[0]; b.hi_shift = plan.shift[1];
        b.n_tiles = (u32)rs_num_tiles(n);
        b.dev_sizes = dev_sizes;
        b.cell_base = cells->cell_base; b.cell_line_cur = cells->line_cur; b.cell_frag_cur = cells->frag_cur; b.miss = cells->miss;
        if (((size_t)gk & 127u) != 0) return hipErrorInvalidValue;  // cell bases are whole lines of a line-aligned buffer
        if (mark) mark(mark_ctx, class_sort);
        e = vals0 ? launch_claim<u32, true>(b, stream) : tree_words ? launch_claim<NoVal, true, 1>(b, stream) : launch_claim<NoVal, true>(b, stream);
        if (mark) mark(mark_ctx, -1);
        if (e != hipSuccess) return e;
        *result_buf = 0;  // (key-only: keys0; with a payload: cells->out_keys / out_vals)
        return hipSuccess;
    }
    for (int p = 0; p < np; ++p) {
        if (mark) mark(mark_ctx, class_sort);  // one timed region per pass = per kernel launch
        const u64* bb = bin_base + (size_t)p * RS_BINS;
        u64* st = status + (size_t)p * tiles * RS_BINS;
        u32* tc = counters + (size_t)p * RS_TICKETS;
        if (claim && p == 0) {
            const bool holes = blockfill != nullptr;
            ClaimArgs a{};
            a.keys_in = kb[cur]; a.keys_out = kb[cur ^ 1];
            a.vals_in = vals0 ? vb[cur] : nullptr; a.vals_out = vals0 ? vb[cur ^ 1] : nullptr;
            a.n = holes ? extent : n; a.n_out = n;
            a.shift = plan.shift[p]; a.bin_base = bb;
            a.line_cur = cursors; a.frag_cur = cursors + SW_CURSOR_WORDS;
            a.box_in = box_in; a.blockfill = blockfill;
            a.n_tiles = (u32)rs_num_tiles(a.n);
            a.dev_sizes = dev_sizes;
            e = vals0 ? launch_claim<u32>(a, stream) : launch_claim<NoVal>(a, stream);
        } else if (!vals0) {
            const bool holes = blockfill && p == 0;  // only the first pass sees the block-structured input
            e = launch_onesweep<u64, NoVal>(kb[cur], kb[cur ^ 1], (const NoVal*)nullptr, (NoVal*)nullptr, holes ? extent : n,
                                            ShiftDigit{plan.shift[p]}, bb, st, tc, sc, stream, nullptr, nullptr,
                                            holes ? blockfill : nullptr, n, !(unstable_first && p == 0), epoch0 ? epoch0 + (u32)p : 0u, false,
                                            dev_sizes);
        } else if (plan.from_val[p]) {
            e = launch_onesweep<u32, u64>(vb[cur], vb[cur ^ 1], kb[cur], kb[cur ^ 1], n, ShiftDigit{plan.shift[p]}, bb, st, tc, sc, stream,
                                          nullptr, nullptr, nullptr, 0, true, epoch0 ? epoch0 + (u32)p : 0u);
        } else {
            // box_in (bytes in the records' CURRENT order) can only be gathered by the very first pass; the codes are
            // split off again by the last pass, which must be a key pass
            e = launch_onesweep<u64, u32>(kb[cur], kb[cur ^ 1], vb[cur], vb[cur ^ 1], n, ShiftDigit{plan.shift[p]}, bb, st, tc, sc, stream,
                                          p == 0 ? box_in : nullptr, p == np - 1 ? box_out : nullptr, nullptr, 0,
                                          !(unstable_first && p == 0), epoch0 ? epoch0 + (u32)p : 0u);
        }
        if (e != hipSuccess) return e;
        if (mark) mark(mark_ctx, -1);
        cur ^= 1;
    }
    *result_buf = cur;
    return hipSuccess;
}

// one stable bucketing pass (<= 256 buckets) with an arbitrary digit functor; hist[256] (device,
// inside the workspace) holds the bucket sizes afterwards
template <typename V, typename DigitFn>
static hipError_t partition_pass(const u64* ki, u64* ko, const V* vi, V* vo, size_t n, DigitFn fn, void* workspace,
                                 size_t workspace_bytes, DevScalars* sc, const u64** d_hist, hipStream_t stream,
                                 const uint8_t* box_in = nullptr, const u32* blockfill = nullptr, size_t n_out = 0,
                                 const u64* ready_hist = nullptr) {
    // blockfill: ki is block-structured (n = extent, n_out = records that exist); the output is dense
    if (n >= 0xFFFFFFFFull) return hipErrorInvalidValue;
    if (radix_workspace_bytes(n, 1) > workspace_bytes) return hipErrorInvalidValue;
    unsigned char* ws = (unsigned char*)workspace;
    u64* hist = (u64*)ws;
    u64* bin_base = hist + RS_BINS;
    u32* counters = (u32*)(bin_base + RS_BINS);
    u64* status = (u64*)((unsigned char*)counters + align_up(sizeof(u32), 256));
    hipError_t e = hipMemsetAsync(workspace, 0, radix_workspace_bytes(n, 1), stream);
    if (e != hipSuccess) return e;
    *d_hist = hist;
    if (n == 0) return hipSuccess;
    if (ready_hist) {  // the producer of the records counted the buckets already
        e = hipMemcpyAsync(hist, ready_hist, RS_BINS * sizeof(u64), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
    } else {
        digit_hist_kernel<DigitFn><<<grid_for(n, HIST_THREADS * 4, HIST_MAX_BLOCKS), HIST_THREADS, 0, stream>>>(ki, n, fn, hist, blockfill);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    radix_binbase_kernel<<<1, RS_BINS, 0, stream>>>(hist, bin_base);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    static const bool no_few = getenv("BP_NO_FEW_RANKING") != nullptr;  // A/B switch
    return launch_onesweep<u64, V, DigitFn>(ki, ko, vi, vo, n, fn, bin_base, status, counters, sc, stream, box_in, nullptr, blockfill, n_out,
                                            true, 0, !no_few);
}

hipError_t partition_entries_known(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* ids_in, uint32_t* ids_out, size_t m,
                                   const unsigned char* host_lut, uint32_t n_cells, unsigned char* d_lut, const unsigned long long* host_hist,
                                   int cell_shift, uint32_t depth_mask, uint32_t split_depth, uint32_t coarse_bucket, void* workspace,
                                   size_t workspace_bytes, DevScalars* sc, const DevScalars& sc_init, const unsigned long long** d_hist,
                                   hipStream_t stream, const uint8_t* box_in) {
    if (m == 0 || m >= 0xFFFFFFFFull || n_cells == 0 || n_cells > PART_SETUP_MAX_CELLS) return hipErrorInvalidValue;
    if (radix_workspace_bytes(m, 1) > workspace_bytes) return hipErrorInvalidValue;
    unsigned char* ws = (unsigned char*)workspace;
    u64* hist = (u64*)ws;
    u64* bin_base = hist + RS_BINS;
    u32* counters = (u32*)(bin_base + RS_BINS);
    u64* status = (u64*)((unsigned char*)counters + align_up(sizeof(u32), 256));
    PartSetupArgs a;
    for (uint32_t c = 0; c < n_cells; ++c) a.lut[c] = host_lut[c];
    for (int k = 0; k < RS_BINS; ++k) a.hist[k] = host_hist[k];
    partition_setup_kernel<<<64, RS_BINS, 0, stream>>>(a, n_cells, d_lut, (u64*)ws, radix_workspace_bytes(m, 1) / sizeof(u64), hist, bin_base, sc, sc_init);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    *d_hist = hist;
    EntryDestDigit fn{d_lut, cell_shift, depth_mask, split_depth, coarse_bucket};
    static const bool no_few = getenv("BP_NO_FEW_RANKING") != nullptr;  // A/B switch
    return launch_onesweep<u64, u32, EntryDestDigit>((const u64*)keys_in, (u64*)keys_out, ids_in, ids_out, m, fn, bin_base, status, counters, sc, stream,
                                                     box_in, nullptr, nullptr, m, true, 0, !no_few);
}

hipError_t partition_entries(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* ids_in, uint32_t* ids_out, size_t m,
                             const unsigned char* d_lut, int cell_shift, uint32_t depth_mask, uint32_t split_depth,
                             uint32_t coarse_bucket, void* workspace, size_t workspace_bytes, DevScalars* sc,
                             const unsigned long long** d_hist, hipStream_t stream, const uint8_t* box_in,
                             const unsigned long long* ready_hist) {
    EntryDestDigit fn{d_lut, cell_shift, depth_mask, split_depth, coarse_bucket};
    return partition_pass<u32, EntryDestDigit>((const u64*)keys_in, (u64*)keys_out, ids_in, ids_out, m, fn, workspace,
                                               workspace_bytes, sc, (const u64**)d_hist, stream, box_in, nullptr, 0,
                                               (const u64*)ready_hist);
}

__global__ __launch_bounds__(256) void split_box_kernel(u32* __restrict__ ids, uint8_t* __restrict__ box, size_t m) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256) {
        const u32 v = ids[i];
        box[i] = (uint8_t)box_from_packed(v >> BOX_ID_BITS);
        ids[i] = v & BOX_ID_MASK;
    }
}
__global__ __launch_bounds__(256) void pack_box_kernel(u32* __restrict__ ids, const uint8_t* __restrict__ box, size_t m) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256)
        ids[i] = (ids[i] & BOX_ID_MASK) | (box_to_packed(box[i]) << BOX_ID_BITS);
}
hipError_t launch_pack_box(uint32_t* ids, const uint8_t* box, size_t m, hipStream_t stream) {
    if (m == 0) return hipSuccess;
    pack_box_kernel<<<grid_for(m, 256, 4096), 256, 0, stream>>>(ids, box, m);
    return hipGetLastError();
}
hipError_t launch_split_box(uint32_t* ids, uint8_t* box, size_t m, hipStream_t stream) {
    if (m == 0) return hipSuccess;
    split_box_kernel<<<grid_for(m, 256, 4096), 256, 0, stream>>>(ids, box, m);
    return hipGetLastError();
}

hipError_t partition_pairs(const uint64_t* in, uint64_t* out, size_t n, const uint32_t* bounds, int n_bounds, void* workspace,
                           size_t workspace_bytes, DevScalars* sc, const unsigned long long** d_hist, hipStream_t stream,
                           const uint32_t* blockfill, size_t n_out, const unsigned long long* ready_hist) {
    PairOwnerDigit fn;
    fn.n = n_bounds;
    for (int i = 0; i < 15; ++i) fn.bounds[i] = i < n_bounds ? bounds[i] : 0xFFFFFFFFu;
    return partition_pass<NoVal, PairOwnerDigit>((const u64*)in, (u64*)out, (const NoVal*)nullptr, (NoVal*)nullptr, n, fn, workspace,
                                                 workspace_bytes, sc, (const u64**)d_hist, stream, nullptr, blockfill, n_out,
                                                 (const u64*)ready_hist);
}

// out[k] = sum over the copies of copies[c * n + k]
__global__ __launch_bounds__(256) void sum_copies_kernel(const u64* __restrict__ copies, int n_copies, u32 n, u64* __restrict__ out) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    u64 t = 0;
#pragma unroll 8
    for (int c = 0; c < n_copies; ++c) t += copies[(size_t)c * n + k];
    out[k] = t;
}
hipError_t launch_sum_copies(const unsigned long long* copies, int n_copies, uint32_t n, unsigned long long* out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    sum_copies_kernel<<<(n + 255) / 256, 256, 0, stream>>>((const u64*)copies, n_copies, n, (u64*)out);
    return hipGetLastError();
}

hipError_t launch_cell_hist(const uint64_t* keys, size_t m, int cell_shift, uint32_t depth_mask, uint32_t split_depth,
                            uint32_t n_cells, unsigned long long* d_hist, hipStream_t stream) {
    if (m == 0) return hipSuccess;
    cell_hist_kernel<<<grid_for(m, 256 * 8, 512), 256, (n_cells + 1) * sizeof(u32), stream>>>((const u64*)keys, m, cell_shift, depth_mask,
                                                                                              split_depth, n_cells, (u64*)d_hist);
    return hipGetLastError();
}

hipError_t launch_scan_skipflags(int kind, ScanParams p, hipStream_t stream) {
    if (p.m == 0) return hipSuccess;
    const unsigned g = (unsigned)scan_blocks(p.m);
    BP_DISPATCH_KIND(kind, (scan_skipflags_kernel<K><<<g, SCAN_THREADS, 0, stream>>>(p)));
    return hipGetLastError();
}

hipError_t launch_scan_count(int kind, ScanParams p, uint32_t* counts, unsigned long long* block_sums,
                             hipStream_t stream) {
    if (p.m == 0) return hipSuccess;
    const unsigned g = (unsigned)scan_blocks(p.m);
    if (p.general) {
        BP_DISPATCH_KIND(kind, (scan_count_kernel<K, true><<<g, SCAN_THREADS, 0, stream>>>(p, counts, block_sums)));
    } else {
        BP_DISPATCH_KIND(kind, (scan_count_kernel<K, false><<<g, SCAN_THREADS, 0, stream>>>(p, counts, block_sums)));
    }
    return hipGetLastError();
}

hipError_t launch_scan_emit(int kind, ScanParams p, const uint32_t* counts, const unsigned long long* block_offs,
                            uint64_t* pairs_out, hipStream_t stream) {
    if (p.m == 0) return hipSuccess;
    const unsigned g = (unsigned)scan_blocks(p.m);
    if (p.general) {
        BP_DISPATCH_KIND(kind, (scan_emit_kernel<K, true><<<g, SCAN_THREADS, 0, stream>>>(p, counts, block_offs, (u64*)pairs_out)));
    } else {
        BP_DISPATCH_KIND(kind, (scan_emit_kernel<K, false><<<g, SCAN_THREADS, 0, stream>>>(p, counts, block_offs, (u64*)pairs_out)));
    }
    return hipGetLastError();
}

hipError_t launch_scan_fused(int kind, ScanParams p, uint64_t* pairs_out, size_t out_cap, DevScalars* sc, hipStream_t stream) {
    if (p.m == 0) return hipSuccess;
    const unsigned g = (unsigned)scan_blocks(p.m);
    if (p.filt_bounds) {
        BP_DISPATCH_KIND(kind, (scan_fused_kernel<K, true><<<g, SCAN_THREADS, 0, stream>>>(p, (u64*)pairs_out, (u64)out_cap, sc)));
    } else {
        BP_DISPATCH_KIND(kind, (scan_fused_kernel<K, false><<<g, SCAN_THREADS, 0, stream>>>(p, (u64*)pairs_out, (u64)out_cap, sc)));
    }
    return hipGetLastError();
}

hipError_t launch_entry_ctx(int kind, const uint64_t* keys, const uint32_t* ids, const uint8_t* box, size_t m,
                            uint32_t inactive_below, uint64_t* idctx, hipStream_t stream, uint32_t active_lo, uint32_t active_hi) {
    if (m == 0) return hipSuccess;
    const unsigned g = (unsigned)((m + 255) / 256);
    switch (kind) {
        case 0: entry_ctx_kernel<0><<<g, 256, 0, stream>>>((const u64*)keys, ids, box, m, inactive_below, (u64*)idctx, active_lo, active_hi); break;
        case 1: entry_ctx_kernel<1><<<g, 256, 0, stream>>>((const u64*)keys, ids, box, m, inactive_below, (u64*)idctx, active_lo, active_hi); break;
        default: entry_ctx_kernel<2><<<g, 256, 0, stream>>>((const u64*)keys, ids, box, m, inactive_below, (u64*)idctx, active_lo, active_hi); break;
    }
    return hipGetLastError();
}

// one launch instead of a scalar upload and three memsets in front of every scan: neutral reduction scalars,
// zeroed per-block fill counts, shard cursors / counters and (optionally) digit histograms
__global__ __launch_bounds__(256) void scan_prepare_kernel(DevScalars* sc, DevScalars init, u32* __restrict__ blockfill, size_t n_blocks,
                                                           u64* __restrict__ shards, u32 n_shard_words, u64* __restrict__ hist,
                                                           u32 n_hist_words) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (t == 0) *sc = init;
    for (size_t i = t; i < n_blocks; i += stride) blockfill[i] = 0u;
    for (size_t i = t; i < n_shard_words; i += stride) shards[i] = 0ull;
    for (size_t i = t; i < n_hist_words; i += stride) hist[i] = 0ull;
}
hipError_t launch_scan_prepare(DevScalars* sc, const DevScalars& init, uint32_t* blockfill, size_t n_blocks, uint64_t* shards,
                               uint32_t n_shard_words, unsigned long long* hist, uint32_t n_hist_words, hipStream_t stream) {
    const unsigned grid = (unsigned)std::min<size_t>(2048, (std::max<size_t>(n_blocks, 1024) + 255) / 256);
    scan_prepare_kernel<<<grid, 256, 0, stream>>>(sc, init, blockfill, n_blocks, (u64*)shards, n_shard_words, (u64*)hist,
                                                  hist ? n_hist_words : 0u);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void scan_finalize_kernel(const u64* __restrict__ shards, u64 out_blocks, u64 n_bound, u64 extent_bound,
                                                           const DevScalars* __restrict__ sc, u32* __restrict__ sizes) {
    scan_finalize_wave(shards, out_blocks, n_bound, extent_bound, sc, sizes);
}
hipError_t launch_scan_finalize(const uint64_t* shards, size_t out_blocks, size_t n_bound, size_t extent_bound, const DevScalars* sc,
                                uint32_t* sizes, hipStream_t stream) {
    scan_finalize_kernel<<<1, 64, 0, stream>>>((const u64*)shards, (u64)out_blocks, (u64)n_bound, (u64)extent_bound, sc, sizes);
    return hipGetLastError();
}

template <int KIND, bool FILT, bool OWNER, int GEN = 0, bool OCC6 = false, bool PRECNT = false>
static hipError_t launch_scan_expand_variant(ExpandParams p, size_t max_grid, hipStream_t stream) {
    // resident workgroups per dynamic-LDS size (0 .. SX_HIST_MAX rows of counters)
    static unsigned resident[SX_HIST_MAX + 1] = {};
    const int rows = sx_hist_rows(p);
    if (rows < 0 || rows > SX_HIST_MAX) return hipErrorInvalidValue;
    if (!resident[rows]) resident[rows] = resident_blocks(scan_expand_kernel<KIND, FILT, OWNER, GEN, OCC6, PRECNT>, SX_THREADS, sx_hist_bytes(p));
    constexpr int occ = (FILT || OCC6) ? (SX_WAVES_PER_EU < 6 ? SX_WAVES_PER_EU : 6) : SX_WAVES_PER_EU;
    unsigned res = std::min(resident[rows], resident_cap(occ ? occ * 4 / SX_WAVES : 64));
    // small trees: a wave should have a dozen chunks to amortise its prologue, its last partial block and the counters' flush
    // (1 M objects: 1536 workgroups 0.421 ms per frame, 1024 0.412, 768 0.414, 512 0.428); never fewer workgroups than CUs
    {
        const size_t chunks = (p.m + 63) / 64;
        res = (unsigned)std::min<size_t>(res, std::max<size_t>(resident_cap(1), chunks / ((size_t)SX_WAVES * 12)));
    }
    if (const char* e = getenv("BP_SX_GRID")) res = (unsigned)atoi(e);  // (measurement knob: workgroups of the persistent grid)
    const unsigned grid = (unsigned)std::min<size_t>(max_grid, res);
    p.n_waves = grid * SX_WAVES;
    scan_expand_kernel<KIND, FILT, OWNER, GEN, OCC6, PRECNT><<<grid, SX_THREADS, sx_hist_bytes(p), stream>>>(p);
    if (!p.big_inline) scan_expand_big_kernel<KIND, FILT, OWNER, GEN><<<SXB_BLOCKS, SX_THREADS, 0, stream>>>(p);
    return hipGetLastError();
}
template <int KIND, bool PRECNT>
static hipError_t launch_scan_expand_gen0(ExpandParams p, size_t max_grid, hipStream_t stream, bool occ6) {
    const bool owner = p.owner_n >= 0;  // the owner-counting variant is a separate instantiation: the default kernel pays nothing
    if (p.filt_bounds) return owner ? launch_scan_expand_variant<KIND, true, true, 0, false, PRECNT>(p, max_grid, stream)
                                    : launch_scan_expand_variant<KIND, true, false, 0, false, PRECNT>(p, max_grid, stream);
    if (owner) return launch_scan_expand_variant<KIND, false, true, 0, false, PRECNT>(p, max_grid, stream);
    return occ6 ? launch_scan_expand_variant<KIND, false, false, 0, true, PRECNT>(p, max_grid, stream)
                : launch_scan_expand_variant<KIND, false, false, 0, false, PRECNT>(p, max_grid, stream);
}
template <int KIND>
static hipError_t launch_scan_expand_kind(ExpandParams p, size_t max_grid, hipStream_t stream, int gen, bool occ6) {
    // (the passes for ids that may repeat mark and re-read the context words of a tree no finish has counted: ranges searched)
    if (gen == 1) { p.dcnt = nullptr; return launch_scan_expand_variant<KIND, false, false, 1>(p, max_grid, stream); }
    if (gen == 2) { p.dcnt = nullptr; return launch_scan_expand_variant<KIND, false, false, 2>(p, max_grid, stream); }
    return p.dcnt ? launch_scan_expand_gen0<KIND, true>(p, max_grid, stream, occ6) : launch_scan_expand_gen0<KIND, false>(p, max_grid, stream, occ6);
}
// digit plans the expanding scan can count while it emits: few enough passes, key digits only, and no digit that
// straddles the two dwords of a packed pair (the kernel extracts a digit from ONE dword)
bool scan_can_count(const RadixPlan& plan) {
    if (plan.n_passes <= 0 || plan.n_passes > SX_HIST_MAX) return false;
    for (int q = 0; q < plan.n_passes; ++q)
        if (plan.from_val[q] || (plan.shift[q] < 32 && plan.shift[q] + 8 > 32)) return false;
    return true;
}
hipError_t launch_scan_expand(int kind, const uint64_t* keys, const uint64_t* idctx, size_t m, const float* filt_bounds, int filt_shape,
                              uint64_t* pairs_out, size_t out_blocks, uint32_t* blockfill, uint64_t* shards,
                              uint64_t* big_list, uint32_t big_cap, DevScalars* sc, hipStream_t stream,
                              const RadixPlan* hist_plan, unsigned long long* hist, const uint32_t* owner_bounds, int owner_n,
                              int gen, int mixed_scale, const uint16_t* dcnt, uint32_t hist_rows) {
    if (m == 0) return hipSuccess;
    // persistent waves: as many workgroups as are resident at once, never more than the work
    const size_t chunks = (m + 63) / 64;
    const size_t max_grid = (chunks + SX_WAVES - 1) / SX_WAVES;
    ExpandParams p;
    p.keys = (const u64*)keys; p.idctx = (const u64*)idctx; p.dcnt = dcnt; p.m = m; p.out = (u64*)pairs_out; p.out_blocks = out_blocks;
    p.blockfill = blockfill; p.shards = (u64*)shards; p.sc = sc; p.big_list = (u64*)big_list; p.big_cap = big_cap;
    p.n_waves = 0; p.filt_bounds = filt_bounds; p.filt_shape = filt_shape;
    // no work list expected (the last scan listed nothing): the work-list kernel is not launched; gen != 0 (the marking passes) always has it
    p.big_inline = (mixed_scale < 0 && gen == 0) ? 1u : 0u;
    p.hist = (u64*)hist; p.hist_passes = 0; p.hist_rows = hist_rows;
    for (int q = 0; q < SX_HIST_MAX; ++q) p.hist_shift[q] = 0;
    if (hist_plan && hist && scan_can_count(*hist_plan)) {
        p.hist_passes = hist_plan->n_passes;
        for (int q = 0; q < hist_plan->n_passes; ++q) p.hist_shift[q] = hist_plan->shift[q];
    }
    p.owner_n = -1;
    for (int i = 0; i < 15; ++i) p.owner_bounds[i] = 0xFFFFFFFFu;
    if (hist && (owner_bounds || owner_n == 0) && owner_n >= 0 && owner_n <= 15 && p.hist_passes == 0) {  // owner_n == 0: one owner, no bounds
        p.owner_n = owner_n;
        for (int i = 0; i < owner_n; ++i) p.owner_bounds[i] = owner_bounds[i];
    }
    switch (kind) {
        case 0: return launch_scan_expand_kind<0>(p, max_grid, stream, gen, mixed_scale > 0);
        case 1: return launch_scan_expand_kind<1>(p, max_grid, stream, gen, mixed_scale > 0);
        default: return launch_scan_expand_kind<2>(p, max_grid, stream, gen, mixed_scale > 0);
    }
}

// =================================================================================================
// Batched region queries: Layer::test_box (src/layer.rs:294-313 -> test :252-279 -> test_impl :167-239
// with BoxTestGeometry, src/geom.rs:353-455), many boxes against one sorted tree.
//
// One lane per query walks the reference's recursion depth-first with an explicit per-level stack in
// scratch memory: a node = (cell, slice of the tree, f32 cell bounds); `should_test` is the inclusive
// overlap of the cell bounds with the test box; the entries exactly at the cell (the head of the
// slice, before the first child's key) are reported, then the 2^dim children in key order, each child's
// slice ending at the lower bound of the next child's key (the reference's binary_search_by split);
// at max_depth or at the depth limit the whole slice is reported.  Child bounds replay
// BoxTestGeometry::subdivide: centre = min + (max - min) / 2 (cgmath ^0.17 EuclideanSpace::midpoint),
// child c takes [min, centre] or [centre, max] per axis by bit `axis` of c -- the same f32 operations
// in the same order, so the cell bounds are bit-identical to the reference's.
// Output: (query << 32 | id) records through the scan's block-structured wave stage; the caller's
// sort + dedup yields each query's sorted duplicate-free id list (results.sort(); results.dedup()).
// =================================================================================================
// RAY = false: Layer::test_box, query row = [min(dim), max(dim)].
// RAY = true:  Layer::test_ray (src/layer.rs:327-351) with RayTestGeometry (src/geom.rs:460-689), query row =
// [origin(dim), direction(dim), range_min, range_max]: with_system_bounds clamps the range to the system
// bounds (:511-537), subdivide narrows it by the distances to the centre planes (:617-661, the non-finite
// distance case included), should_test = range_min < range_max (:686-688).  test_order only matters for the
// nearest-hit early-out of `pick`, not for the result set of `test`.
// Slice bounds of the top levels, shared by every query: qtab[qtab_offset(l) + c] = first index whose key is >=
// the key of cell c (Morton prefix, in key order) at depth l, for l = 1..levels.  The walk below reads these
// instead of bisecting the largest slices (the values are the same lower bounds).
template <int KIND> __host__ __device__ constexpr u32 qtab_offset(u32 l) {  // entries of the levels 1..l-1
    u32 off = 0, n = 1u << KT<KIND>::DIM;
    for (u32 k = 1; k < l; ++k) { off += n; n <<= KT<KIND>::DIM; }
    return off;
}
template <int KIND>
__global__ __launch_bounds__(256) void query_table_kernel(const u64* __restrict__ keys, u32 m, u32 levels, u32* __restrict__ qtab) {
    constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB, DB = KT<KIND>::DB, TOP = DIM * AB + DB;
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    u32 l = 1, base = 0, n = 1u << DIM;
    while (l <= levels && t >= base + n) { base += n; n <<= DIM; ++l; }
    if (l > levels) return;
    const u64 key = ((u64)(t - base) << (TOP - DIM * l)) | (u64)l;
    u32 a = 0, b = m;
    while (a < b) {
        const u32 mid = a + ((b - a) >> 1);
        if (keys[mid] < key) a = mid + 1; else b = mid;
    }
    qtab[t] = a;
}

template <int KIND, bool RAY>
__global__ __launch_bounds__(SX_THREADS) void region_query_kernel(ExpandParams p, const u32* __restrict__ ids, SysBounds sb,
                                                                  const float* __restrict__ queries, u32 n_queries,
                                                                  int has_max_depth, u32 max_depth,
                                                                  const u32* __restrict__ qtab, u32 qtab_levels) {
    constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB, DB = KT<KIND>::DB, NSUB = 1u << DIM;
    constexpr u32 TOP = DIM * AB + DB;
    constexpr u32 ROW = RAY ? 2 * DIM + 2 : 2 * DIM;
    __shared__ u64 s_stage[SX_WAVES][SX_STAGE];
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const u32 gw = blockIdx.x * SX_WAVES + (u32)w;
    WaveStage st{s_stage[w], 0u, gw % SX_SHARDS, 0ull, 0u, nullptr, 0ull};
    const u32 q = blockIdx.x * SX_THREADS + threadIdx.x;
    // per-level stack (dynamically indexed -> scratch)
    u32 lo[AB + 1], hi[AB + 1], cur[AB + 1];
    u32 child[AB + 1];            // next child to visit; NSUB + 1 marks "node not entered yet"
    float mn[AB + 1][DIM], mx[AB + 1][DIM];
    float rmin[RAY ? AB + 1 : 1], rmax[RAY ? AB + 1 : 1];
    u64 cell[AB + 1];
    float qa[DIM], qb[DIM];       // box: test min / max; ray: origin / direction
    int level = -1;
    if (q < n_queries) {
        const float* row = queries + (size_t)q * ROW;
#pragma unroll
        for (u32 a = 0; a < DIM; ++a) {
            qa[a] = row[a];
            qb[a] = row[DIM + a];
            mn[0][a] = sb.mn[a];
            mx[0][a] = sb.mx[a];
        }
        if (RAY) {  // RayTestGeometry::with_system_bounds (geom.rs:511-537)
            float r0 = row[2 * DIM], r1 = row[2 * DIM + 1];
#pragma unroll
            for (u32 a = 0; a < DIM; ++a) {
                const float distance_0 = (sb.mn[a] - qa[a]) / qb[a];
                const float distance_1 = (sb.mx[a] - qa[a]) / qb[a];
                const bool is_forward = qb[a] > 0.0f;
                const float d0 = is_forward ? distance_0 : distance_1, d1 = is_forward ? distance_1 : distance_0;
                if (isfinite(d0)) r0 = fmaxf(r0, d0);  // f32::max / min: a NaN operand loses, like fmaxf / fminf
                if (isfinite(d1)) r1 = fminf(r1, d1);
            }
            rmin[0] = r0; rmax[0] = r1;
        }
        lo[0] = 0; hi[0] = (u32)p.m; cell[0] = 0ull; child[0] = NSUB + 1u;
        level = 0;
    }
    u32 ea = 0, eb = 0;  // pending report: entries [ea, eb) of the tree
    const u64 qhi = (u64)q << 32;
    auto lower_bound = [&](u32 a, u32 b, u64 key) {  // first index in [a, b) whose key is >= key
        while (a < b) {
            const u32 mid = a + ((b - a) >> 1);
            if (p.keys[mid] < key) a = mid + 1; else b = mid;
        }
        return a;
    };
    while (__any(level >= 0 || ea < eb)) {  // wave-uniform loop: every lane reports at most one id per turn
        bool keep = false;
        u64 rec = 0;
        if (ea < eb) {
            keep = true;
            rec = qhi | (u64)ids[ea];
            ++ea;
        } else if (level >= 0) {
            const int l = level;
            if (child[l] == NSUB + 1u) {  // entering the node (test_impl :181-197, :232-237)
                bool ov = lo[l] < hi[l];
                if (RAY) {
                    ov = ov && rmin[RAY ? l : 0] < rmax[RAY ? l : 0] && rmin[RAY ? l : 0] < INFINITY;
                } else {
#pragma unroll
                    for (u32 a = 0; a < DIM; ++a) ov = ov && !(mn[l][a] > qb[a] || mx[l][a] < qa[a]);  // geom.rs:112-119
                }
                if (!ov) {
                    --level;
                } else if ((has_max_depth && (u32)l >= max_depth) || (u32)l >= AB) {
                    ea = lo[l]; eb = hi[l];
                    --level;
                } else {
                    const u64 first_child = cell[l] + 1ull;  // child 0: same Morton bits, depth + 1
                    const u32 s0 = (u32)l + 1u <= qtab_levels
                                       ? qtab[qtab_offset<KIND>((u32)l + 1u) + (u32)(first_child >> (TOP - DIM * ((u32)l + 1u)))]
                                       : lower_bound(lo[l], hi[l], first_child);
                    ea = lo[l]; eb = s0;  // the entries exactly at this cell (:216-219)
                    cur[l] = s0;
                    child[l] = 0;
                }
            } else if (child[l] >= NSUB) {
                --level;
            } else {
                const u32 c = child[l];
                const u32 shift = TOP - DIM * ((u32)l + 1u);
                const u64 base_key = (cell[l] & ~(u64)((1u << DB) - 1u)) | (u64)((u32)l + 1u);
                // the child's geometry first (subdivide + should_test need no tree access): a child the query misses
                // is never entered, so its slice is not located at all -- only the children that are visited pay
                // for their bisections (the slices are the same lower bounds either way)
                float cmn[DIM], cmx[DIM];
                float r0 = RAY ? rmin[RAY ? l : 0] : 0.f, r1 = RAY ? rmax[RAY ? l : 0] : 0.f;
                bool ovc = true;
#pragma unroll
                for (u32 a = 0; a < DIM; ++a) {
                    const float ctr = mn[l][a] + (mx[l][a] - mn[l][a]) / 2.0f;
                    const bool upper = (c >> a) & 1u;
                    if (RAY) {
                        const float distance = (ctr - qa[a]) / qb[a];
                        if (isfinite(distance)) {
                            const bool is_towards = (qb[a] > 0.0f) != upper;
                            if (is_towards) r1 = fminf(r1, distance); else r0 = fmaxf(r0, distance);
                        } else if ((qa[a] > ctr) != upper) {
                            r0 = INFINITY; r1 = -INFINITY;
                        }
                    }
                    cmn[a] = upper ? ctr : mn[l][a];
                    cmx[a] = upper ? mx[l][a] : ctr;
                    if (!RAY) ovc = ovc && !(cmn[a] > qb[a] || cmx[a] < qa[a]);
                }
                if (RAY) ovc = r0 < r1 && r0 < INFINITY;
                child[l] = c + 1u;
                if (!ovc) {
                    cur[l] = 0xFFFFFFFFu;  // the next visited child locates its own start
                } else {
                    const u64 this_key = base_key | ((u64)c << shift);
                    const bool tab = (u32)l + 1u <= qtab_levels;
                    const u32 a0 = cur[l] != 0xFFFFFFFFu ? cur[l]
                                   : (tab ? qtab[qtab_offset<KIND>((u32)l + 1u) + (u32)(this_key >> shift)] : lower_bound(lo[l], hi[l], this_key));
                    const u64 next_key = base_key | ((u64)(c + 1u) << shift);
                    const u32 b0 = c + 1u == NSUB ? hi[l]
                                   : (tab ? qtab[qtab_offset<KIND>((u32)l + 1u) + (u32)(next_key >> shift)] : lower_bound(a0, hi[l], next_key));
                    cur[l] = b0;
                    if (a0 < b0) {
#pragma unroll
                        for (u32 a = 0; a < DIM; ++a) { mn[l + 1][a] = cmn[a]; mx[l + 1][a] = cmx[a]; }
                        if (RAY) { rmin[RAY ? l + 1 : 0] = r0; rmax[RAY ? l + 1 : 0] = r1; }
                        lo[l + 1] = a0; hi[l + 1] = b0;
                        cell[l + 1] = this_key;
                        child[l + 1] = NSUB + 1u;
                        level = l + 1;
                    }
                }
            }
        }
        st.push(keep, rec, p);
    }
    st.finish(p);
}

// =================================================================================================
// Batched nearest-hit queries: Layer::pick_ray (src/layer.rs:418-446 -> pick :364-404 -> test_impl :167-239) with
// a built-in distance callback in place of the caller's closure: the entry distance of the ray into the AABB of
// the object (rows of `bounds` indexed by id; slab test, 0 when the origin is inside, INFINITY on a miss).
// One lane per ray walks the reference's recursion in RayTestGeometry::test_order (src/geom.rs:664-684) with the
// nearest-hit pruning of should_test (:686-688); `nearest` only ever shrinks, the id is replaced on a strict
// improvement (layer.rs:389-392), so ties keep the first candidate in the reference's visiting order.
// =================================================================================================
template <int DIM> __device__ __forceinline__ float ray_aabb_distance(const float* o, const float* d, const float* __restrict__ row) {
    float tmin = 0.0f, tmax = INFINITY;
    bool miss = false;
#pragma unroll
    for (int a = 0; a < DIM; ++a) {
        const float bmin = row[a], bmax = row[DIM + a];
        if (d[a] == 0.0f) {
            if (o[a] < bmin || o[a] > bmax) miss = true;
        } else {
            float t0 = (bmin - o[a]) / d[a], t1 = (bmax - o[a]) / d[a];
            if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
            tmin = fmaxf(tmin, t0);
            tmax = fminf(tmax, t1);
        }
    }
    return (!miss && tmin <= tmax) ? tmin : INFINITY;
}

template <int KIND>
__global__ __launch_bounds__(256) void pick_ray_kernel(const u64* __restrict__ keys, const u32* __restrict__ ids, u32 m, SysBounds sb,
                                                       const float* __restrict__ rays, u32 n_queries, int has_max_depth,
                                                       u32 max_depth, const float* __restrict__ bounds, u32 n_rows,
                                                       float* __restrict__ out_dist, u32* __restrict__ out_id, DevScalars* sc) {
    constexpr u32 DIM = KT<KIND>::DIM, AB = KT<KIND>::AB, DB = KT<KIND>::DB, NSUB = 1u << DIM;
    constexpr u32 TOP = DIM * AB + DB;
    const u32 q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n_queries) return;
    const float* row = rays + (size_t)q * (2 * DIM + 1);
    float org[DIM], dir[DIM];
#pragma unroll
    for (u32 a = 0; a < DIM; ++a) { org[a] = row[a]; dir[a] = row[DIM + a]; }
    const float max_dist = row[2 * DIM];
    // RayTestGeometry::test_order
    u32 order[NSUB];
    {
        float ab[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (u32 a = 0; a < DIM; ++a) ab[a] = fabsf(dir[a]);
        u32 ax0, ax1, ax2 = 2;
        if (DIM == 2) {
            if (ab[0] <= ab[1]) { ax0 = 0; ax1 = 1; } else { ax0 = 1; ax1 = 0; }
        } else if (ab[0] <= ab[1] && ab[0] <= ab[2]) {
            if (ab[1] <= ab[2]) { ax0 = 0; ax1 = 1; ax2 = 2; } else { ax0 = 0; ax1 = 2; ax2 = 1; }
        } else if (ab[1] <= ab[2]) {
            if (ab[0] <= ab[2]) { ax0 = 1; ax1 = 0; ax2 = 2; } else { ax0 = 1; ax1 = 2; ax2 = 0; }
        } else {
            if (ab[0] <= ab[1]) { ax0 = 2; ax1 = 0; ax2 = 1; } else { ax0 = 2; ax1 = 1; ax2 = 0; }
        }
        const u32 axes[3] = {ax0, ax1, ax2};
#pragma unroll
        for (u32 src = 0; src < NSUB; ++src) {
            u32 dst = 0;
#pragma unroll
            for (u32 k = 0; k < DIM; ++k) {
                const bool bit = (((src >> k) & 1u) != 0) == (dir[axes[k]] >= 0.0f);
                dst |= (bit ? 1u : 0u) << axes[k];
            }
            order[src] = dst;
        }
    }
    // per-level stack (dynamically indexed -> scratch)
    u32 bnd[AB + 1][NSUB + 1];   // slice boundaries of the children (bnd[l][NSUB] = end of the node's slice)
    u32 pos[AB + 1];             // next position in `order`; NSUB + 1 marks "node not entered yet"
    u32 lo[AB + 1], hi[AB + 1];
    float mn[AB + 1][DIM], mx[AB + 1][DIM], rmin[AB + 1], rmax[AB + 1];
    u64 cell[AB + 1];
    float r0 = 0.0f, r1 = max_dist;  // pick_ray: with_system_bounds(.., 0, max_dist) (layer.rs:434-439)
#pragma unroll
    for (u32 a = 0; a < DIM; ++a) {
        mn[0][a] = sb.mn[a]; mx[0][a] = sb.mx[a];
        const float distance_0 = (sb.mn[a] - org[a]) / dir[a], distance_1 = (sb.mx[a] - org[a]) / dir[a];
        const bool is_forward = dir[a] > 0.0f;
        const float d0 = is_forward ? distance_0 : distance_1, d1 = is_forward ? distance_1 : distance_0;
        if (isfinite(d0)) r0 = fmaxf(r0, d0);
        if (isfinite(d1)) r1 = fminf(r1, d1);
    }
    rmin[0] = r0; rmax[0] = r1;
    lo[0] = 0; hi[0] = m; cell[0] = 0ull; pos[0] = NSUB + 1u;
    int level = 0;
    float nearest = max_dist;
    u32 result = 0xFFFFFFFFu;
    bool bad_id = false;
    auto fold = [&](u32 a, u32 b) {  // callback(...).min(nearest) over a slice (layer.rs:193-196, 216-219)
        for (u32 e = a; e < b; ++e) {
            const u32 id = ids[e];
            if (id >= n_rows) { bad_id = true; continue; }
            const float dist = ray_aabb_distance<(int)DIM>(org, dir, bounds + (size_t)id * (2 * DIM));
            if (isfinite(dist)) {
                if (dist < nearest) result = id;
                nearest = fminf(dist, nearest);
            }
        }
    };
    while (level >= 0) {
        const int l = level;
        if (pos[l] == NSUB + 1u) {  // entering the node
            if (!(lo[l] < hi[l] && rmin[l] < rmax[l] && rmin[l] < nearest)) { --level; continue; }
            if ((has_max_depth && (u32)l >= max_depth) || (u32)l >= AB) { fold(lo[l], hi[l]); --level; continue; }
            const u32 shift = TOP - DIM * ((u32)l + 1u);
            const u64 base_key = (cell[l] & ~(u64)((1u << DB) - 1u)) | (u64)((u32)l + 1u);
            u32 a = lo[l];
            for (u32 c = 0; c < NSUB; ++c) {  // the reference's successive binary_search_by splits
                u32 b = hi[l];
                const u64 key = base_key | ((u64)c << shift);
                while (a < b) { const u32 mid = a + ((b - a) >> 1); if (keys[mid] < key) a = mid + 1; else b = mid; }
                bnd[l][c] = a;
            }
            bnd[l][NSUB] = hi[l];
            fold(lo[l], bnd[l][0]);
            pos[l] = 0;
            continue;
        }
        if (pos[l] >= NSUB) { --level; continue; }
        const u32 c = order[pos[l]];
        pos[l] += 1u;
        const u32 a0 = bnd[l][c], b0 = bnd[l][c + 1u];
        if (a0 >= b0) continue;
        const u32 shift = TOP - DIM * ((u32)l + 1u);
        float c0 = rmin[l], c1 = rmax[l];
#pragma unroll
        for (u32 a = 0; a < DIM; ++a) {
            const float ctr = mn[l][a] + (mx[l][a] - mn[l][a]) / 2.0f;
            const bool upper = (c >> a) & 1u;
            const float distance = (ctr - org[a]) / dir[a];
            if (isfinite(distance)) {
                const bool is_towards = (dir[a] > 0.0f) != upper;
                if (is_towards) c1 = fminf(c1, distance); else c0 = fmaxf(c0, distance);
            } else if ((org[a] > ctr) != upper) {
                c0 = INFINITY; c1 = -INFINITY;
            }
            mn[l + 1][a] = upper ? ctr : mn[l][a];
            mx[l + 1][a] = upper ? mx[l][a] : ctr;
        }
        rmin[l + 1] = c0; rmax[l + 1] = c1;
        lo[l + 1] = a0; hi[l + 1] = b0;
        cell[l + 1] = ((cell[l] & ~(u64)((1u << DB) - 1u)) | (u64)((u32)l + 1u)) | ((u64)c << shift);
        pos[l + 1] = NSUB + 1u;
        level = l + 1;
    }
    out_dist[q] = result == 0xFFFFFFFFu ? INFINITY : nearest;
    out_id[q] = result;
    if (bad_id) atomicOr(&sc->err, (u32)KERR_BAD_ID);
}

// =================================================================================================
// Sorted merge of two (key, id)-sorted runs (static-layer reuse): the crate's per-frame
// pattern merges a pre-sorted static layer into the frame's dynamic layer (src/layer.rs:123-138) and sorts
// again; with the static entries kept as one sorted run only the dynamic run needs a radix sort, then one
// merge pass.  Merge path: every workgroup finds where its MG_TILE output slots start in A and B (bisection on
// its diagonal), loads the two pieces into LDS and places each element at own index + rank in the other
// piece (A before B on ties: exact duplicates are identical records).
// =================================================================================================
constexpr int MG_THREADS = 256;
constexpr int MG_TILE = 2048;
__device__ __forceinline__ bool rec_less(u64 ka, u32 ia, u64 kb, u32 ib) { return ka < kb || (ka == kb && ia < ib); }
// merge-path split of diagonal d: a = #elements of A among the first d outputs
__device__ __forceinline__ u32 merge_split(const u64* __restrict__ ka, const u32* __restrict__ ia, u32 na, const u64* __restrict__ kb,
                                           const u32* __restrict__ ib, u32 nb, u32 d) {
    u32 lo = d > nb ? d - nb : 0u, hi = min(d, na);
    while (lo < hi) {
        const u32 a = lo + ((hi - lo) >> 1), b = d - a;  // A[a] vs B[b - 1]: is A[a] among the first d?
        // A[a] precedes B[b - 1] (ties: A first) -> the split lies further right
        if (b > 0 && !rec_less(kb[b - 1], ib[b - 1], ka[a], ia[a])) lo = a + 1; else hi = a;
    }
    return lo;
}
// every tile's split in ONE small launch, one diagonal per thread: inside the merge kernel the bisection was ~20 dependent pairs of HBM
// loads by two threads at the start of every workgroup -- most of a tile's life (33.7 M + 0.34 M entries: 397 us for 0.88 GB)
__global__ __launch_bounds__(256) void merge_splits_kernel(const u64* __restrict__ ka, const u32* __restrict__ ia, u32 na, const u64* __restrict__ kb,
                                                           const u32* __restrict__ ib, u32 nb, u32* __restrict__ splits, u32 n_tiles) {
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t > n_tiles) return;
    const u32 total = na + nb;
    splits[t] = merge_split(ka, ia, na, kb, ib, nb, min(t * (u32)MG_TILE, total));
}
__global__ __launch_bounds__(MG_THREADS) void merge_runs_kernel(const u64* __restrict__ ka, const u32* __restrict__ ia,
                                                               const uint8_t* __restrict__ ba, u32 na,
                                                               const u64* __restrict__ kb, const u32* __restrict__ ib,
                                                               const uint8_t* __restrict__ bb, u32 nb, u64* __restrict__ ko,
                                                               u32* __restrict__ io, uint8_t* __restrict__ bo, const u32* __restrict__ splits) {
    __shared__ u64 s_k[MG_TILE];
    __shared__ u32 s_i[MG_TILE];
    __shared__ u32 s_split[2];
    const u32 total = na + nb;
    const u32 d0 = blockIdx.x * MG_TILE, d1 = min(d0 + (u32)MG_TILE, total);
    if (threadIdx.x < 2) s_split[threadIdx.x] = splits ? splits[blockIdx.x + threadIdx.x] : merge_split(ka, ia, na, kb, ib, nb, threadIdx.x == 0 ? d0 : d1);
    __syncthreads();
    const u32 a0 = s_split[0], a1 = s_split[1], b0 = d0 - a0, b1 = d1 - a1;
    const u32 la = a1 - a0, lb = b1 - b0;  // la + lb = d1 - d0 <= MG_TILE
    for (u32 k = threadIdx.x; k < la + lb; k += MG_THREADS) {
        const bool from_a = k < la;
        s_k[k] = from_a ? ka[a0 + k] : kb[b0 + (k - la)];
        s_i[k] = from_a ? ia[a0 + k] : ib[b0 + (k - la)];
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < la + lb; k += MG_THREADS) {
        const bool from_a = k < la;
        const u64 key = s_k[k];
        const u32 id = s_i[k];
        u32 lo, hi;
        if (from_a) { lo = la; hi = la + lb; } else { lo = 0; hi = la; }
        // rank in the other piece: A counts B elements strictly less; B counts A elements less-or-equal
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            const bool go_right = from_a ? rec_less(s_k[mid], s_i[mid], key, id) : !rec_less(key, id, s_k[mid], s_i[mid]);
            if (go_right) lo = mid + 1; else hi = mid;
        }
        const u32 pos = from_a ? k + (lo - la) : (k - la) + lo;
        const size_t o = (size_t)d0 + pos;
        ko[o] = key;
        io[o] = id;
        if (bo) bo[o] = from_a ? ba[a0 + k] : bb[b0 + (k - la)];
    }
}

__global__ __launch_bounds__(256) void blockfill_to_u64_kernel(const u32* __restrict__ f, size_t nb, u64* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nb) out[i] = f[i];
}
__global__ __launch_bounds__(256) void compact_blocks_kernel(const u64* __restrict__ in, size_t nb, const u32* __restrict__ fill,
                                                             const u64* __restrict__ offs, u64* __restrict__ out) {
    const size_t g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per block
    if (g >= nb) return;
    const u32 cnt = fill[g];
    const u64 o = offs[g];
    for (u32 k = (u32)lane_id(); k < cnt; k += 64) out[o + k] = in[g * SX_BLOCK + k];
}
uint32_t query_table_entries(int kind, uint32_t levels) {
    const KindInfo ki = kind_info(kind);
    uint32_t tot = 0, n = 1u << ki.dim;
    for (uint32_t l = 1; l <= levels; ++l) { tot += n; n <<= ki.dim; }
    return tot;
}
hipError_t launch_query_table(int kind, const uint64_t* keys, size_t m, uint32_t levels, uint32_t* qtab, hipStream_t stream) {
    const uint32_t n = query_table_entries(kind, levels);
    if (n == 0) return hipSuccess;
    const unsigned g = (n + 255) / 256;
    BP_DISPATCH_KIND(kind, (query_table_kernel<K><<<g, 256, 0, stream>>>((const u64*)keys, (u32)m, levels, qtab)));
    return hipGetLastError();
}
hipError_t launch_region_query(int kind, int ray, const uint64_t* keys, const uint32_t* ids, size_t m, SysBounds sb, const float* boxes,
                               uint32_t n_queries, int has_max_depth, uint32_t max_depth, uint64_t* out, size_t out_blocks,
                               uint32_t* blockfill, uint64_t* shards, hipStream_t stream, const uint32_t* qtab, uint32_t qtab_levels) {
    if (n_queries == 0 || m == 0) return hipSuccess;
    ExpandParams p;
    p.keys = (const u64*)keys; p.idctx = nullptr; p.m = m; p.out = (u64*)out; p.out_blocks = out_blocks;
    p.blockfill = blockfill; p.shards = (u64*)shards; p.sc = nullptr; p.big_list = nullptr; p.big_cap = 0; p.big_inline = 0;
    p.n_waves = 0; p.filt_bounds = nullptr; p.filt_shape = FILT_AABB; p.hist = nullptr; p.hist_passes = 0; p.hist_rows = 0; p.owner_n = -1;
    for (int q = 0; q < SX_HIST_MAX; ++q) p.hist_shift[q] = 0;
    const unsigned g = (n_queries + SX_THREADS - 1) / SX_THREADS;
    if (ray) {
        BP_DISPATCH_KIND(kind, (region_query_kernel<K, true><<<g, SX_THREADS, 0, stream>>>(p, ids, sb, boxes, n_queries, has_max_depth, max_depth, qtab, qtab_levels)));
    } else {
        BP_DISPATCH_KIND(kind, (region_query_kernel<K, false><<<g, SX_THREADS, 0, stream>>>(p, ids, sb, boxes, n_queries, has_max_depth, max_depth, qtab, qtab_levels)));
    }
    return hipGetLastError();
}

hipError_t launch_pick_ray(int kind, const uint64_t* keys, const uint32_t* ids, size_t m, SysBounds sb, const float* rays,
                           uint32_t n_queries, int has_max_depth, uint32_t max_depth, const float* bounds, uint32_t n_rows,
                           float* out_dist, uint32_t* out_id, DevScalars* sc, hipStream_t stream) {
    if (n_queries == 0) return hipSuccess;
    const unsigned g = (n_queries + 255) / 256;
    BP_DISPATCH_KIND(kind, (pick_ray_kernel<K><<<g, 256, 0, stream>>>((const u64*)keys, ids, (u32)m, sb, rays, n_queries, has_max_depth,
                                                                      max_depth, bounds, n_rows, out_dist, out_id, sc)));
    return hipGetLastError();
}

hipError_t launch_merge_runs(const uint64_t* ka, const uint32_t* ia, const uint8_t* ba, size_t na, const uint64_t* kb,
                             const uint32_t* ib, const uint8_t* bb, size_t nb, uint64_t* ko, uint32_t* io, uint8_t* bo,
                             hipStream_t stream, uint32_t* splits) {
    if (na + nb == 0) return hipSuccess;
    if (na + nb >= 0xFFFFFFFFull) return hipErrorInvalidValue;
    const unsigned g = (unsigned)((na + nb + MG_TILE - 1) / MG_TILE);
    if (splits) merge_splits_kernel<<<(g + 1 + 255) / 256, 256, 0, stream>>>((const u64*)ka, ia, (u32)na, (const u64*)kb, ib, (u32)nb, splits, g);
    merge_runs_kernel<<<g, MG_THREADS, 0, stream>>>((const u64*)ka, ia, ba, (u32)na, (const u64*)kb, ib, bb, (u32)nb, (u64*)ko, io, bo, splits);
    return hipGetLastError();
}
size_t merge_runs_split_words(size_t n_total) { return (n_total + MG_TILE - 1) / MG_TILE + 1; }

// A large device-to-device copy as a kernel: hipMemcpyAsync moved a sorted layer's 0.27 GB of index values into another layer (merge)
// at 1.5 TB/s; four independent 16-byte loads per thread and step run at the memory's rate.  T: the widest type both pointers and the
// size are aligned to.
template <typename T>
__global__ __launch_bounds__(256) void copy_kernel(T* __restrict__ dst, const T* __restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const T a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
hipError_t launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    if (bytes < ((size_t)1 << 20)) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    const size_t al = (size_t)dst | (size_t)src | bytes;
    const unsigned grid = 2048;
    if ((al & 15) == 0) copy_kernel<uint4><<<grid, 256, 0, stream>>>((uint4*)dst, (const uint4*)src, bytes / 16);
    else if ((al & 7) == 0) copy_kernel<u64><<<grid, 256, 0, stream>>>((u64*)dst, (const u64*)src, bytes / 8);
    else if ((al & 3) == 0) copy_kernel<u32><<<grid, 256, 0, stream>>>((u32*)dst, (const u32*)src, bytes / 4);
    else copy_kernel<uint8_t><<<grid, 256, 0, stream>>>((uint8_t*)dst, (const uint8_t*)src, bytes);
    return hipGetLastError();
}

hipError_t launch_blockfill_to_u64(const uint32_t* blockfill, size_t n_blocks, unsigned long long* out, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    blockfill_to_u64_kernel<<<(unsigned)((n_blocks + 255) / 256), 256, 0, stream>>>(blockfill, n_blocks, (u64*)out);
    return hipGetLastError();
}
hipError_t launch_compact_blocks(const uint64_t* in, size_t n_blocks, const uint32_t* blockfill,
                                 const unsigned long long* block_offs, uint64_t* out, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    compact_blocks_kernel<<<(unsigned)((n_blocks + 3) / 4), 256, 0, stream>>>((const u64*)in, n_blocks, blockfill, (const u64*)block_offs,
                                                                             (u64*)out);
    return hipGetLastError();
}

hipError_t launch_unique_count(const uint64_t* sorted, size_t n, unsigned long long* block_sums, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    unique_count_kernel<<<(unsigned)uq_blocks(n), UQ_THREADS, 0, stream>>>((const u64*)sorted, n, block_sums);
    return hipGetLastError();
}

hipError_t launch_unique_emit(const uint64_t* sorted, size_t n, const unsigned long long* block_offs,
                              uint32_t* pairs_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    unique_emit_kernel<<<(unsigned)uq_blocks(n), UQ_THREADS, 0, stream>>>((const u64*)sorted, n, block_offs, (uint2*)pairs_out);
    return hipGetLastError();
}

// =================================================================================================
// Grouped finish: the pair sort stops one digit early.  The stream is sorted by (first id >> G) only;
// the pairs of one group of 2^G consecutive first ids (GRP_CAP at most) are contiguous, and one
// workgroup puts a group in final order in LDS: counting sort by the low G bits of the first id
// (= the id within the group), then each element is ranked by second id inside its id's segment
// (the pairs are distinct), and the group leaves in the ABI layout with coalesced stores.  Replaces
// the last radix pass + segment_finish by two small kernels.
// =================================================================================================
#ifndef GRP_THREADS_MACRO
#define GRP_THREADS_MACRO 512
#endif
#ifndef GRP_DIRECT_STORE
#define GRP_DIRECT_STORE 1  // 0: the group is put in final order in LDS first and leaves with coalesced stores (0.369 against 0.352 ms at 10 M)
#endif
constexpr int GRP_THREADS = GRP_THREADS_MACRO, GRP_CAP = 4096, GRP_ITEMS = GRP_CAP / GRP_THREADS, GRP_MAX_BITS = 8;
constexpr u32 GRP_SEG_MAX = 4096;  // an id with more pairs than this: fall back (the rank loop is quadratic in it)
constexpr u32 GRP_SLOW_MAX = 16u * GRP_CAP;
#ifndef GRP_SUB_DEF
#define GRP_SUB_DEF 8
#endif
constexpr u32 GRP_SUB = GRP_SUB_DEF;  // sub-bins per id by the range of the second id (fast path)
template <bool GAPPED>
__global__ __launch_bounds__(GRP_THREADS) void group_finish_kernel(const u64* __restrict__ in, const u32* __restrict__ bounds, u32 g0,
                                                                   int gbits, u32 sub_scale, uint2* __restrict__ out, DevScalars* sc,
                                                                   const u32* __restrict__ stats, u32 abort_over,
                                                                   const u32* __restrict__ dev_sizes, CellRoute cells) {
    u32 dev_n = dev_sizes ? dev_sizes[SIZES_N] : 0u;  // (requested together with the two loads below: see there)
    // bins of the fast path: id within the group x GRP_SUB equal ranges of the second id (second * sub_scale >> 32, clamped):
    // about one pair per bin -- the rank loops are what this kernel spends its instructions on (their cost is quadratic in the
    // bin length, and a wave runs to the longest bin of its 64 lanes)
    constexpr u32 NBINS = 256u * GRP_SUB;
    static_assert(NBINS == 4u * GRP_THREADS, "four bins per thread in the prefix scan");
    __shared__ u32 s_second[GRP_CAP + 8];  // second ids grouped by bin for the rank loops (+ 8: the rank loop's unconditional reads)
    __shared__ __attribute__((aligned(16))) u32 s_cnt[NBINS];
    __shared__ __attribute__((aligned(16))) u32 s_off[NBINS + 4];
    __shared__ u32 s_cur[256];             // (the rounds of an oversize group)
    __shared__ u32 s_ws[GRP_THREADS / 64];
    const u32 g = blockIdx.x, tid = threadIdx.x;
    // most pairs sit in oversize groups (ids that follow the scene's clusters): rounds over every such group would cost far
    // more than the full-width sort the host runs when it sees this flag
    // where record x of the group is read: `in` dense (the group is in[lo .. hi)), or the gapped result of a cell sort (the
    // cell's whole lines from its start, then what was claimed downwards from its end: ONE 16-byte word per group, written by
    // launch_cell_count, instead of the two bounds -- a second, dependent load would add a memory latency to every workgroup;
    // requested together with the abort statistic for the same reason)
    uint4 si = make_uint4(0u, 0u, 0u, 0u);
    if (GAPPED) si = cells.src[g];
    const u32 over_stat = stats ? stats[1] : 0u;
    if (GAPPED) asm volatile("" : "+s"(si.x), "+s"(si.y), "+s"(si.z), "+s"(si.w), "+s"(dev_n));  // (else the loads sink below the abort test, one round trip each)
    if (dev_sizes) abort_over = dev_n / 4u;
    if (over_stat > abort_over) {
        if (g == 0 && tid == 0) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
        return;
    }
    u32 lo, ng, src_a, src_nf, src_b = 0;
    if (GAPPED) {
        src_a = si.x; src_b = si.y; lo = si.z; ng = si.w >> 16; src_nf = si.w & 0xFFFFu;
    } else {
        lo = bounds[g];
        ng = bounds[g + 1] - lo;
        src_a = lo; src_nf = ng;
    }
    if (ng == 0) return;  // workgroup-uniform
    auto src = [&](u32 x) -> size_t { return !GAPPED || x < src_nf ? (size_t)src_a + x : (size_t)src_b + (x - src_nf); };
    const u32 bmask = (1u << gbits) - 1u;
    if (ng > (u32)GRP_CAP) {
        // A group that does not fit the LDS buffers (rare: the host only picks this kernel when the average group is
        // well below the capacity) is finished in several rounds over ranges of ids that do fit, re-reading the group
        // (L2-resident) in every round.  Only an id with more than GRP_CAP pairs of its own cannot be handled.
        if (ng > GRP_SLOW_MAX) {
            if (tid == 0) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
            return;
        }
        if (tid < 256u) s_cnt[tid] = 0;
        __syncthreads();
        for (u32 x = tid; x < ng; x += GRP_THREADS)
            __hip_atomic_fetch_add(&s_cnt[(u32)(in[src(x)] >> 32) & bmask], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        const u32 mine_s = tid < 256u ? s_cnt[tid] : 0u;
        u32 total_s;
        const u32 excl_s = block_excl_scan_u32<GRP_THREADS>(mine_s, s_ws, total_s);
        if (tid < 256u) s_off[tid] = excl_s;
        if (tid == 0) s_off[256] = ng;
        if (mine_s > (u32)GRP_CAP) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);  // the host discards this output
        __syncthreads();
        const u32 first_base_s = (g0 + g) << gbits;
        u32 b0 = 0;
        while (b0 < 256u) {  // workgroup-uniform: every thread walks the same offsets
            u32 b1 = b0 + 1u;
            while (b1 < 256u && s_off[b1 + 1u] - s_off[b0] <= (u32)GRP_CAP) ++b1;
            const u32 base = s_off[b0];
            const u32 cnt = min(s_off[b1] - base, (u32)GRP_CAP);  // > GRP_CAP only for an over-long id (flagged above)
            if (tid >= b0 && tid < b1) s_cur[tid] = s_off[tid] - base;
            __syncthreads();
            for (u32 x = tid; x < ng; x += GRP_THREADS) {
                const u64 vv = in[src(x)];
                const u32 b = (u32)(vv >> 32) & bmask;
                if (b >= b0 && b < b1) {
                    const u32 pos = __hip_atomic_fetch_add(&s_cur[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (pos < (u32)GRP_CAP) s_second[pos] = (u32)vv;
                }
            }
            __syncthreads();
            for (u32 x = tid; x < ng; x += GRP_THREADS) {
                const u64 vv = in[src(x)];
                const u32 b = (u32)(vv >> 32) & bmask, vb = (u32)vv;
                if (b >= b0 && b < b1) {
                    const u32 start = s_off[b] - base, end = min(s_off[b + 1u] - base, (u32)GRP_CAP);
                    u32 rank = 0;
                    for (u32 y = start; y < end; ++y) rank += s_second[y] < vb ? 1u : 0u;
                    // rare path: ranks go straight to the output (stores scattered inside the round's range)
                    if (start + rank < cnt) out[(size_t)lo + base + start + rank] = make_uint2(first_base_s | b, vb);
                }
            }
            __syncthreads();
            b0 = b1;
        }
        return;
    }
    // (sub_scale below 2^24 says that the second ids are too: one v_mul_hi_u32_u24 instead of v_mul_hi_u32 + the shift that undoes
    // the halved scale)
    const bool sub24 = (sub_scale >> 31) != 0u;  // (the launcher's flag, in the top bit of the scale's word: see there)
    const u32 sub_mul = sub_scale & 0x7FFFFFFFu;
    auto bin_of = [&](u64 v) {
        u32 h;
        if (sub24) asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(h) : "v"((u32)v), "v"(sub_mul));
        else h = __umulhi((u32)v, sub_mul) << 1;  // (the scale was halved to make room for the flag)
        const u32 sub = min(h, GRP_SUB - 1u);  // (monotone in the second id)
        return (((u32)(v >> 32) & bmask) * GRP_SUB) | sub;
    };
    reinterpret_cast<uint4*>(s_cnt)[tid] = make_uint4(0u, 0u, 0u, 0u);
    u64 v[GRP_ITEMS];
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; ++k) {
        const u32 x = (u32)k * GRP_THREADS + tid;
        v[k] = x < ng ? in[src(x)] : 0ull;
    }
    __syncthreads();
    // one returning LDS atomic per pair: the arrival ticket inside its bin is also its slot there (no second, cursor pass)
    u32 ticket[GRP_ITEMS];  // bin << 16 | arrival ticket (both < 2^16: NBINS = 1024, GRP_CAP = 4096)
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; ++k) {
        const u32 x = (u32)k * GRP_THREADS + tid;
        ticket[k] = 0;
        if (x < ng) {
            const u32 b = bin_of(v[k]);
            ticket[k] = (b << 16) | __hip_atomic_fetch_add(&s_cnt[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const uint4 c4 = reinterpret_cast<const uint4*>(s_cnt)[tid];
    u32 total;
    const u32 excl = block_excl_scan_u32<GRP_THREADS>(c4.x + c4.y + c4.z + c4.w, s_ws, total);
    reinterpret_cast<uint4*>(s_off)[tid] = make_uint4(excl, excl + c4.x, excl + c4.x + c4.y, excl + c4.x + c4.y + c4.z);
    if (tid == 0) s_off[NBINS] = ng;
    const bool heavy = max(max(c4.x, c4.y), max(c4.z, c4.w)) > GRP_SEG_MAX;
    if (__any(heavy) && lane_id() == 0) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; ++k) {
        const u32 x = (u32)k * GRP_THREADS + tid;
        if (x < ng) s_second[s_off[ticket[k] >> 16] + (ticket[k] & 0xFFFFu)] = (u32)v[k];
    }
    __syncthreads();
    u32 dst[GRP_ITEMS];
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; ++k) {
        const u32 x = (u32)k * GRP_THREADS + tid;
        dst[k] = 0;
        if (x < ng) {
            const u32 b = ticket[k] >> 16, vb = (u32)v[k];
            const u32 start = s_off[b], len = s_off[b + 1] - start;
            // the bin's first four unconditionally (what lies behind a shorter bin is masked by the length: straight-line code,
            // no loop for nine bins in ten) -- the general loop the compiler made of this (unrolled by 8, by 4, remainder, each
            // behind its own exec-mask bookkeeping) cost ~100 instructions per pair for bins of two
            const u32* const q = s_second + start;
            const u32 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];  // (all four issued together, whatever the length)
            u32 rank = ((u32)(len > 0u) & (u32)(q0 < vb)) + ((u32)(len > 1u) & (u32)(q1 < vb)) + ((u32)(len > 2u) & (u32)(q2 < vb)) +
                       ((u32)(len > 3u) & (u32)(q3 < vb));
            if (len > 4u && len <= GRP_SEG_MAX) {  // (longer than GRP_SEG_MAX: the host discards this output)
#pragma unroll 1
                for (u32 y = 4u; y < len; ++y) rank += q[y] < vb ? 1u : 0u;
            }
            dst[k] = start + rank;
        }
    }
#if GRP_DIRECT_STORE
    // every pair goes straight to its final slot: 8-byte stores scattered inside the group's ~18 KB, which the L2 merges into
    // whole lines (all of a group's stores leave within a microsecond) -- cheaper than two more barriers and an LDS round trip
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; ++k) {
        const u32 x = (u32)k * GRP_THREADS + tid;
        if (x < ng) out[(size_t)lo + dst[k]] = make_uint2((u32)(v[k] >> 32), (u32)v[k]);
    }
#else
    __syncthreads();  // every rank is known: the buffer now takes the group in final order
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; ++k) {
        const u32 x = (u32)k * GRP_THREADS + tid;
        if (x < ng) {
            s_second[dst[k]] = (u32)v[k];
            reinterpret_cast<uint8_t*>(s_cnt)[dst[k]] = (uint8_t)((u32)(v[k] >> 32) & bmask);  // (the counters are dead: their storage takes the ids)
        }
    }
    __syncthreads();
    const u32 first_base = (g0 + g) << gbits;
    for (u32 x = tid; x < ng; x += GRP_THREADS) out[(size_t)lo + x] = make_uint2(first_base | (u32)reinterpret_cast<const uint8_t*>(s_cnt)[x], s_second[x]);
#endif
}
hipError_t launch_group_finish(const uint64_t* in, size_t n, uint32_t n_groups, uint32_t g0, int gbits, uint32_t* bounds,
                               uint32_t* pairs_out, DevScalars* sc, hipStream_t stream, uint32_t* stats, uint32_t second_max,
                               const uint32_t* dev_sizes, const CellRoute* cells) {
    if (n == 0 || n_groups == 0) return hipSuccess;
    if (n >= 0xFFFFFFFFull || gbits < 1 || gbits > GRP_MAX_BITS) return hipErrorInvalidValue;
    // stats (two words of the freshly reset scalars): [0] = largest group, [1] = pairs in groups above the capacity; a
    // quarter of all pairs there: the finish bails out
    // (cells: the bounds and the statistics come from launch_cell_count -- the input is gapped, nothing to search)
    // (cells are numbered by bits 8 .. 23 of the first id: the group's cell is g0's low 16 bits + its index; the kernel keeps the full g0 for the ids it writes)
    hipError_t e = cells ? launch_cell_count(*cells, gbits == 8 ? (g0 & 0xFFFFu) : g0, n_groups, bounds, (u32)GRP_CAP, stats, dev_sizes, stream)
                         : launch_group_bounds_stats((const u64*)in, (u32)n, n_groups, (u64)g0, 32 + gbits, bounds, (u32)GRP_CAP, stats, stream, dev_sizes);
    if (e != hipSuccess) return e;
    // second ids below 2^bits: their top two bits split every id's pairs into GRP_SUB ranges (larger values are clamped)
    // second ids at most second_max: GRP_SUB equal ranges of them split every id's pairs (second * scale >> 32; larger values are clamped)
    const unsigned long long scale64 = ((unsigned long long)GRP_SUB << 32) / ((unsigned long long)second_max + 1ull);
    u32 sub_scale = scale64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)scale64;
    // second ids below 2^24 (and enough of them for the scale to fit 24 bits): the kernel multiplies 24-bit operands.  Any monotone
    // map of the second id onto the sub-bins keeps the order; with few ids a clamped scale only leaves the upper sub-bins empty
    const bool sub24 = second_max < (1u << 24);
    if (sub24 && sub_scale >= (1u << 24)) sub_scale = (1u << 24) - 1u;
    sub_scale = sub24 ? (0x80000000u | sub_scale) : (sub_scale >> 1);  // (bit 31: the kernel's switch; else the scale halved, the product doubled)
    if (cells)
        group_finish_kernel<true><<<n_groups, GRP_THREADS, 0, stream>>>((const u64*)in, bounds, g0, gbits, sub_scale, (uint2*)pairs_out, sc, stats,
                                                                        (u32)(n / 4), dev_sizes, *cells);
    else
        group_finish_kernel<false><<<n_groups, GRP_THREADS, 0, stream>>>((const u64*)in, bounds, g0, gbits, sub_scale, (uint2*)pairs_out, sc, stats,
                                                                         (u32)(n / 4), dev_sizes, CellRoute{});
    return hipGetLastError();
}
int group_finish_capacity() { return GRP_CAP; }

// The grouped finish for pairs that MAY REPEAT (ids shared between objects: the source cannot suppress duplicates).
// Same plan -- the radix passes stop above the low G bits of the first id (G <= 8, chosen by the host so that a group
// of 2^G ids fits), one workgroup orders a group in LDS -- with
// every element ranked by the elements of its bin BELOW it, so that the copies of a pair claim the same slot of the final
// order and the slots behind it stay free; the taken slots leave, compacted, at the group's input offset in `tmp`, and
// counts[g] says how many did.  launch_group_compact then closes the gaps
// between groups (exclusive scan of counts -> final offsets).  Raw lists with repeats are about twice as long per id,
// hence the larger group capacity (8192) than the distinct kernel's.
constexpr int GRD_THREADS = 1024, GRD_ITEMS = 12, GRD_CAP = GRD_THREADS * GRD_ITEMS, GRD_WAVES = GRD_THREADS / 64;
constexpr int GRD_HEADS = GRD_ITEMS * GRD_WAVES, GRD_HEADS_PER_LANE = GRD_HEADS / 64;
// LDS (dynamic: above the 64 KB a kernel may declare statically): second ids, id-in-group bytes, bin counts / offsets
constexpr size_t GRD_LDS_BYTES = (size_t)GRD_CAP * 5 + GRD_CAP / 8 + (1024 + 1024 + 1) * 4 + GRD_WAVES * 4 + (GRD_HEADS + 1) * 4 + 64;
constexpr int GRD_MAX_BITS = 8;
constexpr u32 GRD_SEG_MAX = 2048;  // a bin (id x second-id range) longer than this: fall back (quadratic rank loop)
__global__ __launch_bounds__(GRD_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void group_finish_dups_kernel(const u64* __restrict__ in, const u32* __restrict__ bounds, u32 g0,
                                                                        int gbits, int sshift, u32 second_lo, uint2* __restrict__ tmp,
                                                                        u64* __restrict__ counts, DevScalars* sc,
                                                                        const u32* __restrict__ stats) {
    constexpr u32 NBINS = 1024u;
    static_assert(NBINS == (u32)GRD_THREADS, "one bin per thread in the prefix scan");
    static_assert(GRD_HEADS % 64 == 0, "the head counts are scanned by one wave, a few per lane");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    u32* s_second = (u32*)s_dyn;                          // [GRD_CAP]
    u32* s_cnt = s_second + GRD_CAP;                      // [NBINS]
    u32* s_off = s_cnt + NBINS;                           // [NBINS + 1]
    u32* s_ws = s_off + NBINS + 1;                        // [GRD_WAVES]
    u32* s_heads = s_ws + GRD_WAVES;                      // [GRD_HEADS + 1] heads per (round, wave), then their exclusive prefix; last = total
    u32* s_taken = s_heads + GRD_HEADS + 1;               // [GRD_CAP / 32] one bit per slot of the final order
    uint8_t* s_idof = (uint8_t*)(s_taken + GRD_CAP / 32);  // [GRD_CAP] id within the group of the slot's pair
    const u32 g = blockIdx.x, tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const u32 lo = bounds[g], hi = bounds[g + 1];
    const u32 ng = hi - lo;
    // any group above the capacity: the host discards this output and runs the ungrouped path
    if ((stats && stats[1] != 0) || ng > (u32)GRD_CAP) {
        if (tid == 0) {
            atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
            counts[g] = 0;
        }
        return;
    }
    if (ng == 0) {  // workgroup-uniform
        if (tid == 0) counts[g] = 0;
        return;
    }
    const u32 bmask = (1u << gbits) - 1u;
    // 1024 bins: id within the group x 2^(10 - gbits) ranges of the second id (second_lo: a lower bound of the ids)
    const u32 sub_bits = 10u - (u32)gbits, sub_max = (1u << sub_bits) - 1u;
    auto bin_of = [&](u64 v) {
        const u32 sub = min(((u32)v - second_lo) >> sshift, sub_max);
        return (((u32)(v >> 32) & bmask) << sub_bits) | sub;
    };
    s_cnt[tid] = 0;
    __syncthreads();
    // registers per element: the second id and one word that is first bin << 16 | arrival ticket, then the final slot
    // (the id within the group is the bin's high part): 24 + a few, so that two workgroups share a CU
    // (element state is updated outside the divergent branches: the compiler keeps the arrays as register tuples, and a
    // tuple assigned inside a branch is copied whole at the join)
    u32 sec[GRD_ITEMS], tk[GRD_ITEMS];
#pragma unroll
    for (int k = 0; k < GRD_ITEMS; ++k) {
        const u32 x = (u32)k * GRD_THREADS + tid;
        const u64 v = in[(size_t)lo + min(x, ng - 1u)];
        const u32 b = bin_of(v);
        u32 t = 0;
        if (x < ng) t = __hip_atomic_fetch_add(&s_cnt[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sec[k] = (u32)v;
        tk[k] = (b << 16) | t;
    }
    __syncthreads();
    const u32 c0 = s_cnt[tid];
    u32 total;
    const u32 excl = block_excl_scan_u32<GRD_THREADS>(c0, s_ws, total);
    s_off[tid] = excl;
    if (tid == 0) s_off[NBINS] = ng;
    if (__any(c0 > GRD_SEG_MAX) && lane == 0) atomicOr(&sc->err, (u32)KERR_SEGMENT_TOO_LONG);
    // no slot of the final order is taken yet
    if (tid < (u32)GRD_CAP / 32u) s_taken[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GRD_ITEMS; ++k) {
        const u32 x = (u32)k * GRD_THREADS + tid;
        if (x < ng) s_second[s_off[tk[k] >> 16] + (tk[k] & 0xFFFFu)] = sec[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GRD_ITEMS; ++k) {
        const u32 x = (u32)k * GRD_THREADS + tid;
        const u32 b = tk[k] >> 16, vb = sec[k];
        u32 slot = 0;
        if (x < ng) {
            const u32 start = s_off[b], end = s_off[b + 1];
            u32 rank = 0;
            if (end - start <= GRD_SEG_MAX) {  // longer: the host discards this output
                // rank = elements of the bin below this one: the copies of a pair all get the same slot
#pragma unroll 4
                for (u32 y = start; y < end; ++y) rank += s_second[y] < vb ? 1u : 0u;
            }
            slot = start + rank;
        }
        tk[k] = ((b >> sub_bits) << 16) | slot;  // id within the group << 16 | final slot (< 2^14)
    }
    __syncthreads();
    // the copies of a pair store the same words to the same slot; the slots behind it (one per extra copy) stay marked free
#pragma unroll
    for (int k = 0; k < GRD_ITEMS; ++k) {
        const u32 x = (u32)k * GRD_THREADS + tid;
        if (x < ng) {
            s_second[tk[k] & 0xFFFFu] = sec[k];
            s_idof[tk[k] & 0xFFFFu] = (uint8_t)(tk[k] >> 16);
            __hip_atomic_fetch_or(&s_taken[(tk[k] & 0xFFFFu) >> 5], 1u << (tk[k] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    // heads: the taken slots, in final order x = round * THREADS + tid
    u32 idof[GRD_ITEMS];
#pragma unroll
    for (int k = 0; k < GRD_ITEMS; ++k) {
        const u32 x = (u32)k * GRD_THREADS + tid;
        const bool taken = x < ng && ((s_taken[x >> 5] >> (x & 31u)) & 1u) != 0u;
        idof[k] = taken ? (u32)s_idof[min(x, ng - 1u)] : 0xFFFFu;
        const u64 bal = ballot64(taken);
        if (lane == 0) s_heads[k * GRD_WAVES + (int)w] = (u32)__popcll(bal);
    }
    __syncthreads();
    if (w == 0) {
        u32 h[GRD_HEADS_PER_LANE], sum = 0;
#pragma unroll
        for (int q = 0; q < GRD_HEADS_PER_LANE; ++q) {
            h[q] = s_heads[(u32)GRD_HEADS_PER_LANE * lane + q];
            sum += h[q];
        }
        const u32 incl = wave_incl_scan_u32(sum);
        u32 run = incl - sum;
#pragma unroll
        for (int q = 0; q < GRD_HEADS_PER_LANE; ++q) {
            s_heads[(u32)GRD_HEADS_PER_LANE * lane + q] = run;
            run += h[q];
        }
        if (lane == 63) s_heads[GRD_HEADS] = incl;
    }
    __syncthreads();
    const u32 first_base = (g0 + g) << gbits;
#pragma unroll
    for (int k = 0; k < GRD_ITEMS; ++k) {
        const u32 x = (u32)k * GRD_THREADS + tid;
        const bool head = idof[k] != 0xFFFFu;
        const u64 bal = ballot64(head);
        if (head) {
            const u32 pos = s_heads[k * GRD_WAVES + (int)w] + (u32)__popcll(bal & ((1ull << lane) - 1ull));
            tmp[(size_t)lo + pos] = make_uint2(first_base | idof[k], s_second[x]);
        }
    }
    if (tid == 0) counts[g] = (u64)s_heads[GRD_HEADS];
}
// offs: exclusive scan of the groups' counts, offs[n_groups] = total; the group's heads move from tmp[bounds[g] ..] to out[offs[g] ..]
__global__ __launch_bounds__(256) void group_compact_kernel(const uint2* __restrict__ tmp, const u32* __restrict__ bounds,
                                                            const u64* __restrict__ offs, u32 n_groups, uint2* __restrict__ out) {
    for (u32 g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const u64 o = offs[g];
        const u32 cnt = (u32)(offs[g + 1] - o), lo = bounds[g];
        for (u32 x = threadIdx.x; x < cnt; x += 256u) out[o + x] = tmp[(size_t)lo + x];
    }
}
hipError_t launch_group_finish_dups(const uint64_t* in, size_t n, uint32_t n_groups, uint32_t g0, int gbits, uint32_t* bounds, uint64_t* tmp,
                                    unsigned long long* counts, DevScalars* sc, hipStream_t stream, uint32_t* stats, uint32_t second_lo,
                                    uint32_t second_hi) {
    // counts: n_groups + 1 words; counts[n_groups] = 0 so that the exclusive scan leaves the total there
    if (n == 0 || n_groups == 0) return hipSuccess;
    if (n >= 0xFFFFFFFFull || gbits < 4 || gbits > GRD_MAX_BITS) return hipErrorInvalidValue;
    hipError_t e = launch_group_bounds_stats((const u64*)in, (u32)n, n_groups, (u64)g0, 32 + gbits, bounds, (u32)GRD_CAP, stats, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(counts + n_groups, 0, sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    int bits = 0;  // second ids lie in [second_lo, second_hi]
    while (bits < 32 && ((second_hi - second_lo) >> bits) != 0) ++bits;
    const int sub_bits = 10 - gbits;
    const int sshift = bits > sub_bits ? bits - sub_bits : 0;
    // more dynamic LDS than a kernel gets by default: raised once per DEVICE (the attribute belongs to the device's copy of the kernel)
    static std::atomic<unsigned long long> lds_set_mask{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(lds_set_mask.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute((const void*)group_finish_dups_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRD_LDS_BYTES);
        if (e != hipSuccess) return e;
        lds_set_mask.fetch_or(bit, std::memory_order_release);
    }
    group_finish_dups_kernel<<<n_groups, GRD_THREADS, GRD_LDS_BYTES, stream>>>((const u64*)in, bounds, g0, gbits, sshift, second_lo, (uint2*)tmp, (u64*)counts, sc, stats);
    return hipGetLastError();
}
hipError_t launch_group_compact(const uint64_t* tmp, const uint32_t* bounds, const unsigned long long* offs, uint32_t n_groups,
                                uint32_t* pairs_out, hipStream_t stream) {
    if (n_groups == 0) return hipSuccess;
    group_compact_kernel<<<std::min<unsigned>(n_groups, 8192u), 256, 0, stream>>>((const uint2*)tmp, bounds, (const u64*)offs, n_groups,
                                                                                 (uint2*)pairs_out);
    return hipGetLastError();
}
int group_finish_dups_capacity() { return GRD_CAP; }
int group_finish_dups_max_bits() { return GRD_MAX_BITS; }

hipError_t launch_segment_finish_dups(const uint64_t* in, size_t n, uint64_t* sorted_out, DevScalars* sc, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned g = (unsigned)((n + SEGP_CHUNK - 1) / SEGP_CHUNK);
    segment_finish_kernel<true><<<g, SEG_THREADS, 0, stream>>>((const u64*)in, n, (uint2*)sorted_out, sc);
    return hipGetLastError();
}
hipError_t launch_segment_finish(const uint64_t* in, size_t n, uint32_t* pairs_out, DevScalars* sc, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned g = (unsigned)((n + SEGP_CHUNK - 1) / SEGP_CHUNK);
    segment_finish_kernel<false><<<g, SEG_THREADS, 0, stream>>>((const u64*)in, n, (uint2*)pairs_out, sc);
    return hipGetLastError();
}

hipError_t launch_tgroup_bounds(const uint64_t* keys, size_t m, uint32_t n_groups, uint64_t g0, int gshift, uint32_t* bounds,
                                uint32_t* stats, hipStream_t stream) {
    if (m == 0 || n_groups == 0) return hipSuccess;
    if (m >= 0xFFFFFFFFull) return hipErrorInvalidValue;
    return launch_group_bounds_stats((const u64*)keys, (u32)m, n_groups, (u64)g0, gshift, bounds, (u32)TGRP_CAP, stats, stream);
}
bool tgroup_composite_fits(int gshift, int id_bits) {
    static const bool disabled = getenv("BP_NO_TGROUP_COMPOSITE") != nullptr;  // A/B switch
    return !disabled && gshift >= TG2_BIN_BITS && id_bits >= 1 && id_bits <= (int)BOX_ID_BITS && (gshift - TG2_BIN_BITS) + id_bits <= 64;
}
hipError_t launch_tgroup_finish(const uint64_t* keys_in, const uint32_t* ids_in, uint32_t n_groups, const uint32_t* bounds, int gshift,
                                int payload_packed, uint64_t* keys_out, uint32_t* ids_out, uint8_t* box_out, DevScalars* sc,
                                hipStream_t stream, int ctx_kind, uint64_t* idctx_out, const uint32_t* stats, uint32_t abort_over,
                                int distinct_id_bits, const CellRoute* cells) {
    if (n_groups == 0) return hipSuccess;
    if (gshift < TGRP_BIN_BITS) return hipErrorInvalidValue;
    if (cells) {  // gapped input (the result of a cell sort): the composite-key kernel only -- the caller checked tgroup_composite_applies
        if (!(payload_packed && box_out && distinct_id_bits > 0 && tgroup_composite_fits(gshift, distinct_id_bits))) return hipErrorInvalidValue;
#define BP_TG2G_LAUNCH(K)                                                                                                                \
    tgroup_finish2_kernel<K, true><<<n_groups, TGRP_THREADS, 0, stream>>>((const u64*)keys_in, ids_in, bounds, gshift, distinct_id_bits,  \
                                                                          (u64*)keys_out, ids_out, box_out, (u64*)idctx_out, sc, stats, abort_over, cells->src)
        if (!idctx_out || ctx_kind < 0) BP_TG2G_LAUNCH(-1);
        else if (ctx_kind == 0) BP_TG2G_LAUNCH(0);
        else if (ctx_kind == 1) BP_TG2G_LAUNCH(1);
        else BP_TG2G_LAUNCH(2);
#undef BP_TG2G_LAUNCH
        return hipGetLastError();
    }
    if (payload_packed && box_out && distinct_id_bits > 0 && tgroup_composite_fits(gshift, distinct_id_bits)) {
#define BP_TG2_LAUNCH(K)                                                                                                          \
    tgroup_finish2_kernel<K><<<n_groups, TGRP_THREADS, 0, stream>>>((const u64*)keys_in, ids_in, bounds, gshift, distinct_id_bits,  \
                                                                    (u64*)keys_out, ids_out, box_out, (u64*)idctx_out, sc, stats, abort_over)
        if (!idctx_out || ctx_kind < 0) BP_TG2_LAUNCH(-1);
        else if (ctx_kind == 0) BP_TG2_LAUNCH(0);
        else if (ctx_kind == 1) BP_TG2_LAUNCH(1);
        else BP_TG2_LAUNCH(2);
#undef BP_TG2_LAUNCH
        return hipGetLastError();
    }
#define BP_TGRP_LAUNCH(P, K)                                                                                                     \
    tgroup_finish_kernel<P, K><<<n_groups, TGRP_THREADS, 0, stream>>>((const u64*)keys_in, ids_in, bounds, gshift, (u64*)keys_out, \
                                                                      ids_out, box_out, (u64*)idctx_out, sc, stats, abort_over)
    if (!payload_packed) BP_TGRP_LAUNCH(false, -1);
    else if (!idctx_out || ctx_kind < 0) BP_TGRP_LAUNCH(true, -1);
    else if (ctx_kind == 0) BP_TGRP_LAUNCH(true, 0);
    else if (ctx_kind == 1) BP_TGRP_LAUNCH(true, 1);
    else BP_TGRP_LAUNCH(true, 2);
#undef BP_TGRP_LAUNCH
    return hipGetLastError();
}
bool tgroup_composite_applies(int gshift, int distinct_id_bits) { return distinct_id_bits > 0 && tgroup_composite_fits(gshift, distinct_id_bits); }
bool tgroup_compact_applies(int comp_gshift) { return comp_gshift >= TG2_BIN_BITS && comp_gshift <= 63; }
hipError_t launch_tgroup_finish_compact(const uint64_t* words_in, uint32_t n_groups, int comp_gshift, const TreeCompact& fmt, uint64_t* keys_out,
                                        uint32_t* ids_out, uint8_t* box_out, DevScalars* sc, hipStream_t stream, int ctx_kind, uint64_t* idctx_out,
                                        uint16_t* dcnt_out, const uint32_t* stats, uint32_t abort_over, const CellRoute& cells) {
    if (n_groups == 0) return hipSuccess;
    if (!dcnt_out || fmt.T > 15) return hipErrorInvalidValue;
    // (the kernel takes the index kind from its context-word instantiation: the compact route always writes the context words)
    if (comp_gshift < TG2_BIN_BITS || comp_gshift > 63 || !idctx_out || (ids_out == nullptr) != (box_out == nullptr) || ctx_kind < 0 || ctx_kind > 2 ||
        !cells.src)
        return hipErrorInvalidValue;
#define BP_TGC_LAUNCH(K, S)                                                                                                                         \
    do {                                                                                                                                            \
        if (ids_out)                                                                                                                                \
            tgroup_finishc_kernel<K, S, true><<<n_groups, TGRP_THREADS, 0, stream>>>((const u64*)words_in, comp_gshift, fmt, (u64*)keys_out, ids_out, \
                                                                                     box_out, (u64*)idctx_out, dcnt_out, sc, stats, abort_over, cells.src);   \
        else                                                                                                                                        \
            tgroup_finishc_kernel<K, S, false><<<n_groups, TGRP_THREADS, 0, stream>>>((const u64*)words_in, comp_gshift, fmt, (u64*)keys_out, ids_out, \
                                                                                      box_out, (u64*)idctx_out, dcnt_out, sc, stats, abort_over, cells.src);  \
    } while (0)
#define BP_TGC_STEPS(K)                          \
    do {                                         \
        if (fmt.T <= 3) BP_TGC_LAUNCH(K, 1);     \
        else if (fmt.T <= 6) BP_TGC_LAUNCH(K, 2); \
        else BP_TGC_LAUNCH(K, 0);                \
    } while (0)
    if (ctx_kind == 0) BP_TGC_STEPS(0);
    else if (ctx_kind == 1) BP_TGC_STEPS(1);
    else BP_TGC_STEPS(2);
#undef BP_TGC_STEPS
#undef BP_TGC_LAUNCH
    return hipGetLastError();
}
// ids and box bytes from the per-entry context words ({id, context}: the context carries the entry's own code, two bits per axis)
__global__ __launch_bounds__(256) void ids_from_ctx_kernel(const u64* __restrict__ idctx, u32* __restrict__ ids_out, uint8_t* __restrict__ box_out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u64 w = idctx[i];
        const u32 c = (u32)w;
        ids_out[i] = (u32)(w >> 32);
        box_out[i] = (uint8_t)(((c >> 6) & 3u) | (((c >> 14) & 3u) << 2) | (((c >> 22) & 3u) << 4));
    }
}
hipError_t launch_ids_from_ctx(const uint64_t* idctx, uint32_t* ids_out, uint8_t* box_out, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    ids_from_ctx_kernel<<<grid_for(n, 256, 4096), 256, 0, stream>>>((const u64*)idctx, ids_out, box_out, n);
    return hipGetLastError();
}
hipError_t launch_tree_expand(int kind, uint64_t* words, uint32_t* ids_out, uint8_t* box_out, size_t n, const TreeCompact& fmt, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    BP_DISPATCH_KIND(kind, (tree_expand_kernel<K><<<grid_for(n, 256, 4096), 256, 0, stream>>>((u64*)words, ids_out, box_out, n, fmt)));
    return hipGetLastError();
}
int tgroup_capacity() { return TGC_CAP < TGRP_CAP ? TGC_CAP : TGRP_CAP; }  // (what every finish kernel of the grouped plan takes)
int tgroup_bin_bits() { return TGRP_BIN_BITS; }

hipError_t launch_tree_segment_finish(const uint64_t* keys_in, const uint32_t* ids_in, size_t m, int seg_shift,
                                      int payload_packed, uint64_t* keys_out, uint32_t* ids_out, uint8_t* box_out,
                                      DevScalars* sc, hipStream_t stream) {
    if (m == 0) return hipSuccess;
    const unsigned g = (unsigned)((m + SEGT_CHUNK - 1) / SEGT_CHUNK);
    if (payload_packed)
        tree_segment_finish_kernel<true><<<g, SEG_THREADS, 0, stream>>>((const u64*)keys_in, ids_in, m, seg_shift, (u64*)keys_out,
                                                                         ids_out, box_out, sc);
    else
        tree_segment_finish_kernel<false><<<g, SEG_THREADS, 0, stream>>>((const u64*)keys_in, ids_in, m, seg_shift, (u64*)keys_out,
                                                                          ids_out, box_out, sc);
    return hipGetLastError();
}

hipError_t launch_filter_pairs_aabb(int dim, const uint64_t* in, size_t n, const float* bounds, int shape, uint64_t* out, DevScalars* sc,
                                    hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned g = (unsigned)((n + 1023) / 1024);
    if (dim == 3) filter_pairs_aabb_kernel<3><<<g, 256, 0, stream>>>((const u64*)in, n, bounds, shape, (u64*)out, sc);
    else filter_pairs_aabb_kernel<2><<<g, 256, 0, stream>>>((const u64*)in, n, bounds, shape, (u64*)out, sc);
    return hipGetLastError();
}

hipError_t launch_pack_pairs(const uint32_t* pairs, size_t n, uint64_t* packed, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    pack_pairs_kernel<<<grid_for(n, 256, 4096), 256, 0, stream>>>((const uint2*)pairs, n, (u64*)packed);
    return hipGetLastError();
}

hipError_t launch_unpack_pairs(const uint64_t* packed, size_t n, uint32_t* pairs, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    unpack_pairs_kernel<<<grid_for(n, 256, 4096), 256, 0, stream>>>((const u64*)packed, n, (uint2*)pairs);
    return hipGetLastError();
}

}  // namespace bp

#ifdef SX_STAMPS
extern "C" int bp_debug_sx_stamps(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(bp::g_sx_stamps), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[16] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(bp::g_sx_stamps), z, sizeof z);
    }
    return (int)e;
}
extern "C" int bp_debug_sx_waves(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bp::g_sx_wave), 8192 * 4 * sizeof(unsigned long long));
}
#endif
