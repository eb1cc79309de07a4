/*
 * bp_oracle.c -- CPU ORACLE (test infrastructure, not product code; see bp_oracle.h).
 *
 * Plain-C restatement of zvxryb/broadphase-rs 0.1.2:
 *   src/index.rs  (SpatialIndex codec), src/geom.rs (Bounds, to_local, IndexGenerator),
 *   src/layer.rs  (Layer clear/extend/merge/sort/scan family).
 * Rust integer arithmetic is restated with release-mode (wrapping) semantics; the
 * float -> u32 `as` cast with the saturating semantics of Rust >= 1.45 (NaN -> 0).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include "bp_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* index kinds: src/index.rs:293-295                                                           */
static const orc_kind_t KINDS[3] = {
    {2, 32, 4, 14}, /* Index32_2D */
    {2, 64, 5, 29}, /* Index64_2D */
    {3, 64, 5, 19}, /* Index64_3D */
};

int orc_kind(int kind_id, orc_kind_t *out) {
    if (kind_id < 0 || kind_id > 2) return -1;
    *out = KINDS[kind_id];
    return 0;
}

/* primitive_type truncation: u32 for Index32_2D, u64 otherwise (index.rs:125-126) */
static inline uint64_t prim(int kind, uint64_t v) { return KINDS[kind].bits == 32 ? (v & 0xffffffffull) : v; }

/* constants: src/index.rs:72-78 */
static inline uint32_t depth_bits(int k) { return KINDS[k].depth_bits; }
static inline uint32_t axis_bits(int k) { return KINDS[k].axis_bits; }
static inline uint32_t origin_bits(int k) { return KINDS[k].dim * KINDS[k].axis_bits; }
static inline uint32_t origin_shift(int k) { return 0 + KINDS[k].depth_bits; /* DEPTH_SHIFT + DEPTH_BITS */ }
static inline uint64_t depth_mask(int k) { return prim(k, ((1ull << depth_bits(k)) - 1ull) << 0); }
static inline uint64_t origin_mask(int k) { return prim(k, ((1ull << origin_bits(k)) - 1ull) << origin_shift(k)); }

/* src/index.rs:155-172 (2-D) and :193-207 (3-D) */
uint64_t orc_encode_axis(int kind, uint32_t origin) {
    const uint64_t a0 = (uint64_t)(origin >> (32 - axis_bits(kind)));
    if (KINDS[kind].dim == 2) {
        uint64_t axis00 = a0 & prim(kind, 0x000000000000ffffull);
        uint64_t axis01 = prim(kind, a0 << 0x10) & prim(kind, 0x0000ffff00000000ull);
        uint64_t axis1_ = axis00 | axis01;
        uint64_t axis10 = axis1_ & prim(kind, 0x000000ff000000ffull);
        uint64_t axis11 = prim(kind, axis1_ << 0x08) & prim(kind, 0x00ff000000ff0000ull);
        uint64_t axis2_ = axis10 | axis11;
        uint64_t axis20 = axis2_ & prim(kind, 0x000f000f000f000full);
        uint64_t axis21 = prim(kind, axis2_ << 0x04) & prim(kind, 0x0f000f000f000f00ull);
        uint64_t axis3_ = axis20 | axis21;
        uint64_t axis30 = axis3_ & prim(kind, 0x0303030303030303ull);
        uint64_t axis31 = prim(kind, axis3_ << 0x02) & prim(kind, 0x3030303030303030ull);
        uint64_t axis4_ = axis30 | axis31;
        uint64_t axis40 = axis4_ & prim(kind, 0x1111111111111111ull);
        uint64_t axis41 = prim(kind, axis4_ << 0x01) & prim(kind, 0x4444444444444444ull);
        return axis40 | axis41;
    } else {
        uint64_t axis00 = a0 & 00000000000000000000777ull;
        uint64_t axis01 = (a0 << 022) & 00000000000777000000000ull;
        uint64_t axis02 = (a0 << 044) & 00777000000000000000000ull;
        uint64_t axis1_ = axis00 | axis01 | axis02;
        uint64_t axis10 = axis1_ & 00007000000007000000007ull;
        uint64_t axis11 = (axis1_ << 006) & 01000000007000000007000ull;
        uint64_t axis12 = (axis1_ << 014) & 00000007000000007000000ull;
        uint64_t axis2_ = axis10 | axis11 | axis12;
        uint64_t axis20 = axis2_ & 01001001001001001001001ull;
        uint64_t axis21 = (axis2_ << 002) & 00010010010010010010010ull;
        uint64_t axis22 = (axis2_ << 004) & 00100100100100100100100ull;
        return axis20 | axis21 | axis22;
    }
}

/* src/index.rs:134-151 (2-D) and :176-190 (3-D) */
uint32_t orc_decode_axis(int kind, uint64_t origin) {
    origin = prim(kind, origin);
    if (KINDS[kind].dim == 2) {
        uint64_t axis00 = origin & prim(kind, 0x1111111111111111ull);
        uint64_t axis01 = (origin & prim(kind, 0x4444444444444444ull)) >> 0x01;
        uint64_t axis0_ = axis00 | axis01;
        uint64_t axis10 = axis0_ & prim(kind, 0x0303030303030303ull);
        uint64_t axis11 = (axis0_ & prim(kind, 0x3030303030303030ull)) >> 0x02;
        uint64_t axis1_ = axis10 | axis11;
        uint64_t axis20 = axis1_ & prim(kind, 0x000f000f000f000full);
        uint64_t axis21 = (axis1_ & prim(kind, 0x0f000f000f000f00ull)) >> 0x04;
        uint64_t axis2_ = axis20 | axis21;
        uint64_t axis30 = axis2_ & prim(kind, 0x000000ff000000ffull);
        uint64_t axis31 = (axis2_ & prim(kind, 0x00ff000000ff0000ull)) >> 0x08;
        uint64_t axis3_ = axis30 | axis31;
        uint64_t axis40 = axis3_ & prim(kind, 0x000000000000ffffull);
        uint64_t axis41 = (axis3_ & prim(kind, 0x0000ffff00000000ull)) >> 0x10;
        uint64_t axis4_ = axis40 | axis41;
        return (uint32_t)axis4_ << (32 - axis_bits(kind));
    } else {
        uint64_t axis00 = origin & 01001001001001001001001ull;
        uint64_t axis01 = (origin & 00010010010010010010010ull) >> 002;
        uint64_t axis02 = (origin & 00100100100100100100100ull) >> 004;
        uint64_t axis0_ = axis00 | axis01 | axis02;
        uint64_t axis10 = axis0_ & 00007000000007000000007ull;
        uint64_t axis11 = (axis0_ & 01000000007000000007000ull) >> 006;
        uint64_t axis12 = (axis0_ & 00000007000000007000000ull) >> 014;
        uint64_t axis1_ = axis10 | axis11 | axis12;
        uint64_t axis20 = axis1_ & 00000000000000000000777ull;
        uint64_t axis21 = (axis1_ & 00000000000777000000000ull) >> 022;
        uint64_t axis22 = (axis1_ & 00777000000000000000000ull) >> 044;
        uint64_t axis2_ = axis20 | axis21 | axis22;
        return (uint32_t)axis2_ << (32 - axis_bits(kind));
    }
}

/* src/index.rs:82-86 */
uint64_t orc_level_mask(int kind, uint32_t depth) {
    if (depth == 0) return 0;
    const uint32_t dim = KINDS[kind].dim;
    return prim(kind, ((1ull << (dim * depth)) - 1ull) << (origin_bits(kind) + origin_shift(kind) - dim * depth));
}

/* src/index.rs:93-95 */
uint32_t orc_clamp_depth(int kind, uint32_t depth) { return depth < axis_bits(kind) ? depth : axis_bits(kind); }

/* src/index.rs:99-102 */
uint32_t orc_depth(int kind, uint64_t index) { return (uint32_t)((index & depth_mask(kind)) >> 0); }

/* src/index.rs:106-112 */
uint64_t orc_set_depth(int kind, uint64_t index, uint32_t depth) {
    index &= ~depth_mask(kind);
    index |= depth_mask(kind) & ((uint64_t)orc_clamp_depth(kind, depth) << 0);
    return prim(kind, index);
}

/* src/index.rs:230-250 */
uint64_t orc_set_origin(int kind, uint64_t index, const uint32_t *o) {
    uint64_t origin = orc_encode_axis(kind, o[0]) | prim(kind, orc_encode_axis(kind, o[1]) << 1);
    if (KINDS[kind].dim == 3) origin |= orc_encode_axis(kind, o[2]) << 2;
    index &= ~origin_mask(kind);
    index |= origin_mask(kind) & prim(kind, origin << origin_shift(kind));
    return prim(kind, index);
}

/* src/index.rs:209-229 */
void orc_origin(int kind, uint64_t index, uint32_t *out) {
    uint64_t origin = (index & origin_mask(kind)) >> origin_shift(kind);
    out[0] = orc_decode_axis(kind, origin);
    out[1] = orc_decode_axis(kind, origin >> 1);
    if (KINDS[kind].dim == 3) out[2] = orc_decode_axis(kind, origin >> 2);
}

/* src/index.rs:120-122 */
int orc_same_cell_at_depth(int kind, uint64_t lhs, uint64_t rhs, uint32_t depth) {
    return ((lhs ^ rhs) & orc_level_mask(kind, depth)) == 0;
}

/* src/index.rs:116-118 */
int orc_overlaps(int kind, uint64_t a, uint64_t b) {
    uint32_t da = orc_depth(kind, a), db = orc_depth(kind, b);
    return orc_same_cell_at_depth(kind, a, b, da < db ? da : db);
}

/* ------------------------------------------------------------------------------------------ */
/* geometry                                                                                    */

/* src/geom.rs:121-128: written with negated >/< so NaN coordinates pass */
int orc_contains(uint32_t dim, const float *smin, const float *smax, const float *bmin, const float *bmax) {
    for (uint32_t i = 0; i < dim; ++i) {
        if (smin[i] > bmin[i] || smax[i] < bmax[i]) return 0;
    }
    return 1;
}

/* Rust `f32 as u32` (saturating, NaN -> 0) */
static inline uint32_t f32_as_u32(float f) {
    if (!(f > 0.0f)) return 0u; /* negative, -0, +0, NaN */
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

/* src/geom.rs:148-163 */
void orc_to_local(uint32_t dim, const float *smin, const float *smax, const float *g, uint32_t *out) {
    const float MIN_VALUE = 0.0f;               /* std::u32::MIN as f32 */
    const float MAX_VALUE = 4294967040.0f;      /* 0xffff_ff00u32 as f32 (exact) */
    const float RANGE = MAX_VALUE - MIN_VALUE;
    for (uint32_t i = 0; i < dim; ++i) {
        volatile float size = smax[i] - smin[i]; /* sizef(): geom.rs:97-102 */
        volatile float t = g[i] - smin[i];
        volatile float q = t / size;
        volatile float m = q * RANGE;
        volatile float r = m + MIN_VALUE;
        out[i] = f32_as_u32(r);
    }
}

/* src/geom.rs:165-180 */
void orc_to_global(uint32_t dim, const float *smin, const float *smax, const uint32_t *l, float *out) {
    const float MIN_VALUE = 0.0f;
    const float MAX_VALUE = 4294967040.0f;
    const float RANGE = MAX_VALUE - MIN_VALUE;
    for (uint32_t i = 0; i < dim; ++i) {
        volatile float size = smax[i] - smin[i];
        volatile float a = (float)l[i] - MIN_VALUE;
        volatile float b = a / RANGE;
        volatile float c = b * size;
        out[i] = smin[i] + c;
    }
}

/* src/geom.rs:48-61 */
static inline uint32_t scale_at_depth(uint32_t depth) { return 1u << (32 - depth); } /* depth != 0 */
static inline uint32_t truncate_to_depth(uint32_t x, uint32_t depth) {
    return depth == 0 ? x : (x & ~(scale_at_depth(depth) - 1u));
}
static inline uint32_t clz32(uint32_t v) { return v == 0 ? 32u : (uint32_t)__builtin_clz(v); }

/* src/geom.rs:189-200 / :247-258 : depth chosen by indices(min_depth) */
uint32_t orc_indices_depth(int kind, const uint32_t *lmin, const uint32_t *lmax, int has_min_depth,
                           uint32_t min_depth) {
    const uint32_t dim = KINDS[kind].dim;
    uint32_t max_axis = 0; /* u32::min_value() : geom.rs:40-46 */
    for (uint32_t i = 0; i < dim; ++i) {
        uint32_t s = (lmax[i] - lmin[i]) + 1u; /* sizei(): geom.rs:104-110 (wrapping) */
        if (s > max_axis) max_axis = s;
    }
    uint32_t depth = clz32(max_axis - 1u);
    if (has_min_depth && depth < min_depth) depth = min_depth;
    return orc_clamp_depth(kind, depth);
}

/* src/geom.rs:202-238 / :260-304 */
uint64_t orc_indices_at_depth(int kind, const uint32_t *lmin, const uint32_t *lmax, uint32_t depth,
                              uint64_t *out, uint64_t cap) {
    const uint32_t dim = KINDS[kind].dim;
    if (depth == 0) {
        if (cap > 0) out[0] = 0; /* Index::default() */
        return 1;
    }
    uint32_t mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (uint32_t i = 0; i < dim; ++i) {
        mn[i] = truncate_to_depth(lmin[i], depth);
        mx[i] = truncate_to_depth(lmax[i], depth);
    }
    const uint32_t step = scale_at_depth(depth);
    uint64_t n = 0;
    uint32_t z = mn[2];
    for (;;) { /* for dim == 2 this outer loop runs exactly once (mn[2] == mx[2] == 0) */
        uint32_t y = mn[1];
        for (;;) {
            uint32_t x = mn[0];
            for (;;) {
                uint32_t p[3] = {x, y, z};
                uint64_t idx = orc_set_origin(kind, orc_set_depth(kind, 0, depth), p);
                if (n < cap) out[n] = idx;
                ++n;
                if (x >= mx[0]) break;
                x += step;
            }
            if (y >= mx[1]) break;
            y += step;
        }
        if (dim == 2 || z >= mx[2]) break;
        z += step;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* sorting helpers (stand in for sort_unstable / par_sort_unstable: layer.rs:149,162,473,516)  */

typedef struct {
    uint64_t key;
    uint32_t id;
    uint32_t pad_;
} rec_t; /* (Index, ID) tuple: 16 bytes like the Rust in-memory tuple */

#define REC_LESS(a, b) ((a).key < (b).key || ((a).key == (b).key && (a).id < (b).id))
#define U64_LESS(a, b) ((a) < (b))

#define DEFINE_SORT(NAME, T, LESS)                                                                 \
    static void NAME##_insertion(T *a, size_t n) {                                                 \
        for (size_t i = 1; i < n; ++i) {                                                           \
            T v = a[i];                                                                            \
            size_t j = i;                                                                          \
            while (j > 0 && LESS(v, a[j - 1])) { a[j] = a[j - 1]; --j; }                           \
            a[j] = v;                                                                              \
        }                                                                                          \
    }                                                                                              \
    static void NAME##_sift(T *a, size_t start, size_t end) {                                      \
        size_t root = start;                                                                       \
        for (;;) {                                                                                 \
            size_t child = 2 * root + 1;                                                           \
            if (child >= end) break;                                                               \
            if (child + 1 < end && LESS(a[child], a[child + 1])) ++child;                          \
            if (LESS(a[root], a[child])) { T t = a[root]; a[root] = a[child]; a[child] = t; root = child; } \
            else break;                                                                            \
        }                                                                                          \
    }                                                                                              \
    static void NAME##_heap(T *a, size_t n) {                                                      \
        for (size_t s = n / 2; s-- > 0;) NAME##_sift(a, s, n);                                     \
        for (size_t e = n; e-- > 1;) { T t = a[0]; a[0] = a[e]; a[e] = t; NAME##_sift(a, 0, e); }  \
    }                                                                                              \
    static void NAME##_rec(T *a, size_t n, int depth_limit, int par) {                             \
        while (n > 24) {                                                                           \
            if (depth_limit-- == 0) { NAME##_heap(a, n); return; }                                 \
            size_t m = n / 2;                                                                      \
            if (LESS(a[m], a[0])) { T t = a[m]; a[m] = a[0]; a[0] = t; }                           \
            if (LESS(a[n - 1], a[0])) { T t = a[n - 1]; a[n - 1] = a[0]; a[0] = t; }               \
            if (LESS(a[n - 1], a[m])) { T t = a[n - 1]; a[n - 1] = a[m]; a[m] = t; }               \
            T pv = a[m];                                                                           \
            size_t i = 0, j = n - 1;                                                               \
            for (;;) {                                                                             \
                while (LESS(a[i], pv)) ++i;                                                        \
                while (LESS(pv, a[j])) --j;                                                        \
                if (i >= j) break;                                                                 \
                T t = a[i]; a[i] = a[j]; a[j] = t;                                                 \
                ++i; --j;                                                                          \
            }                                                                                      \
            size_t left = j + 1;                                                                   \
            if (par && n > (1u << 14)) {                                                           \
                T *pa = a; size_t pn = left; int dl = depth_limit;                                 \
                _Pragma("omp task firstprivate(pa, pn, dl)") NAME##_rec(pa, pn, dl, 1);            \
                a += left; n -= left;                                                              \
            } else if (left < n - left) {                                                          \
                NAME##_rec(a, left, depth_limit, par); a += left; n -= left;                       \
            } else {                                                                               \
                NAME##_rec(a + left, n - left, depth_limit, par); n = left;                        \
            }                                                                                      \
        }                                                                                          \
        NAME##_insertion(a, n);                                                                    \
    }                                                                                              \
    static void NAME(T *a, size_t n, int threads) {                                                \
        int dl = 2; for (size_t t = n; t > 1; t >>= 1) dl += 2;                                    \
        if (threads <= 1) { NAME##_rec(a, n, dl, 0); return; }                                     \
        _Pragma("omp parallel num_threads(threads)")                                               \
        {                                                                                          \
            _Pragma("omp single nowait") NAME##_rec(a, n, dl, 1);                                  \
        }                                                                                          \
    }

DEFINE_SORT(sort_rec, rec_t, REC_LESS)
DEFINE_SORT(sort_u64, uint64_t, U64_LESS)

int orc_hw_threads(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* growable vectors                                                                            */
typedef struct { rec_t *p; size_t n, cap; } recvec_t;
typedef struct { uint64_t *p; size_t n, cap; } u64vec_t;
typedef struct { uint32_t *p; size_t n, cap; } u32vec_t;

static int recvec_reserve(recvec_t *v, size_t want) {
    if (want <= v->cap) return 0;
    size_t c = v->cap ? v->cap : 64;
    while (c < want) c *= 2;
    rec_t *q = (rec_t *)realloc(v->p, c * sizeof(rec_t));
    if (!q) return -1;
    v->p = q; v->cap = c;
    return 0;
}
static int u64vec_reserve(u64vec_t *v, size_t want) {
    if (want <= v->cap) return 0;
    size_t c = v->cap ? v->cap : 64;
    while (c < want) c *= 2;
    uint64_t *q = (uint64_t *)realloc(v->p, c * sizeof(uint64_t));
    if (!q) return -1;
    v->p = q; v->cap = c;
    return 0;
}
static inline int u64vec_push(u64vec_t *v, uint64_t x) {
    if (v->n == v->cap && u64vec_reserve(v, v->n + 1)) return -1;
    v->p[v->n++] = x;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Layer: src/layer.rs:42-68                                                                   */
struct orc_layer {
    int kind;
    uint32_t min_depth;   /* persistent */
    recvec_t tree;        /* persistent: tree.0 */
    int sorted;           /* persistent: tree.1 */
    u64vec_t collisions;  /* temporary: packed (id << 32) | id_ */
    u32vec_t invalid;     /* temporary */
    /* views handed to the caller */
    uint64_t *view_keys; uint32_t *view_ids; size_t view_cap;
    uint32_t *view_pairs; size_t view_pairs_cap;
    size_t num_invalid_total;
};

orc_layer_t *orc_layer_new(int kind, uint32_t min_depth) {
    if (kind < 0 || kind > 2) return NULL;
    orc_layer_t *l = (orc_layer_t *)calloc(1, sizeof(*l));
    if (!l) return NULL;
    l->kind = kind;
    l->min_depth = min_depth;
    l->sorted = 1; /* LayerBuilder::build: tree = (Vec, true)  layer.rs:678-681 */
    return l;
}

void orc_layer_free(orc_layer_t *l) {
    if (!l) return;
    free(l->tree.p); free(l->collisions.p); free(l->invalid.p);
    free(l->view_keys); free(l->view_ids); free(l->view_pairs);
    free(l);
}

/* layer.rs:84-88 */
void orc_layer_clear(orc_layer_t *l) { l->tree.n = 0; l->sorted = 1; }

/* layer.rs:94-121 */
int orc_layer_extend(orc_layer_t *l, const float *smin, const float *smax, const float *bounds,
                     const uint32_t *ids, size_t n) {
    const uint32_t dim = KINDS[l->kind].dim;
    if (recvec_reserve(&l->tree, l->tree.n + n)) return -1; /* size_hint reserve: layer.rs:103-105 */
    uint64_t tmp[64];
    for (size_t o = 0; o < n; ++o) {
        const float *bmin = bounds + o * 2 * dim, *bmax = bmin + dim;
        if (!orc_contains(dim, smin, smax, bmin, bmax)) { /* layer.rs:108-111 */
            if (l->invalid.n == l->invalid.cap) {
                size_t c = l->invalid.cap ? l->invalid.cap * 2 : 64;
                uint32_t *q = (uint32_t *)realloc(l->invalid.p, c * sizeof(uint32_t));
                if (!q) return -1;
                l->invalid.p = q; l->invalid.cap = c;
            }
            l->invalid.p[l->invalid.n++] = ids[o];
            l->num_invalid_total++;
            continue;
        }
        uint32_t lmin[3], lmax[3];
        orc_to_local(dim, smin, smax, bmin, lmin);
        orc_to_local(dim, smin, smax, bmax, lmax);
        uint32_t depth = orc_indices_depth(l->kind, lmin, lmax, 1, l->min_depth); /* layer.rs:115 */
        uint64_t cnt = orc_indices_at_depth(l->kind, lmin, lmax, depth, tmp, 64);
        if (recvec_reserve(&l->tree, l->tree.n + cnt)) return -1;
        if (cnt <= 64) {
            for (uint64_t c = 0; c < cnt; ++c) {
                rec_t r = {tmp[c], ids[o], 0};
                l->tree.p[l->tree.n++] = r;
            }
        } else { /* SmallVec spilled in the reference (warn!): same output, heap path */
            uint64_t *big = (uint64_t *)malloc(cnt * sizeof(uint64_t));
            if (!big) return -1;
            orc_indices_at_depth(l->kind, lmin, lmax, depth, big, cnt);
            for (uint64_t c = 0; c < cnt; ++c) {
                rec_t r = {big[c], ids[o], 0};
                l->tree.p[l->tree.n++] = r;
            }
            free(big);
        }
        l->sorted = 0; /* layer.rs:119 */
    }
    return 0;
}

/* layer.rs:127-138 */
int orc_layer_merge(orc_layer_t *l, const orc_layer_t *other) {
    if (other->min_depth < l->min_depth) l->min_depth = other->min_depth; /* + warn! */
    if (recvec_reserve(&l->tree, l->tree.n + other->tree.n)) return -1;
    memcpy(l->tree.p + l->tree.n, other->tree.p, other->tree.n * sizeof(rec_t));
    l->tree.n += other->tree.n;
    l->sorted = 0;
    return 0;
}

int orc_layer_append_entries(orc_layer_t *l, const uint64_t *keys, const uint32_t *ids, size_t m) {
    if (recvec_reserve(&l->tree, l->tree.n + m)) return -1;
    for (size_t i = 0; i < m; ++i) {
        rec_t r = {keys[i], ids[i], 0};
        l->tree.p[l->tree.n++] = r;
    }
    l->sorted = 0;
    return 0;
}

/* layer.rs:159-165 */
void orc_layer_sort(orc_layer_t *l) {
    if (!l->sorted) { sort_rec(l->tree.p, l->tree.n, 1); l->sorted = 1; }
}
/* layer.rs:146-152 */
void orc_layer_par_sort(orc_layer_t *l, int threads) {
    if (!l->sorted) { sort_rec(l->tree.p, l->tree.n, threads); l->sorted = 1; }
}

/* layer.rs:550-573 ; stack: SmallVec<[(Index, ID); 256]> -- here a growable array */
static int scan_impl(int kind, const rec_t *tree, size_t n, u64vec_t *collisions, orc_filter_fn filter,
                     void *user, size_t *max_stack) {
    size_t cap = 256, sp = 0, ms = 0;
    rec_t *stack = (rec_t *)malloc(cap * sizeof(rec_t));
    if (!stack) return -1;
    for (size_t t = 0; t < n; ++t) {
        const uint64_t index = tree[t].key;
        const uint32_t id = tree[t].id;
        while (sp > 0) { /* layer.rs:557-562 */
            if (orc_overlaps(kind, index, stack[sp - 1].key)) break;
            --sp;
        }
        int dup = 0; /* layer.rs:563-565 */
        for (size_t s = 0; s < sp; ++s) if (stack[s].id == id) { dup = 1; break; }
        if (dup) continue;
        for (size_t s = 0; s < sp; ++s) { /* layer.rs:566-570 */
            const uint32_t id_ = stack[s].id;
            if (id != id_ && (!filter || filter(id, id_, user))) {
                if (u64vec_push(collisions, ((uint64_t)id << 32) | id_)) { free(stack); return -1; }
            }
        }
        if (sp == cap) {
            cap *= 2;
            rec_t *q = (rec_t *)realloc(stack, cap * sizeof(rec_t));
            if (!q) { free(stack); return -1; }
            stack = q;
        }
        stack[sp++] = tree[t]; /* layer.rs:571 */
        if (sp > ms) ms = sp;
    }
    free(stack);
    if (max_stack && ms > *max_stack) *max_stack = ms;
    return 0;
}

static void dedup_u64(u64vec_t *v) { /* Vec::dedup: layer.rs:474,517 */
    if (v->n == 0) return;
    size_t w = 1;
    for (size_t r = 1; r < v->n; ++r)
        if (v->p[r] != v->p[w - 1]) v->p[w++] = v->p[r];
    v->n = w;
}

static const uint32_t *pairs_view(orc_layer_t *l, size_t *n_pairs) {
    size_t n = l->collisions.n;
    if (n > l->view_pairs_cap) {
        free(l->view_pairs);
        l->view_pairs = (uint32_t *)malloc((n ? n : 1) * 2 * sizeof(uint32_t));
        l->view_pairs_cap = n;
    }
    for (size_t i = 0; i < n; ++i) {
        l->view_pairs[2 * i] = (uint32_t)(l->collisions.p[i] >> 32);
        l->view_pairs[2 * i + 1] = (uint32_t)(l->collisions.p[i] & 0xffffffffu);
    }
    *n_pairs = n;
    return l->view_pairs;
}

/* layer.rs:460-477 */
const uint32_t *orc_layer_scan_filtered(orc_layer_t *l, orc_filter_fn f, void *user, size_t *n_pairs) {
    orc_layer_sort(l);
    l->collisions.n = 0;
    l->invalid.n = 0;
    if (scan_impl(l->kind, l->tree.p, l->tree.n, &l->collisions, f, user, NULL)) { *n_pairs = 0; return NULL; }
    sort_u64(l->collisions.p, l->collisions.n, 1); /* sort_unstable: lexicographic (ID, ID) */
    dedup_u64(&l->collisions);
    return pairs_view(l, n_pairs);
}

/* layer.rs:449-453 */
const uint32_t *orc_layer_scan(orc_layer_t *l, size_t *n_pairs) {
    return orc_layer_scan_filtered(l, NULL, NULL, n_pairs);
}

const uint32_t *orc_layer_scan_raw(orc_layer_t *l, size_t *n_raw, size_t *max_stack) {
    orc_layer_sort(l);
    l->collisions.n = 0;
    size_t ms = 0;
    if (scan_impl(l->kind, l->tree.p, l->tree.n, &l->collisions, NULL, NULL, &ms)) { *n_raw = 0; return NULL; }
    if (max_stack) *max_stack = ms;
    return pairs_view(l, n_raw);
}

/* layer.rs:523-548: recursive halving at depth-min_depth cell boundaries.  The leaves are
 * independent; they are collected first, then run (in parallel) into per-leaf vectors which stand
 * in for the reference's per-thread CachedThreadLocal vectors (layer.rs:67, 530-531). */
typedef struct { size_t begin, end; } leaf_t;

static void par_split(const orc_layer_t *l, size_t threads, size_t begin, size_t end, leaf_t *leaves,
                      size_t *nleaves) {
    const size_t SPLIT_THRESHOLD = 64;
    const size_t n = end - begin;
    if (threads <= 1 || n <= SPLIT_THRESHOLD) {
        leaves[(*nleaves)++] = (leaf_t){begin, end};
        return;
    }
    const rec_t *tree = l->tree.p + begin;
    size_t i = n / 2;
    while (i < n) {
        if (!orc_same_cell_at_depth(l->kind, tree[i - 1].key, tree[i].key, l->min_depth)) break;
        ++i;
    }
    par_split(l, threads >> 1, begin, begin + i, leaves, nleaves);
    par_split(l, threads >> 1, begin + i, end, leaves, nleaves);
}

/* layer.rs:493-520 */
const uint32_t *orc_layer_par_scan_filtered(orc_layer_t *l, int threads, orc_filter_fn f, void *user,
                                            size_t *n_pairs) {
    if (threads < 1) threads = 1;
    orc_layer_par_sort(l, threads);
    l->collisions.n = 0;
    l->invalid.n = 0;

    size_t max_leaves = 2 * (size_t)threads + 2, nleaves = 0;
    leaf_t *leaves = (leaf_t *)malloc(max_leaves * sizeof(leaf_t));
    u64vec_t *outs = NULL;
    int err = 0;
    if (!leaves) { *n_pairs = 0; return NULL; }
    par_split(l, (size_t)threads, 0, l->tree.n, leaves, &nleaves); /* rayon::current_num_threads() */
    outs = (u64vec_t *)calloc(nleaves ? nleaves : 1, sizeof(u64vec_t));
    if (!outs) { free(leaves); *n_pairs = 0; return NULL; }

#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (long li = 0; li < (long)nleaves; ++li) {
        if (scan_impl(l->kind, l->tree.p + leaves[li].begin, leaves[li].end - leaves[li].begin, &outs[li], f,
                      user, NULL)) {
#pragma omp atomic write
            err = 1;
        }
    }
    size_t total = 0;
    for (size_t li = 0; li < nleaves; ++li) total += outs[li].n;
    if (!err && u64vec_reserve(&l->collisions, total)) err = 1;
    if (!err) {
        for (size_t li = 0; li < nleaves; ++li) { /* layer.rs:509-514 */
            memcpy(l->collisions.p + l->collisions.n, outs[li].p, outs[li].n * sizeof(uint64_t));
            l->collisions.n += outs[li].n;
        }
    }
    for (size_t li = 0; li < nleaves; ++li) free(outs[li].p);
    free(outs); free(leaves);
    if (err) { *n_pairs = 0; return NULL; }
    sort_u64(l->collisions.p, l->collisions.n, threads); /* par_sort_unstable: layer.rs:516 */
    dedup_u64(&l->collisions);                           /* layer.rs:517 */
    return pairs_view(l, n_pairs);
}

/* layer.rs:482-488 */
const uint32_t *orc_layer_par_scan(orc_layer_t *l, int threads, size_t *n_pairs) {
    return orc_layer_par_scan_filtered(l, threads, NULL, NULL, n_pairs);
}

/* ------------------------------------------------------------------------------------------ */
size_t orc_layer_len(const orc_layer_t *l) { return l->tree.n; }
int orc_layer_sorted(const orc_layer_t *l) { return l->sorted; }
uint32_t orc_layer_min_depth(const orc_layer_t *l) { return l->min_depth; }
size_t orc_layer_num_invalid(const orc_layer_t *l) { return l->num_invalid_total; }

static void refresh_views(orc_layer_t *l) {
    if (l->tree.n > l->view_cap || !l->view_keys) {
        free(l->view_keys); free(l->view_ids);
        size_t c = l->tree.n ? l->tree.n : 1;
        l->view_keys = (uint64_t *)malloc(c * sizeof(uint64_t));
        l->view_ids = (uint32_t *)malloc(c * sizeof(uint32_t));
        l->view_cap = c;
    }
    for (size_t i = 0; i < l->tree.n; ++i) { l->view_keys[i] = l->tree.p[i].key; l->view_ids[i] = l->tree.p[i].id; }
}
const uint64_t *orc_layer_keys(const orc_layer_t *l) { refresh_views((orc_layer_t *)l); return l->view_keys; }
const uint32_t *orc_layer_ids(const orc_layer_t *l) { refresh_views((orc_layer_t *)l); return l->view_ids; }

/* ------------------------------------------------------------------------------------------ */
/* region query: Layer::test / test_box with BoxTestGeometry (section 8(f) row 1)               */
/*                                                                                              */
/* cgmath ^0.17 (Cargo.toml dependency, absent from the tree, no lockfile) supplies             */
/* EuclideanSpace::midpoint for Bounds::center (geom.rs:130-132); its published definition is   */
/* `self + (other - self) / 2`, restated here in f32 in that order.                             */

typedef struct { float mn[3], mx[3]; } fbounds_t;

/* geom.rs:112-119 */
static int fbounds_overlaps(uint32_t dim, const fbounds_t *a, const float *bmin, const float *bmax) {
    for (uint32_t i = 0; i < dim; ++i)
        if (a->mn[i] > bmax[i] || a->mx[i] < bmin[i]) return 0;
    return 1;
}

/* index.rs:253-289: the 2^dim children in key order, or 0 at the depth limit */
static int subdivide_index(int kind, uint64_t cell, uint64_t *out) {
    orc_kind_t k; orc_kind(kind, &k);
    uint32_t depth = orc_depth(kind, cell);
    if (depth >= k.axis_bits) return 0;
    uint32_t shift = k.dim * k.axis_bits + k.depth_bits - k.dim * (depth + 1);
    for (uint32_t c = 0; c < (1u << k.dim); ++c)
        out[c] = orc_set_depth(kind, cell | ((uint64_t)c << shift), depth + 1);
    return 1;
}

/* Test geometry of one cell: BoxTestGeometry (geom.rs:353-363) or RayTestGeometry (geom.rs:460-470) */
typedef struct {
    fbounds_t cell_bounds;
    float range_min, range_max;    /* ray only */
} tgeom_t;

typedef struct {
    int kind; uint32_t dim;
    int is_ray;
    const float *tmin, *tmax;      /* box: test bounds */
    const float *origin, *direction; /* ray */
    int has_max_depth; uint32_t max_depth;
    u32vec_t *results; int err;
} regiontest_t;

static void results_push_all(regiontest_t *t, const rec_t *tree, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (t->results->n == t->results->cap) {
            size_t c = t->results->cap ? t->results->cap * 2 : 64;
            uint32_t *q = (uint32_t *)realloc(t->results->p, c * sizeof(uint32_t));
            if (!q) { t->err = -1; return; }
            t->results->p = q; t->results->cap = c;
        }
        t->results->p[t->results->n++] = tree[i].id;
    }
}

/* should_test with nearest = INFINITY: box geom.rs:409-412 / :449-452, ray :609-611 / :686-688 */
static int geom_should_test(const regiontest_t *t, const tgeom_t *g) {
    if (t->is_ray) return g->range_min < g->range_max && g->range_min < INFINITY;
    return fbounds_overlaps(t->dim, &g->cell_bounds, t->tmin, t->tmax);
}

/* TestGeometry::subdivide for child `c`: box geom.rs:383-407 / :418-446, ray :543-588 / :617-661 */
static void geom_child(const regiontest_t *t, const tgeom_t *g, uint32_t c, tgeom_t *out) {
    float center[3];
    for (uint32_t a = 0; a < t->dim; ++a)   /* Bounds::center = min.midpoint(max), geom.rs:130-132 */
        center[a] = g->cell_bounds.mn[a] + (g->cell_bounds.mx[a] - g->cell_bounds.mn[a]) / 2.0f;
    *out = *g;
    if (t->is_ray) {
        for (uint32_t a = 0; a < t->dim; ++a) {
            const float distance = (center[a] - t->origin[a]) / t->direction[a];
            const int side = (c >> a) & 1u;
            if (isfinite(distance)) {
                const int is_towards = (t->direction[a] > 0.0f) != side;
                if (is_towards) out->range_max = fminf(out->range_max, distance);   /* f32::min: NaN -> the other */
                else out->range_min = fmaxf(out->range_min, distance);
            } else if ((t->origin[a] > center[a]) != side) {
                out->range_min = INFINITY;
                out->range_max = -INFINITY;
            }
        }
    }
    for (uint32_t a = 0; a < t->dim; ++a) {
        if (c & (1u << a)) out->cell_bounds.mn[a] = center[a]; else out->cell_bounds.mx[a] = center[a];
    }
}

/* layer.rs:167-239; `nearest` stays INFINITY for Layer::test (its callback returns it unchanged,
 * layer.rs:268-271), so the visiting order (test_order) cannot change the result set and is dropped */
static void test_impl(regiontest_t *t, const rec_t *tree, size_t n, uint64_t cell, const tgeom_t *g) {
    if (n == 0 || !geom_should_test(t, g)) return;                                             /* :181-183 */
    if (tree[0].key < cell || !orc_overlaps(t->kind, cell, tree[n - 1].key)) { t->err = -2; return; }   /* :185-187 panic */
    uint32_t depth = orc_depth(t->kind, cell);
    if (t->has_max_depth && depth >= t->max_depth) { results_push_all(t, tree, n); return; }   /* :190-197 */
    uint64_t sub[8];
    if (!subdivide_index(t->kind, cell, sub)) { results_push_all(t, tree, n); return; }       /* :232-237 */
    const uint32_t nsub = 1u << t->dim;
    /* :200-215: split the slice at each child's first key (binary_search_by ... Less/Greater) */
    size_t start[9];
    const rec_t *rest = tree; size_t nrest = n, consumed = 0;
    for (uint32_t c = 0; c < nsub; ++c) {
        size_t lo = 0, hi = nrest;
        while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (rest[mid].key < sub[c]) lo = mid + 1; else hi = mid; }
        start[c] = consumed + lo;
        rest += lo; nrest -= lo; consumed += lo;
    }
    start[nsub] = n;
    results_push_all(t, tree, start[0]);                                                       /* head :216-219 */
    for (uint32_t c = 0; c < nsub; ++c) {
        tgeom_t child;
        geom_child(t, g, c, &child);
        test_impl(t, tree + start[c], start[c + 1] - start[c], sub[c], &child);
        if (t->err) return;
    }
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* Layer::test (layer.rs:252-279): sort, test_impl from the root cell, results.sort(); results.dedup() */
static long run_region_test(orc_layer_t *l, regiontest_t *t, const tgeom_t *root, uint32_t *out, size_t cap) {
    orc_layer_sort(l);                                                                          /* :260 */
    u32vec_t res = {0, 0, 0};
    t->results = &res; t->err = 0;
    test_impl(t, l->tree.p, l->tree.n, 0, root);                                                /* Index::default() :265 */
    if (t->err) { free(res.p); return t->err; }
    qsort(res.p, res.n, sizeof(uint32_t), cmp_u32);                                             /* :274 */
    size_t w = 0;
    for (size_t i = 0; i < res.n; ++i) if (i == 0 || res.p[i] != res.p[i - 1]) res.p[w++] = res.p[i];   /* :275 */
    long ret = (long)w;
    if (w > cap) ret = -3; else memcpy(out, res.p, w * sizeof(uint32_t));
    free(res.p);
    return ret;
}

/* Layer::test_box (layer.rs:294-313) */
long orc_layer_test_box(orc_layer_t *l, const float *sys_min, const float *sys_max, const float *tmin, const float *tmax,
                        int has_max_depth, uint32_t max_depth, uint32_t *out, size_t cap) {
    orc_kind_t k; orc_kind(l->kind, &k);
    regiontest_t t; memset(&t, 0, sizeof t);
    t.kind = l->kind; t.dim = k.dim; t.is_ray = 0; t.tmin = tmin; t.tmax = tmax;
    t.has_max_depth = has_max_depth; t.max_depth = max_depth;
    tgeom_t root; memset(&root, 0, sizeof root);
    for (uint32_t a = 0; a < k.dim; ++a) { root.cell_bounds.mn[a] = sys_min[a]; root.cell_bounds.mx[a] = sys_max[a]; }
    return run_region_test(l, &t, &root, out, cap);
}

/* Layer::test_ray (layer.rs:327-351) with RayTestGeometry::with_system_bounds (geom.rs:511-537) */
long orc_layer_test_ray(orc_layer_t *l, const float *sys_min, const float *sys_max, const float *origin,
                        const float *direction, float range_min, float range_max, int has_max_depth, uint32_t max_depth,
                        uint32_t *out, size_t cap) {
    orc_kind_t k; orc_kind(l->kind, &k);
    regiontest_t t; memset(&t, 0, sizeof t);
    t.kind = l->kind; t.dim = k.dim; t.is_ray = 1; t.origin = origin; t.direction = direction;
    t.has_max_depth = has_max_depth; t.max_depth = max_depth;
    tgeom_t root; memset(&root, 0, sizeof root);
    for (uint32_t a = 0; a < k.dim; ++a) {
        root.cell_bounds.mn[a] = sys_min[a]; root.cell_bounds.mx[a] = sys_max[a];
        const float distance_0 = (sys_min[a] - origin[a]) / direction[a];
        const float distance_1 = (sys_max[a] - origin[a]) / direction[a];
        const int is_forward = direction[a] > 0.0f;
        const float d0 = is_forward ? distance_0 : distance_1, d1 = is_forward ? distance_1 : distance_0;
        if (isfinite(d0)) range_min = fmaxf(range_min, d0);
        if (isfinite(d1)) range_max = fminf(range_max, d1);
    }
    root.range_min = range_min; root.range_max = range_max;
    return run_region_test(l, &t, &root, out, cap);
}

/* ------------------------------------------------------------------------------------------ */
/* nearest-hit query: Layer::pick / pick_ray (layer.rs:364-446) with a BUILT-IN distance callback  */
/* (ray vs. the AABB of the object, rows of bounds indexed by id) in place of the caller's closure  */

/* RayTestGeometry::test_order (2-D geom.rs:591-607, 3-D :664-684) */
static void ray_test_order(uint32_t dim, const float *direction, uint32_t *order) {
    float ab[3] = {0, 0, 0};
    for (uint32_t a = 0; a < dim; ++a) ab[a] = fabsf(direction[a]);
    uint32_t axes[3] = {0, 1, 2};
    if (dim == 2) {
        if (ab[0] <= ab[1]) { axes[0] = 0; axes[1] = 1; } else { axes[0] = 1; axes[1] = 0; }
    } else {
        if (ab[0] <= ab[1] && ab[0] <= ab[2]) {
            if (ab[1] <= ab[2]) { axes[0] = 0; axes[1] = 1; axes[2] = 2; } else { axes[0] = 0; axes[1] = 2; axes[2] = 1; }
        } else if (ab[1] <= ab[2]) {
            if (ab[0] <= ab[2]) { axes[0] = 1; axes[1] = 0; axes[2] = 2; } else { axes[0] = 1; axes[1] = 2; axes[2] = 0; }
        } else {
            if (ab[0] <= ab[1]) { axes[0] = 2; axes[1] = 0; axes[2] = 1; } else { axes[0] = 2; axes[1] = 1; axes[2] = 0; }
        }
    }
    for (uint32_t src = 0; src < (1u << dim); ++src) {
        uint32_t dst = 0;
        for (uint32_t k = 0; k < dim; ++k) {
            const uint32_t bit = (((src >> k) & 1u) != 0) == (direction[axes[k]] >= 0.0f);
            dst |= bit << axes[k];
        }
        order[src] = dst;
    }
}

/* the built-in GetDist: slab test of origin + t * direction, t >= 0, against the inclusive box; entry distance
 * (0 when the origin is inside), INFINITY on a miss.  Written once here and once in the kernel, same f32 steps. */
static float ray_aabb_distance(uint32_t dim, const float *o, const float *d, const float *bmin, const float *bmax) {
    float tmin = 0.0f, tmax = INFINITY;
    for (uint32_t a = 0; a < dim; ++a) {
        if (d[a] == 0.0f) {
            if (o[a] < bmin[a] || o[a] > bmax[a]) return INFINITY;
        } else {
            float t0 = (bmin[a] - o[a]) / d[a], t1 = (bmax[a] - o[a]) / d[a];
            if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
            tmin = fmaxf(tmin, t0);
            tmax = fminf(tmax, t1);
        }
    }
    return tmin <= tmax ? tmin : INFINITY;
}

typedef struct {
    regiontest_t t;
    const float *bounds_by_id; size_t n_rows;
    uint32_t order[8];
    int have; uint32_t result;
} picktest_t;

/* the closure of Layer::pick (layer.rs:386-401) around the built-in distance; `processed` (a set that makes the
 * user's closure run once per id) is dropped: the built-in distance is pure, a repeat cannot beat `nearest` */
static float pick_callback(picktest_t *p, float nearest, uint32_t id) {
    if ((size_t)id >= p->n_rows) { p->t.err = -4; return INFINITY; }
    const float *row = p->bounds_by_id + (size_t)id * 2 * p->t.dim;
    const float dist = ray_aabb_distance(p->t.dim, p->t.origin, p->t.direction, row, row + p->t.dim);
    if (isfinite(dist)) {
        if (dist < nearest) { p->result = id; p->have = 1; }
        return dist;
    }
    return INFINITY;
}
static float pick_fold(picktest_t *p, const rec_t *tree, size_t n, float nearest) {   /* layer.rs:193-196 */
    for (size_t i = 0; i < n; ++i) nearest = fminf(pick_callback(p, nearest, tree[i].id), nearest);
    return nearest;
}
/* test_impl (layer.rs:167-239) with nearest-hit pruning: should_test(nearest) geom.rs:686-688 */
static float pick_impl(picktest_t *p, const rec_t *tree, size_t n, uint64_t cell, const tgeom_t *g, float nearest) {
    regiontest_t *t = &p->t;
    if (n == 0 || !(g->range_min < g->range_max && g->range_min < nearest)) return nearest;
    if (tree[0].key < cell || !orc_overlaps(t->kind, cell, tree[n - 1].key)) { t->err = -2; return nearest; }
    uint32_t depth = orc_depth(t->kind, cell);
    if (t->has_max_depth && depth >= t->max_depth) return pick_fold(p, tree, n, nearest);
    uint64_t sub[8];
    if (!subdivide_index(t->kind, cell, sub)) return pick_fold(p, tree, n, nearest);
    const uint32_t nsub = 1u << t->dim;
    size_t start[9];
    const rec_t *rest = tree; size_t nrest = n, consumed = 0;
    for (uint32_t c = 0; c < nsub; ++c) {
        size_t lo = 0, hi = nrest;
        while (lo < hi) { size_t mid = lo + (hi - lo) / 2; if (rest[mid].key < sub[c]) lo = mid + 1; else hi = mid; }
        start[c] = consumed + lo;
        rest += lo; nrest -= lo; consumed += lo;
    }
    start[nsub] = n;
    nearest = pick_fold(p, tree, start[0], nearest);                                           /* :216-219 */
    for (uint32_t k = 0; k < nsub; ++k) {                                                      /* :224-232, in test_order */
        const uint32_t c = p->order[k];
        tgeom_t child;
        geom_child(t, g, c, &child);
        nearest = pick_impl(p, tree + start[c], start[c + 1] - start[c], sub[c], &child, nearest);
        if (t->err) return nearest;
    }
    return nearest;
}

/* Layer::pick_ray (layer.rs:418-446) with the built-in ray/AABB distance.  Returns 1 and (*dist, *id) when
 * something was hit within max_dist, 0 when nothing was, < 0 on error. */
int orc_layer_pick_ray_aabb(orc_layer_t *l, const float *sys_min, const float *sys_max, const float *origin,
                            const float *direction, float max_dist, int has_max_depth, uint32_t max_depth,
                            const float *bounds_by_id, size_t n_rows, float *dist, uint32_t *id) {
    orc_kind_t k; orc_kind(l->kind, &k);
    picktest_t p; memset(&p, 0, sizeof p);
    p.t.kind = l->kind; p.t.dim = k.dim; p.t.is_ray = 1; p.t.origin = origin; p.t.direction = direction;
    p.t.has_max_depth = has_max_depth; p.t.max_depth = max_depth;
    p.bounds_by_id = bounds_by_id; p.n_rows = n_rows;
    ray_test_order(k.dim, direction, p.order);
    tgeom_t root; memset(&root, 0, sizeof root);
    float range_min = 0.0f, range_max = max_dist;                                              /* :434-439 */
    for (uint32_t a = 0; a < k.dim; ++a) {
        root.cell_bounds.mn[a] = sys_min[a]; root.cell_bounds.mx[a] = sys_max[a];
        const float distance_0 = (sys_min[a] - origin[a]) / direction[a];
        const float distance_1 = (sys_max[a] - origin[a]) / direction[a];
        const int is_forward = direction[a] > 0.0f;
        const float d0 = is_forward ? distance_0 : distance_1, d1 = is_forward ? distance_1 : distance_0;
        if (isfinite(d0)) range_min = fmaxf(range_min, d0);
        if (isfinite(d1)) range_max = fminf(range_max, d1);
    }
    root.range_min = range_min; root.range_max = range_max;
    orc_layer_sort(l);                                                                          /* :374 */
    const float d = pick_impl(&p, l->tree.p, l->tree.n, 0, &root, max_dist);
    if (p.t.err) return p.t.err;
    if (!p.have) return 0;
    *dist = d; *id = p.result;
    return 1;
}

/* geom.rs:112-119 (Bounds::overlaps, inclusive) over all i<j */
size_t orc_brute_force_overlaps(uint32_t dim, const float *b, size_t n, uint32_t *pairs, size_t cap) {
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        const float *imin = b + i * 2 * dim, *imax = imin + dim;
        for (size_t j = i + 1; j < n; ++j) {
            const float *jmin = b + j * 2 * dim, *jmax = jmin + dim;
            int ov = 1;
            for (uint32_t a = 0; a < dim; ++a)
                if (imin[a] > jmax[a] || imax[a] < jmin[a]) { ov = 0; break; }
            if (ov) {
                if (cnt < cap) { pairs[2 * cnt] = (uint32_t)i; pairs[2 * cnt + 1] = (uint32_t)j; }
                ++cnt;
            }
        }
    }
    return cnt;
}
