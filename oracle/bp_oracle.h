/*
 * bp_oracle.h -- CPU ORACLE for the broadphase Layer extend -> sort -> scan path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library
 * (broadphase_rs_amd/csrc, include/bp_layer.h) never links or calls anything here.
 *
 * It is a plain-C restatement of the reference algorithm (zvxryb/broadphase-rs 0.1.2,
 * Rust).  Every function cites the reference file:line it follows.
 *
 * PARITY PINNING STATUS: PINNED END TO END for the extend -> sort -> scan path.  The reference
 * crate cannot be compiled here (no rustc/cargo) and its test data under tests/data are Git-LFS
 * pointer stubs -- but the pointers carry the SHA-256 of every file.  tests/golden/ref_scene_gen.py
 * regenerates the reference's input scenes bit-exactly (its RNG and sampling restated; hashes equal
 * the LFS oids), and this oracle run through gen_validation_data's recipe on the 10 000-object scene
 * reproduces the reference's three validation goldens byte for byte: 0_layer_unsorted, 1_layer_sorted,
 * 2_layer_collisions all hash to their LFS oids (tests/test_reference_goldens.py).  Also pinned: the
 * in-source known-answer tests of the reference (src/index.rs:343-374 codec, src/geom.rs:696-706
 * quantisation), the structural assertions of tests/test_layer.rs:42-54, a second independent
 * restatement in Python (oracle/py_oracle.py) and a brute-force AABB no-false-negative check.
 * The region queries (test_box / test_ray / pick_ray) have no golden in the reference: they are
 * pinned by the agreement of two independent restatements only.  See DESIGN.md "Oracle".
 */
#ifndef BP_ORACLE_H
#define BP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* index kinds: src/index.rs:293-295  index_impl!{index: NAME, dim, bits, depth_bits, axis_bits} */
enum { ORC_INDEX32_2D = 0, ORC_INDEX64_2D = 1, ORC_INDEX64_3D = 2 };

typedef struct {
    uint32_t dim, bits, depth_bits, axis_bits;
} orc_kind_t;

int orc_kind(int kind_id, orc_kind_t *out);

/* codec: src/index.rs:134-207 */
uint64_t orc_encode_axis(int kind, uint32_t origin);
uint32_t orc_decode_axis(int kind, uint64_t origin);
/* src/index.rs:82-86, 93-122, 230-250 */
uint64_t orc_level_mask(int kind, uint32_t depth);
uint32_t orc_clamp_depth(int kind, uint32_t depth);
uint32_t orc_depth(int kind, uint64_t index);
uint64_t orc_set_depth(int kind, uint64_t index, uint32_t depth);
uint64_t orc_set_origin(int kind, uint64_t index, const uint32_t *origin);
void     orc_origin(int kind, uint64_t index, uint32_t *origin_out);
int      orc_same_cell_at_depth(int kind, uint64_t lhs, uint64_t rhs, uint32_t depth);
int      orc_overlaps(int kind, uint64_t a, uint64_t b);

/* geometry: src/geom.rs:121-128 (contains), :148-163 (to_local), :165-180 (to_global) */
int  orc_contains(uint32_t dim, const float *sys_min, const float *sys_max,
                  const float *bmin, const float *bmax);
void orc_to_local(uint32_t dim, const float *sys_min, const float *sys_max,
                  const float *g, uint32_t *out);
void orc_to_global(uint32_t dim, const float *sys_min, const float *sys_max,
                   const uint32_t *l, float *out);

/* index generation: src/geom.rs:189-238 (2-D), :247-304 (3-D).  Returns the number of indices
 * the reference would produce; writes at most cap of them to out (z outer, y, x inner). */
uint32_t orc_indices_depth(int kind, const uint32_t *lmin, const uint32_t *lmax, int has_min_depth,
                           uint32_t min_depth);
uint64_t orc_indices_at_depth(int kind, const uint32_t *lmin, const uint32_t *lmax, uint32_t depth,
                              uint64_t *out, uint64_t cap);

/* Layer: src/layer.rs:42-68 */
typedef struct orc_layer orc_layer_t;
typedef int (*orc_filter_fn)(uint32_t id, uint32_t id_, void *user);

orc_layer_t *orc_layer_new(int kind, uint32_t min_depth);
void         orc_layer_free(orc_layer_t *);
void         orc_layer_clear(orc_layer_t *);                          /* layer.rs:84-88  */
/* bounds: n rows of 2*dim floats (min[dim] then max[dim]); layer.rs:94-121 */
int          orc_layer_extend(orc_layer_t *, const float *sys_min, const float *sys_max,
                              const float *bounds, const uint32_t *ids, size_t n);
int          orc_layer_merge(orc_layer_t *, const orc_layer_t *other); /* layer.rs:127-138 */
/* merge() with the other layer's tree given as raw arrays (test helper for the sharded pipeline) */
int          orc_layer_append_entries(orc_layer_t *, const uint64_t *keys, const uint32_t *ids, size_t m);
void         orc_layer_sort(orc_layer_t *);                           /* layer.rs:159-165 */
void         orc_layer_par_sort(orc_layer_t *, int threads);          /* layer.rs:146-152 */
/* scan family: layer.rs:449-573.  Returned pointer is owned by the layer (interleaved id,id_). */
const uint32_t *orc_layer_scan(orc_layer_t *, size_t *n_pairs);
const uint32_t *orc_layer_scan_filtered(orc_layer_t *, orc_filter_fn f, void *user, size_t *n_pairs);
const uint32_t *orc_layer_par_scan(orc_layer_t *, int threads, size_t *n_pairs);
const uint32_t *orc_layer_par_scan_filtered(orc_layer_t *, int threads, orc_filter_fn f, void *user,
                                            size_t *n_pairs);
/* raw output of scan_impl over the whole (sorted) tree, before sort+dedup, in emission order */
const uint32_t *orc_layer_scan_raw(orc_layer_t *, size_t *n_raw, size_t *max_stack);

/* accessors (iter(), PartialEq: layer.rs:79-81, 576-586) */
size_t          orc_layer_len(const orc_layer_t *);
const uint64_t *orc_layer_keys(const orc_layer_t *);
const uint32_t *orc_layer_ids(const orc_layer_t *);
int             orc_layer_sorted(const orc_layer_t *);
uint32_t        orc_layer_min_depth(const orc_layer_t *);
size_t          orc_layer_num_invalid(const orc_layer_t *);

/* brute force: all (i<j) whose inclusive AABBs overlap (geom.rs:112-119); returns count,
 * writes up to cap pairs (interleaved i,j of input row indices). */
/* Layer::test_box (layer.rs:294-313, :252-279; BoxTestGeometry geom.rs:353-455): sorted unique ids whose
 * cells overlap the test box; has_max_depth = 0 stands for None.  Returns the count, < 0 on error
 * (-2: the reference would panic, -3: out too small) */
long orc_layer_test_box(orc_layer_t *, const float *sys_min, const float *sys_max, const float *test_min,
                        const float *test_max, int has_max_depth, uint32_t max_depth, uint32_t *out, size_t cap);

/* Layer::test_ray (layer.rs:327-351; RayTestGeometry geom.rs:460-689): ids whose cells the ray
 * origin + t * direction, t in [range_min, range_max] (clamped to the system bounds), passes through */
long orc_layer_test_ray(orc_layer_t *, const float *sys_min, const float *sys_max, const float *origin,
                        const float *direction, float range_min, float range_max, int has_max_depth, uint32_t max_depth,
                        uint32_t *out, size_t cap);

/* Layer::pick_ray (layer.rs:418-446; pick :364-404) with a built-in distance callback: entry distance of the ray
 * into the AABB bounds_by_id[id] (rows of 2*dim f32).  1 = hit (*dist, *id), 0 = none, < 0 = error */
int orc_layer_pick_ray_aabb(orc_layer_t *, const float *sys_min, const float *sys_max, const float *origin,
                            const float *direction, float max_dist, int has_max_depth, uint32_t max_depth,
                            const float *bounds_by_id, size_t n_rows, float *dist, uint32_t *id);

size_t orc_brute_force_overlaps(uint32_t dim, const float *bounds, size_t n, uint32_t *pairs, size_t cap);

/* number of hardware threads the par_* baseline can use */
int orc_hw_threads(void);

#ifdef __cplusplus
}
#endif
#endif
