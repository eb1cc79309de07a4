/*
 * bp_layer.h -- C ABI of the MI355X-native broadphase Layer (extend -> sort -> scan).
 *
 * This is the drop-in boundary for the hot path of zvxryb/broadphase-rs: the reference has no FFI
 * layer of its own -- its public surface for this path is the Rust type `Layer<Index, ID>`
 * (re-exported at src/lib.rs:80-82).  Each entry point below replaces one `Layer` /
 * `LayerBuilder` method; the Rust shim a maintainer would add on top is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns BP_OK (0) or a negative bp_status; bp_status_string() names it and
 *     bp_layer_last_error() gives the detailed message of the last failure on a handle;
 *   - a handle is NOT thread-safe (it mirrors `&mut self`); each handle owns one HIP stream and
 *     all its device and pinned-host buffers; every call is complete when it returns;
 *   - STREAM ORDERING: the handle's stream is non-blocking (it does not synchronise with the legacy default
 *     stream).  Results are complete on return, so any stream may read them afterwards.  DEVICE INPUTS,
 *     however, are read on the handle's stream: if kernels that write them may still be in flight on another
 *     stream, call bp_layer_wait_stream(layer, that_stream) first (or synchronise that stream) -- and they may
 *     STILL BE BEING READ when bp_layer_extend_device returns (in the per-frame pattern the call returns as soon
 *     as the frame's entry count is known, with the pass that writes the entries still running behind it): do not
 *     overwrite or free them before the next call on the handle that returns a result (a scan, bp_layer_tree*,
 *     a query), or order the writer behind bp_layer_stream().  Host inputs (bp_layer_extend_host, and
 *     bp_layer_extend with host pointers) are copied before the call returns and are the caller's again at once;
 *   - returned pointers are BORROWED: valid until the next mutating call on the same handle
 *     (the reference returns `&'a Vec<(ID, ID)>` tied to `&'a mut self`, src/layer.rs:449-451);
 *   - ID is u32, by decision.  The reference's `ObjectID` (src/traits.rs:6-16) admits any `Copy + Hash + Ord` type; the only
 *     ID the reference itself instantiates on this path is u32 (`data/src/lib.rs:16`, the example's entity ids), and every
 *     byte of the pair side -- 57 % of a frame -- doubles with u64 ids (pairs are sorted as one packed 64-bit word here).  A host
 *     with wider ids keeps a u32 -> ID table on its side (ids are dense per frame in the reference's use).  Two id ranges cost
 *     time, never correctness: ids >= 2^27 (the 5 spare payload bits then cannot carry the box codes through the tree sort, so
 *     duplicates are removed after the pair sort as in the reference: about 2 x the frame time -- except in the per-frame pattern,
 *     where from the third frame on the tree travels as compact records with a code field of their own and an id field that holds
 *     id - min(id): ids of any magnitude whose range is below 2^27 then cost about 10 % more than small ones) and ids that repeat
 *     across objects (the reference's skip rule, src/layer.rs:563-565).  Index values travel as u64 for every index kind
 *     (Index32_2D is zero-extended);
 *   - no CPU fallback exists: without a usable HIP device every call fails with BP_ERR_NO_DEVICE.
 */
#ifndef BP_LAYER_H
#define BP_LAYER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum bp_status {
    BP_OK = 0,
    BP_ERR_INVALID_ARG = -1,
    BP_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at init            */
    BP_ERR_HIP = -3,         /* a HIP call failed (see bp_layer_last_error)            */
    BP_ERR_OOM = -4,         /* device or pinned-host allocation failed                */
    BP_ERR_TOO_LARGE = -5,   /* an intermediate size exceeds what the kernels index    */
    BP_ERR_KERNEL = -6,      /* a kernel reported an internal error (bounded spin hit) */
    BP_ERR_COMM = -7         /* multi-GPU exchange failed                              */
} bp_status;

/* SpatialIndex implementations: src/index.rs:293-295 */
typedef enum bp_index_kind {
    BP_INDEX32_2D = 0, /* 4 depth bits + 2 x 14 origin bits in a u32 */
    BP_INDEX64_2D = 1, /* 5 depth bits + 2 x 29 origin bits in a u64 */
    BP_INDEX64_3D = 2  /* 5 depth bits + 3 x 19 origin bits in a u64 */
} bp_index_kind;

typedef struct bp_layer bp_layer_t; /* opaque: Layer<Index, u32>  (src/layer.rs:42-68) */

/* scan_filtered's `F: FnMut(ID, ID) -> bool` (src/layer.rs:460-464): return non-zero to keep */
typedef int (*bp_filter_fn)(uint32_t id, uint32_t id_, void *user);

const char *bp_status_string(int status);
const char *bp_layer_last_error(const bp_layer_t *layer);

/* ---- construction: LayerBuilder (src/layer.rs:619-696) ------------------------------------ */
/* LayerBuilder::new().with_min_depth(min_depth).with_index_capacity(index_capacity)
 *   .with_collision_capacity(collision_capacity).build::<Index, u32>()
 * `device` is the HIP device ordinal the layer lives on (-1: the current device).
 * with_test_capacity (src/layer.rs:665) has no counterpart: the batched queries size their own scratch. */
int bp_layer_create(bp_layer_t **out, int index_kind, uint32_t min_depth, size_t index_capacity,
                    size_t collision_capacity, int device);
int bp_layer_destroy(bp_layer_t *layer);
/* impl Clone (src/layer.rs:595-616): copies min_depth + tree, drops temporaries */
int bp_layer_clone(const bp_layer_t *layer, bp_layer_t **out);
/* impl PartialEq (src/layer.rs:576-586): min_depth, entries in current order, sorted flag */
int bp_layer_equals(const bp_layer_t *a, const bp_layer_t *b, int *equal);

/* Orders everything the layer enqueues from now on after the work currently enqueued on `producer_stream`
 * (a hipStream_t; NULL = the legacy default stream): event record on the producer + hipStreamWaitEvent on the
 * layer's stream, no host wait.  Required before a call that takes device pointers whose contents are still
 * being produced on another stream (see STREAM ORDERING above). */
int bp_layer_wait_stream(bp_layer_t *layer, void *producer_stream);
/* the layer's own hipStream_t (for callers that want to enqueue consumers behind it) */
int bp_layer_stream(const bp_layer_t *layer, void **stream);

/* ---- Layer methods on the hot path --------------------------------------------------------- */
/* Layer::clear (src/layer.rs:84-88) */
int bp_layer_clear(bp_layer_t *layer);

/* Layer::extend (src/layer.rs:94-121).  system bounds: sys_min[dim], sys_max[dim].
 * bounds: n rows of 2*dim f32 (min[dim] then max[dim], inclusive); ids: n u32.
 * Objects not contained in the system bounds are skipped silently, like the reference.
 * _host takes host pointers (copied H2D on the layer's stream); _device takes pointers that
 * are already resident on the layer's device (no copy); bp_layer_extend detects which. */
int bp_layer_extend(bp_layer_t *layer, const float *sys_min, const float *sys_max, const float *bounds,
                    const uint32_t *ids, size_t n);
int bp_layer_extend_host(bp_layer_t *layer, const float *sys_min, const float *sys_max, const float *bounds,
                         const uint32_t *ids, size_t n);
int bp_layer_extend_device(bp_layer_t *layer, const float *sys_min, const float *sys_max,
                           const float *d_bounds, const uint32_t *d_ids, size_t n);

/* Layer::merge (src/layer.rs:127-138); `other` must live on the same device */
int bp_layer_merge(bp_layer_t *layer, const bp_layer_t *other);

/* Layer::sort / Layer::par_sort (src/layer.rs:146-165): one implementation, two names.
 * Completion is DEFERRED: the call may return with the sort only enqueued on the layer's stream and its validation (does
 * every prefix group fit the LDS finish?) still open.  Every accessor of the tree (bp_layer_tree*, the scans, queries,
 * merge, clone, equals) and bp_layer_begin_entries settle it first -- they wait for the stream, read the sort's error word
 * and, should the check have failed, repair the tree with the full-width sort -- so the deferral is not observable through
 * this ABI; an error of the sort (BP_ERR_KERNEL) is then reported by THAT call.  Callers that enqueue their own work on
 * another stream must order it behind bp_layer_stream() themselves. */
int bp_layer_sort(bp_layer_t *layer);
int bp_layer_par_sort(bp_layer_t *layer);

/* Layer::scan / par_scan (src/layer.rs:449-453, 482-488): sorted, duplicate-free list of
 * (id, id_) pairs as interleaved u32.  *pairs is host-visible (pinned) memory. */
int bp_layer_scan(bp_layer_t *layer, const uint32_t **pairs, size_t *n_pairs);
int bp_layer_par_scan(bp_layer_t *layer, const uint32_t **pairs, size_t *n_pairs);
/* same result left in HBM: *d_pairs is a device pointer (no D2H copy) */
int bp_layer_scan_device(bp_layer_t *layer, const uint32_t **d_pairs, size_t *n_pairs);

/* Layer::scan_filtered / par_scan_filtered (src/layer.rs:460-477, 493-520).  The filter sees
 * every raw pair before duplicate removal, in scan_impl's emission order (src/layer.rs:566-570):
 * the raw pair stream is produced on the device in exactly that order and filtered on the host. */
int bp_layer_scan_filtered(bp_layer_t *layer, bp_filter_fn filter, void *user, const uint32_t **pairs,
                           size_t *n_pairs);
int bp_layer_par_scan_filtered(bp_layer_t *layer, bp_filter_fn filter, void *user, const uint32_t **pairs,
                               size_t *n_pairs);

/* scan_filtered with the narrow-phase box test as a built-in DEVICE predicate (SURVEY.md 8(f) #3;
 * the example runs its narrow phase on the host over the candidate list, examples/main.rs:459-479):
 * same result as scan_filtered(|id, id_| bounds[id].overlaps(bounds[id_])) with Bounds::overlaps
 * of src/geom.rs:112-119 (inclusive).  bounds_by_id: n_rows rows of 2*dim f32 (min then max), row r
 * = the box of object id r (host or device memory, detected); every id in the layer must be
 * < n_rows.  The predicate runs inside the scan, before any pair is sorted. */
int bp_layer_scan_filtered_aabb(bp_layer_t *layer, const float *bounds_by_id, size_t n_rows, const uint32_t **pairs,
                                size_t *n_pairs);
int bp_layer_scan_filtered_aabb_device(bp_layer_t *layer, const float *bounds_by_id, size_t n_rows,
                                       const uint32_t **d_pairs, size_t *n_pairs);
/* the same with the example's own narrow phase (examples/main.rs:459-474): spheres_by_id = n_rows rows
 * of dim+1 f32 (centre, radius); a pair is kept unless |c1 - c0| > r0 + r1, with the magnitude
 * computed as the example does in f32 (sqrt of the left-to-right sum of squares); NaN keeps. */
int bp_layer_scan_filtered_sphere(bp_layer_t *layer, const float *spheres_by_id, size_t n_rows, const uint32_t **pairs,
                                  size_t *n_pairs);
int bp_layer_scan_filtered_sphere_device(bp_layer_t *layer, const float *spheres_by_id, size_t n_rows,
                                         const uint32_t **d_pairs, size_t *n_pairs);

/* Raw output of scan_impl (src/layer.rs:550-573) before sort+dedup, in emission order; used by
 * parity tests and by scan_filtered.  Host-visible, interleaved (id, id_). */
int bp_layer_scan_raw(bp_layer_t *layer, const uint32_t **pairs, size_t *n_raw);

/* Layer::iter (src/layer.rs:79-81) / serde view of the persistent state (src/layer.rs:49-50):
 * entries in current order as SoA: keys[m] (u64, Index value zero-extended), ids[m]. */
int bp_layer_tree(bp_layer_t *layer, const uint64_t **keys, const uint32_t **ids, size_t *m, int *sorted);
int bp_layer_tree_device(bp_layer_t *layer, const uint64_t **d_keys, const uint32_t **d_ids, size_t *m,
                         int *sorted);
/* Deserialize counterpart: replace the persistent state (host or device pointers, detected). */
int bp_layer_set_tree(bp_layer_t *layer, uint32_t min_depth, const uint64_t *keys, const uint32_t *ids,
                      size_t m, int sorted);
/* Append raw entries (what merge does with another layer's tree); used by the multi-GPU exchange. */
int bp_layer_append_entries(bp_layer_t *layer, const uint64_t *keys, const uint32_t *ids, size_t m);

int bp_layer_len(const bp_layer_t *layer, size_t *m);
int bp_layer_is_sorted(const bp_layer_t *layer, int *sorted);
int bp_layer_min_depth(const bp_layer_t *layer, uint32_t *min_depth);
int bp_layer_index_kind(const bp_layer_t *layer, int *index_kind);
int bp_layer_device(const bp_layer_t *layer, int *device);

/* ---- multi-GPU support (one process per GPU; the exchange itself is RCCL via the host side) --- */
/* Scan variant for a Morton-range shard: entries whose depth is < coarse_depth are replicated on
 * every rank; pairs whose deeper entry is such a coarse entry must be emitted by exactly ONE rank, so that
 * the union over ranks is the single-GPU raw set.  emit_coarse (this and every *_shard_* call below):
 *   0  this rank emits none of them,   1  all of them (one designated rank),
 *   2  those whose deeper entry's id lies in [id_lo, id_hi] as set by bp_layer_set_coarse_emit_range -- the ranks' ranges
 *      partition the ids, so the coarse-coarse pairs of a heavy-tailed scene spread over the ranks instead of landing on one. */
int bp_layer_scan_shard_device(bp_layer_t *layer, uint32_t coarse_depth, int emit_coarse,
                               const uint32_t **d_pairs, size_t *n_pairs);
int bp_layer_set_coarse_emit_range(bp_layer_t *layer, uint32_t id_lo, uint32_t id_hi);
/* Sort + dedup an arbitrary device-resident list of n (id, id_) pairs (interleaved u32) in place
 * semantics of collisions.sort_unstable(); collisions.dedup() (src/layer.rs:473-474, 516-517). */
int bp_layer_sort_unique_pairs_device(bp_layer_t *layer, const uint32_t *d_pairs_in, size_t n,
                                      const uint32_t **d_pairs, size_t *n_pairs);

/* Building blocks of the Morton-range sharded pipeline (host side: broadphase_rs_amd/dist.py).
 * All results are borrowed device pointers, valid until the next mutating call on the handle. */
enum { BP_FACT_ORDER_OK = 1u, BP_FACT_IDS_DISTINCT = 2u, BP_FACT_IDS_CARRY_BOX = 4u, BP_FACT_BOX_ALL_KNOWN = 8u };
/* append raw entries and assert facts the caller can prove about them (they only select faster
 * kernels): ORDER_OK = among equal index values the ids are non-decreasing in the resulting
 * array order; IDS_DISTINCT = no id occurs on two objects; IDS_CARRY_BOX = bits 27..31 of every id
 * hold the entry's cell-box code as packed by bp_layer_partition_entries_device (they are split
 * off again here); BOX_ALL_KNOWN = none of those codes is "unknown".  Facts are AND-ed with the
 * layer's. */
int bp_layer_append_entries_assume(bp_layer_t *layer, const uint64_t *keys, const uint32_t *ids, size_t m,
                                   uint32_t facts);
/* What a layer knows about its entries; travels with them across the exchange.  OR fields may
 * contain extra bits and AND fields fewer (a superset description is always safe). */
typedef struct bp_facts {
    uint64_t key_or, key_and;   /* OR / AND of the index values               */
    uint32_t id_or, id_and;     /* OR / AND of the ids (without box-code bits) */
    uint32_t id_min, id_max;
    uint32_t depth_mask;        /* bit d set <=> an entry of depth d may exist */
    uint32_t flags;             /* BP_FACT_*                                   */
} bp_facts;
int bp_layer_get_facts(const bp_layer_t *layer, bp_facts *out);
/* Replace the tree by m entries the caller writes itself (e.g. as the output of an all-to-all):
 * begin clears the layer and returns device pointers to keys[m] / ids[m]; commit publishes them
 * with the facts the caller vouches for.  With BP_FACT_IDS_CARRY_BOX the ids stay packed until
 * the next sort splits the codes off (any other access to the entries unpacks first). */
int bp_layer_begin_entries(bp_layer_t *layer, size_t m, uint64_t **d_keys, uint32_t **d_ids);
int bp_layer_commit_entries(bp_layer_t *layer, size_t m, const bp_facts *facts);

/* entries per Morton cell at split_depth: hist[n_cells + 1] (host), last slot = coarse entries
 * (depth < split_depth).  n_cells must be 2^(dim * split_depth) and <= 4096. */
int bp_layer_cell_histogram(bp_layer_t *layer, uint32_t split_depth, uint64_t *hist, size_t n_cells);
/* announce the split depth in advance (0 = off): the next extend into the empty layer then counts this
 * histogram while it emits the entries and bp_layer_cell_histogram only sums the partial counts */
int bp_layer_set_cell_histogram_depth(bp_layer_t *layer, uint32_t split_depth);
/* stable bucketing of the layer's entries by owner: lut[n_cells] (host) maps a cell to its owner
 * 0..n_dest-1, coarse entries form bucket n_dest.  counts[n_dest + 1] (host).  The layer itself
 * is not modified. */
int bp_layer_partition_entries_device(bp_layer_t *layer, uint32_t split_depth, const uint8_t *lut, size_t n_cells,
                                      uint32_t n_dest, uint64_t *counts, const uint64_t **d_keys,
                                      const uint32_t **d_ids);
/* *can: bit 0 = this layer's ids can carry their box codes (ids < 2^27, distinct, id-ordered),
 * bit 1 = every box code of the layer is known */
int bp_layer_can_pack_box(const bp_layer_t *layer, int *can);
/* as above; flags: BP_PARTITION_PACK_BOX = the box code of every entry packed into bits 27..31 of its id;
 * BP_PARTITION_ENQUEUE_ONLY = when the bucket sizes follow from the cell histogram the layer has just reported
 * (bp_layer_cell_histogram at the same depth), return as soon as the pass is enqueued: counts are filled from the histogram,
 * and the caller makes whatever reads the buckets wait for the layer's stream (bp_layer_stream) */
enum { BP_PARTITION_PACK_BOX = 1, BP_PARTITION_ENQUEUE_ONLY = 2 };
int bp_layer_partition_entries_packed_device(bp_layer_t *layer, uint32_t split_depth, const uint8_t *lut,
                                             size_t n_cells, uint32_t n_dest, int flags, uint64_t *counts,
                                             const uint64_t **d_keys, const uint32_t **d_ids);
/* raw pair stream of the shard scan as packed u64 (id << 32 | id_), left in HBM */
int bp_layer_scan_raw_shard_device(bp_layer_t *layer, uint32_t coarse_depth, int emit_coarse,
                                   const uint64_t **d_packed, size_t *n_raw);
/* same stream in arbitrary order (single-pass scan); *unique = 1 if duplicates were suppressed at
 * the source for every entry of this layer (all box codes known) */
int bp_layer_scan_shard_pairs_device(bp_layer_t *layer, uint32_t coarse_depth, int emit_coarse,
                                     const uint64_t **d_packed, size_t *n_raw, int *unique);
/* the shard scan followed by this shard's own sort + dedup (par_scan restricted to the shard's Morton
 * range): sorted duplicate-free (id, id_) rows in HBM.  *unique = 1 if no other shard can hold any of these
 * pairs as well (duplicates suppressed at the source for every entry) */
int bp_layer_par_scan_shard_device(bp_layer_t *layer, uint32_t coarse_depth, int emit_coarse, const uint32_t **d_pairs,
                                   size_t *n_pairs, int *unique);
/* sort n DISTINCT packed pairs (no dedup needed); with the inclusive range of their first ids known the sort can stop
 * one digit early and finish 256-id groups in LDS (bp_layer_sort_packed_device: the same without the range) */
int bp_layer_sort_packed_range_device(bp_layer_t *layer, uint64_t *d_packed, size_t n, uint64_t varying_mask,
                                      uint32_t first_lo, uint32_t first_hi, const uint32_t **d_pairs, size_t *n_pairs);
/* stable bucketing of n packed pairs by the owner of the first id: bounds[i] = first id owned by
 * rank i + 1 (n_bounds <= 15); counts[n_bounds + 1] (host) */
int bp_layer_partition_pairs_device(bp_layer_t *layer, const uint64_t *d_packed, size_t n, const uint32_t *bounds,
                                    uint32_t n_bounds, uint64_t *counts, const uint64_t **d_out);
/* bp_layer_scan_shard_pairs_device + bp_layer_partition_pairs_device in one step (the partition pass
 * reads the scan's block-structured output directly, no dense intermediate): *d_out = *n_raw packed
 * pairs grouped by owner rank, counts[n_bounds + 1] (host) */
int bp_layer_scan_shard_partition_device(bp_layer_t *layer, uint32_t coarse_depth, int emit_coarse, const uint32_t *bounds,
                                         uint32_t n_bounds, uint64_t *counts, const uint64_t **d_out, size_t *n_raw,
                                         int *unique);
/* sort + dedup n packed pairs in device memory (d_packed is used as scratch and clobbered);
 * varying_mask: bits that may differ between the pairs (0 = all 64) */
int bp_layer_sort_unique_packed_device(bp_layer_t *layer, uint64_t *d_packed, size_t n, uint64_t varying_mask,
                                       const uint32_t **d_pairs, size_t *n_pairs);
/* sort only (the caller knows the n pairs are already distinct) */
int bp_layer_sort_packed_device(bp_layer_t *layer, uint64_t *d_packed, size_t n, uint64_t varying_mask,
                                const uint32_t **d_pairs, size_t *n_pairs);

/* ---- measurement ----------------------------------------------------------------------------- */
typedef struct bp_stats {
    uint64_t objects_valid;   /* objects appended by extend since the last clear       */
    uint64_t objects_invalid; /* objects skipped (outside system bounds)                */
    uint64_t entries;         /* M: index entries in the tree                           */
    uint64_t raw_pairs;       /* P_raw of the last scan                                 */
    uint64_t pairs;           /* P of the last scan                                     */
    uint64_t raw_candidates;  /* containment combos the last scan examined = the reference's raw pair count
                                 before dedup (src/layer.rs:566-570); 0 when the scan path does not count them */
    uint32_t tree_sort_passes; /* radix passes of the last tree sort (the LDS finish kernels are not counted)  */
    uint32_t pair_sort_passes; /* radix passes of the last pair sort                                            */
    uint32_t depth_mask;      /* bit d set <=> some entry has depth d                   */
    uint32_t flags;           /* bit0: key-only stable sort valid, bit1: ids distinct   */
    uint64_t host_syncs;      /* stream synchronisations since create (each is a host round trip) */
    /* plan events since create: what a scene that CHANGES from frame to frame costs beyond the steady state (every one of
     * them only costs time, never changes a result) */
    uint32_t extend_reruns;   /* extends launched on predicted sizes that had to run again on the plain path          */
    uint32_t tree_plan_flips; /* tree sorts whose plan (grouped finish on / off) differs from the previous sort's     */
    uint32_t pair_plan_flips; /* the same for the pair sort's grouped finish                                          */
    uint32_t sort_repairs;    /* deferred tree-sort checks that failed: full-width re-sort + scan run again           */
    uint32_t sorts_ahead;     /* scans whose pair sort + finish were enqueued behind them, sized by the previous frame
                                 (no host round trip between scan and sort)                                            */
    uint32_t sorts_ahead_missed; /* ... of which the frame outgrew those sizes: scan and sort ran again, blocking      */
    uint32_t sorts_in_cells;  /* ... of which took the cell route: two line-claiming passes into ranges laid out from the
                                 previous frame's group sizes (no stable pass, no search for the group bounds)            */
    uint32_t tree_sorts_in_cells; /* grouped tree sorts that took the cell route (the same idea for the (index, id) sort: a
                                     cell that outgrows its range is a sort_repair)                                        */
    uint32_t tree_sorts_compact;  /* ... of which moved 8-byte compact records (index bits that vary | id | box code in one
                                     word, written by extend) through the passes instead of (u64 index, u32 id) pairs      */
    uint32_t tree_expands;        /* compact records expanded to (index, id) arrays because something other than that sort
                                     asked for the tree first (iter / merge / a second extend / a repair ...)              */
} bp_stats;
int bp_layer_stats(const bp_layer_t *layer, bp_stats *out);

/* ---- batched region queries ----------------------------------------------------
 * Layer::test_box (src/layer.rs:294-313; test :252-279, test_impl :167-239, BoxTestGeometry
 * src/geom.rs:353-455) for n_queries boxes at once.  boxes: rows of 2*dim f32 [min..., max...], host or
 * device memory.  has_max_depth = 0 stands for max_depth = None.  Result: rows (query index, id),
 * sorted by query then id, duplicate-free -- row group q is exactly what test_box(system_bounds,
 * boxes[q], max_depth) returns.  Sorts the layer first, like the reference (src/layer.rs:260).
 * The rows are borrowed until the next call on the handle and share storage with the scan results. */
int bp_layer_test_box_batch(bp_layer_t *layer, const float *sys_min, const float *sys_max, const float *boxes,
                            size_t n_queries, int has_max_depth, uint32_t max_depth, const uint32_t **rows, size_t *n_rows);
int bp_layer_test_box_batch_device(bp_layer_t *layer, const float *sys_min, const float *sys_max, const float *boxes,
                                   size_t n_queries, int has_max_depth, uint32_t max_depth, const uint32_t **d_rows,
                                   size_t *n_rows);
/* Layer::test_ray (src/layer.rs:327-351, RayTestGeometry src/geom.rs:460-689) for n_queries rays at once.
 * rays: rows of 2*dim + 2 f32 [origin..., direction..., range_min, range_max] (the range may be -inf / +inf:
 * it is clamped to the system bounds).  Same result layout as bp_layer_test_box_batch. */
int bp_layer_test_ray_batch(bp_layer_t *layer, const float *sys_min, const float *sys_max, const float *rays,
                            size_t n_queries, int has_max_depth, uint32_t max_depth, const uint32_t **rows, size_t *n_rows);
int bp_layer_test_ray_batch_device(bp_layer_t *layer, const float *sys_min, const float *sys_max, const float *rays,
                                   size_t n_queries, int has_max_depth, uint32_t max_depth, const uint32_t **d_rows,
                                   size_t *n_rows);

/* Layer::pick_ray (src/layer.rs:418-446; pick :364-404) for n_queries rays at once, with a BUILT-IN distance
 * callback in place of the caller's closure: the entry distance of the ray into the AABB of the object
 * (bounds_by_id: n_rows rows of 2*dim f32 indexed by id; 0 when the origin is inside the box).  rays: rows of
 * 2*dim + 1 f32 [origin..., direction..., max_dist].  out_id[q] = the nearest object hit within max_dist
 * (first one in the reference's visiting order on ties) or 0xFFFFFFFF, out_dist[q] = its distance or +inf; the
 * reference's third result is origin + direction * dist.  Inputs and outputs may be host or device memory. */
int bp_layer_pick_ray_aabb_batch(bp_layer_t *layer, const float *sys_min, const float *sys_max, const float *rays,
                                 size_t n_queries, int has_max_depth, uint32_t max_depth, const float *bounds_by_id,
                                 size_t n_rows, float *out_dist, uint32_t *out_id);

/* Kernel classes timed with HIP events on the layer's own stream when profiling is enabled */
typedef enum bp_kernel_class {
    BP_K_ENCODE = 0,      /* encode_count + encode_emit                   */
    BP_K_TREE_HIST = 1,   /* digit histograms of the tree sort            */
    BP_K_TREE_SORT = 2,   /* onesweep radix passes over (key, id) entries */
    BP_K_SCAN_COUNT = 3,  /* overlap scan, counting pass                  */
    BP_K_SCAN_EMIT = 4,   /* overlap scan, emitting pass                  */
    BP_K_PAIR_HIST = 5,   /* digit histograms of the pair sort            */
    BP_K_PAIR_SORT = 6,   /* onesweep radix passes over packed pairs      */
    BP_K_DEDUP = 7,       /* adjacent-unique flag/compact                 */
    BP_K_MISC = 8,        /* block-sum scans, memsets, small reductions   */
    BP_K_TREE_FINISH = 9, /* LDS finish of the partial tree sort (group bounds + finish kernels) */
    BP_K_COUNT = 10
} bp_kernel_class;
int bp_layer_profile_enable(bp_layer_t *layer, int on);
/* restrict the event pairs to the classes whose bit (1 << class) is set (default: all).  Every
 * event pair costs a few microseconds of stream bubble, so a timed run selects only what it reports */
int bp_layer_profile_select(bp_layer_t *layer, uint32_t class_mask);
/* time only every `every`-th frame (frames are counted by bp_layer_clear; 1 = every frame, the default): a timed run
 * that reports one kernel's duration samples it instead of paying the bubbles in every frame */
int bp_layer_profile_sample(bp_layer_t *layer, uint32_t every);
int bp_layer_profile_reset(bp_layer_t *layer);
/* total device milliseconds and number of launches accumulated for one class */
int bp_layer_profile_get(bp_layer_t *layer, int kernel_class, double *total_ms, uint64_t *launches);

/* Kernel sequences are chosen from facts and from what earlier sorts on this handle learned about the scene
 * ("plan hints": whether prefix groups fit LDS, ...).  This forgets the hints, so the next frame runs the plan of
 * a handle's FIRST frame -- results are identical either way; bench.py uses it to time the cold plan. */
int bp_layer_reset_plan_hints(bp_layer_t *layer);

/* library version string, e.g. "broadphase_amd 0.1 gfx950" */
const char *bp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BP_LAYER_H */
