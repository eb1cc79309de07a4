/*
 * bp_sharded.h -- C ABI of the multi-GPU `Layer`: objects shard across the GPUs of one node, index entries move to
 * the owner of their Morton range, boundary keys (entries coarser than the split depth) are all-gathered, raw pairs
 * move to the owner of their first id.  One rank per GPU.
 *
 * The reference (zvxryb/broadphase-rs) is single-process; its only partitioning idea is the `min_depth` split of the
 * sorted array at cell boundaries (src/layer.rs:533-546), which works because an entry only ever interacts with the
 * entries whose cell contains its own.  The same property shards the path across GPUs.  This header
 * is what a Rust / C++ host of the example's frame loop (examples/main.rs:410-459) binds for the multi-GPU case; the
 * single-GPU boundary is bp_layer.h.  The Python driver broadphase_rs_amd/dist.py is the same pipeline over
 * torch.distributed and stays as the reference implementation of the exchange logic for the CPU (gloo) tests.
 *
 * Ranks and transport: a bp_group is either
 *   - RCCL: one process (or thread) per GPU, `world` ranks joined by a 128-byte unique id that rank 0 creates and the
 *     host distributes by its own means (bp_group_unique_id / bp_group_create_rccl).  librccl is loaded at run time
 *     (dlopen "librccl.so.1"): the library has no link-time dependency on it, and inside a PyTorch process it binds
 *     the RCCL that torch has already mapped.  Exchanges: grouped ncclSend / ncclRecv (all-to-all-v, cut into pieces
 *     of <= 256 MB per peer: the RCCL bundled with PyTorch-ROCm 2.10 halves single messages above 1 GiB) and
 *     ncclAllGather for the small count / fact vectors.
 *   - LOCAL: `world` ranks inside ONE process, each driven by its own host thread; exchanges are device-to-device
 *     copies between the ranks' buffers (same GPU or peers).  For single-process hosts and for tests on one GPU.
 * Every bp_sharded_* call is COLLECTIVE: all ranks of the group must make the same calls in the same order.
 * Status codes, error strings and the stream-ordering contract are those of bp_layer.h.
 */
#ifndef BP_SHARDED_H
#define BP_SHARDED_H

#include <stddef.h>
#include <stdint.h>

#include "bp_layer.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bp_group bp_group_t;     /* the ranks of one job and their transport */
typedef struct bp_sharded bp_sharded_t; /* one rank's share of a multi-GPU Layer<Index, u32> */

#define BP_UNIQUE_ID_BYTES 128
/* ranks per group: the owner tables of the pair exchange (first-id ranges) hold 15 bounds; one node has 8 GPUs */
#define BP_SHARDED_MAX_RANKS 16

/* rank 0: a fresh id for a new RCCL group (ncclGetUniqueId); the host hands the 128 bytes to every rank */
int bp_group_unique_id(void *out_128_bytes);
/* join an RCCL group as `rank` of `world` on HIP device `device` (blocks until all ranks have joined) */
int bp_group_create_rccl(bp_group_t **out, int rank, int world, const void *unique_id_128_bytes, int device);
/* a group of `world` ranks inside this process; every rank passes the SAME group to bp_sharded_create */
int bp_group_create_local(bp_group_t **out, int world);
int bp_group_destroy(bp_group_t *group);
const char *bp_group_last_error(const bp_group_t *group);

typedef enum bp_pair_order {
    /* rank r returns the r-th first-id range of the reference's sorted list: the concatenation over ranks, in rank
     * order, IS Layer::par_scan's result, bit for bit */
    BP_PAIRS_GLOBAL = 0,
    /* every rank returns, sorted and duplicate-free, the pairs found in its own Morton range: the ranks' lists are
     * disjoint and their union is the reference's set, but no pair crosses the wire.  Only while every rank can
     * suppress duplicates at the source; otherwise the call takes the GLOBAL route (bp_sharded_info says which) */
    BP_PAIRS_LOCAL = 1
} bp_pair_order;

/* split_depth = 0: the default (3 in 3-D: 512 cells, 4 in 2-D: 256 cells are the units of ownership).
 * id_count: ids are known to lie in [0, id_count) (0: unknown -- the id range is taken from the ranks' own facts).  Every
 * rank must pass the same value; bp_sharded_par_scan returns BP_ERR_INVALID_ARG on every rank when the values differ or
 * some rank was handed an id >= id_count (checked on the gathered facts, before anything is exchanged).
 *
 * Failure model of the collective calls.  A rank whose own layer calls fail BEFORE the first exchange still takes part in
 * it: its status word travels with the facts and every rank returns non-zero together.  A rank that fails later leaves the
 * sequence: LOCAL groups break their barrier (the waiting ranks return BP_ERR_COMM); an RCCL rank aborts its communicator
 * (ncclCommAbort) and the group refuses further use; its peers' waits on the exchange stream are bounded
 * (BP_SHARDED_TIMEOUT_MS, default 60000) and poll the communicator's asynchronous error state, so they return
 * BP_ERR_COMM instead of blocking.  After such a return the group is dead: destroy it and build a new one.  (Work the
 * layers chain behind an exchange by stream events is bounded by the same exchange.) */
int bp_sharded_create(bp_sharded_t **out, bp_group_t *group, int rank, int index_kind, uint32_t min_depth,
                      uint32_t split_depth, int device, uint64_t id_count);
int bp_sharded_destroy(bp_sharded_t *sh);
const char *bp_sharded_last_error(const bp_sharded_t *sh);

/* Layer::clear / Layer::extend on this rank's own objects (host or device pointers, as bp_layer_extend) */
int bp_sharded_clear(bp_sharded_t *sh);
int bp_sharded_extend(bp_sharded_t *sh, const float *sys_min, const float *sys_max, const float *bounds,
                      const uint32_t *ids, size_t n);
/* Layer::par_scan over the union of all ranks' objects.  *d_pairs: this rank's share of the pair list, interleaved
 * (id, id_) u32 in HBM, borrowed until the next call on the handle */
int bp_sharded_par_scan(bp_sharded_t *sh, int pair_order, const uint32_t **d_pairs, size_t *n_pairs);
/* the same, copied to host memory the caller provides (n_pairs from a previous bp_sharded_par_scan) */
int bp_sharded_copy_pairs(bp_sharded_t *sh, uint32_t *host_dst, size_t n_pairs);

typedef struct bp_sharded_info {
    uint64_t entries_local;  /* entries this rank encoded                                        */
    uint64_t entries_shard;  /* entries of this rank's Morton range + boundary keys               */
    uint64_t coarse;         /* boundary keys (replicated on every rank)                          */
    uint64_t raw_pairs;      /* pairs this rank's scan emitted                                    */
    uint64_t raw_pairs_recv; /* pairs received for this rank's first-id range (0 on the LOCAL route) */
    uint64_t pairs;          /* pairs returned                                                    */
    uint32_t id_lo, id_hi;   /* first-id range of this rank (GLOBAL route)                        */
    uint32_t box_codes;      /* the box codes travelled in the ids (duplicates suppressed at the source) */
    uint32_t pair_order;     /* the route taken: BP_PAIRS_GLOBAL / BP_PAIRS_LOCAL                 */
    /* host wall time of the stages of the last bp_sharded_par_scan on this rank, milliseconds (every stage ends with the
     * rank's device work complete): 0 cell histogram + all-gather of histograms and facts, 1 partition of the entries by
     * owner, 2 exchange of entries and boundary keys, 3 commit into the shard layer, 4 shard sort + scan (+ bucketing of
     * the pairs by owner), 5 exchange of pairs, 6 sort (+ dedup) of this rank's pairs, 7 the whole call */
    float stage_ms[8];
    /* host round trips of the last bp_sharded_par_scan on this rank (each costs a launch -> completion -> host latency that
     * no amount of GPUs shortens): stream synchronisations of the rank's two layers, and exchanges / collectives waited for */
    uint32_t round_trips_layers;
    uint32_t round_trips_exchanges;
} bp_sharded_info;
int bp_sharded_last_info(const bp_sharded_t *sh, bp_sharded_info *out);
/* the rank's shard layer (sort / scan target), e.g. for bp_layer_stats; owned by the handle */
int bp_sharded_shard_layer(bp_sharded_t *sh, bp_layer_t **layer);

#ifdef __cplusplus
}
#endif
#endif /* BP_SHARDED_H */
