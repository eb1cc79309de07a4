// bp_kernels.h -- host-side launch interface of the gfx950 kernels (internal; the public
// boundary is include/bp_layer.h).  Every launcher enqueues on `stream` and returns a hipError_t.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace bp {

// ---- index kinds (reference: src/index.rs:293-295) -------------------------------------------
struct KindInfo {
    uint32_t dim, bits, depth_bits, axis_bits;
};
inline KindInfo kind_info(int kind) {
    switch (kind) {
        case 0: return {2, 32, 4, 14};
        case 1: return {2, 64, 5, 29};
        default: return {3, 64, 5, 19};
    }
}

// ---- device scalars shared by the kernels of one layer ------------------------------------------
// Zeroed / initialised by the host before each phase, read back with one small D2H copy.
struct DevScalars {
    unsigned long long total;      // result of the last block-sum scan
    unsigned long long key_or;     // OR / AND of every index value appended (digit planning)
    unsigned long long key_and;
    unsigned int id_or, id_and;    // OR / AND of the ids appended
    unsigned int depth_mask;       // bit d set <=> an entry of depth d exists
    unsigned int flags;            // EXT_* bits below
    unsigned int err;              // KERR_* bits below
    unsigned int n_invalid;        // objects outside the system bounds
    unsigned int id_max;           // largest id handed to the last extend (valid or not)
    unsigned int id_min;           // smallest id handed to the last extend
    unsigned int big_count;        // expanding scan: slices on the long-range work list
    unsigned int grp_max;          // grouped finishes: the largest group ...
    unsigned int grp_over;         // ... and the records in groups above the finish kernel's capacity (group_stats_words)
};
// the two words a bounds kernel reduces the group sizes into (zero after a reset of the scalars): [0] = grp_max, [1] = grp_over
inline unsigned int* group_stats_words(DevScalars* d_sc) { return &d_sc->grp_max; }
enum : unsigned {
    EXT_IDS_DECREASE = 1u,   // some id is smaller than its predecessor in append order
    EXT_IDS_REPEAT = 2u,     // some id is <= its predecessor (ids not provably distinct)
    EXT_BOX_UNKNOWN = 4u,    // some object spans more than 2 cells on an axis (min_depth forced it)
};
enum : unsigned {
    KERR_LOOKBACK_TIMEOUT = 1u,  // a decoupled look-back spin hit its bound
    KERR_COUNT_OVERFLOW = 2u,    // an object expands to more cells than a launch can index
    KERR_SEGMENT_TOO_LONG = 4u,  // segment_finish: one first id owns more pairs than its window holds
    KERR_BIGLIST_OVERFLOW = 8u,  // expanding scan: more long-range slices than the work list holds
    KERR_BAD_ID = 16u,           // pick_ray: an id of the tree has no row in the bounds table
    KERR_ENCODE_OVERFLOW = 32u,  // encode_emit launched on a PREDICTED capacity that was too small (nothing was written)
};

// Where a kernel leaves results for a host that is waiting for them: pinned, host-coherent memory (hipHostMalloc; the pointers are
// its device addresses).  The kernel writes the data to dst, then -- behind a system-scope release -- seq to *flag; seq is a number
// the host has not used for this flag before.
struct Published {
    void* dst;
    unsigned int* flag;
    unsigned int seq;
};

// per-workgroup partial results of encode_count, folded into DevScalars by scan_block_sums
struct BlockAux {
    unsigned long long key_or, key_and;
    unsigned int depth_mask, flags, err, n_invalid, id_or, id_and, id_max, id_min;
};

// Per-entry description of its object's cell set, when that set is a box of at most 2 cells per
// axis (always, unless min_depth forces more): per axis t = 0 (one cell), 1 (two cells, this is the
// lower), 2 (two cells, this is the upper).  In memory: one byte, 2 bits per axis (t_x | t_y << 2 |
// t_z << 4), BOX_UNKNOWN otherwise.  Through the tree sort it rides in the 5 spare high bits of
// the id payload (ids < 2^27) as the ternary number t_x + 3 t_y + 9 t_z (31 = unknown).
enum : unsigned { BOX_UNKNOWN = 0xFFu, BOX_PACKED_UNKNOWN = 31u, BOX_ID_BITS = 27u, BOX_ID_MASK = (1u << 27) - 1u };

struct SysBounds {
    float mn[3];
    float mx[3];
};

// ---- compact tree records (round 4) ---------------------------------------------------------------------------------------
// Where an entry's index value, its id and its 5-bit box code fit ONE 64-bit word, the per-frame path moves the tree through
// encode -> radix passes -> finish as such words instead of (u64 index, u32 id) pairs: 8 bytes per record and pass instead of 12.
//   word = (prefix << tbits | tail) << low | id << 5 | code                                     (low = idbits + 5)
//   prefix = index >> pshift: the Morton bits of the levels 1 .. dmin (dmin = the shallowest depth present; pshift = DEPTH_BITS +
//            DIM * (AXIS_BITS - dmin)) -- every entry has all of them;
//   tail   = the levels dmin + 1 .. dmax as a number of T = dmax - dmin digits in base 2^DIM + 1: digit k = 1 + the entry's
//            octant (quadrant) at level dmin + k while that level is at most the entry's depth, 0 ("the cell ends here") below
//            it.  This is the pre-order of the tree written as a number: a cell comes before the cells it contains, which come
//            in Morton order -- exactly the order of the index values (Morton bits above the depth field, src/index.rs:72-78),
//            with no separate depth field and no bits for levels an entry does not have (at 10 M objects, depths 7 .. 11: 34
//            bits where Morton bits + depth take 36 -- the difference between fitting 64 bits and not).
// The id sits above the code, so the unsigned order of the words is the (index, id) order of the entries (src/layer.rs:149).
struct TreeCompact {
    int pshift, tbits, T, low, idbits;
    uint32_t dmin, dmax;
    uint32_t id_base;  // the id field holds id - id_base (ids of any magnitude whose RANGE fits idbits)
};
// the format for depths dmin .. dmax and ids below 2^idbits; false: does not fit 64 bits
inline bool tree_compact_format(int kind, uint32_t dmin, uint32_t dmax, int idbits, TreeCompact* f) {
    const KindInfo ki = kind_info(kind);
    if (dmax > ki.axis_bits || dmin > dmax || idbits < 1 || idbits > 27) return false;
    const int T = (int)(dmax - dmin);
    const unsigned long long base = (1ull << ki.dim) + 1ull;
    unsigned long long span = 1;  // base^T
    for (int k = 0; k < T; ++k) {
        span *= base;
        if (span > 0xFFFFFFFFull) return false;  // (the kernels handle the tail as a 32-bit number: at most 10 levels in 3-D, 13 in 2-D)
    }
    int tbits = 0;
    while ((1ull << tbits) < span) ++tbits;
    const int total = (int)(ki.dim * dmin) + tbits + idbits + 5;
    if (total > 64) return false;
    f->pshift = (int)(ki.depth_bits + ki.dim * (ki.axis_bits - dmin));
    f->tbits = tbits; f->T = T; f->low = idbits + 5; f->idbits = idbits; f->dmin = dmin; f->dmax = dmax; f->id_base = 0;
    return true;
}
// composites (n of them, in place over `words`) -> index values in words[], ids in ids_out[], box bytes (BOX_*) in box_out[]
hipError_t launch_tree_expand(int kind, uint64_t* words, uint32_t* ids_out, uint8_t* box_out, size_t n, const TreeCompact& fmt, hipStream_t stream);

// ---- encode (K1..K3): reference src/layer.rs:94-121, src/geom.rs:121-128,148-163,189-304 ------
constexpr int ENC_THREADS = 256;
#ifndef ENC_ROUNDS_DEF
#define ENC_ROUNDS_DEF 4
#endif
constexpr int ENC_ROUNDS = ENC_ROUNDS_DEF;     // blocks of ENC_THREADS objects per workgroup of encode_count / encode_emit
constexpr int ENC_STAGE = 2048;   // entries of one round staged in LDS by encode_emit (8 per object)
constexpr int ENC_CELLS_MAX = 512; // cells encode_emit can histogram (8^3 in 3-D, 4^4 in 2-D)
constexpr int HIST_COPIES = 32;   // copies of the digit histograms encode_emit spreads its atomics over
// workgroups (= block sums / BlockAux partials) for n objects
inline size_t enc_blocks(size_t n) { return (n + (size_t)ENC_THREADS * ENC_ROUNDS - 1) / ((size_t)ENC_THREADS * ENC_ROUNDS); }

hipError_t launch_encode_count(int kind, SysBounds sb, const float* bounds, const uint32_t* ids, size_t n,
                               uint32_t min_depth, int have_prev_id, uint32_t prev_id, void* recs,
                               unsigned long long* block_sums, BlockAux* aux, DevScalars* sc, hipStream_t stream,
                               unsigned long long* zero_a = nullptr, size_t n_a = 0, unsigned long long* zero_b = nullptr, size_t n_b = 0);
// zero_a / zero_b (optional): words the count pass also zero-fills, spread over its workgroups (the histograms the emit pass adds into)
// recs: ENC_REC_BYTES per object -- what the emit pass needs of it (first cell per axis, depth, cells per axis), so that
// launch_encode_emit neither reads the bounds again nor repeats the quantisation
constexpr size_t ENC_REC_BYTES = 16;
hipError_t launch_encode_emit(int kind, SysBounds sb, const float* bounds, const uint32_t* ids, size_t n,
                              uint32_t min_depth, const void* recs, const unsigned long long* block_offs,
                              uint64_t* keys_out, uint32_t* ids_out, uint8_t* box_out, hipStream_t stream,
                              const struct RadixPlan* hist_plan = nullptr, unsigned long long* hist = nullptr, int pack_box = 0,
                              unsigned long long* cell_hist = nullptr, uint32_t cell_depth = 0, DevScalars* sc_guard = nullptr,
                              size_t capacity = 0, const TreeCompact* compact = nullptr);
// compact (no cell_hist; pack_box is ignored): keys_out receives one composite word per entry (TreeCompact) and nothing else is
// written; hist_plan's shifts then refer to the composite
// sc_guard / capacity: launched straight behind launch_scan_block_sums, without the host having read the entry total: if
// sc_guard->total exceeds `capacity` entries the kernel writes nothing and sets KERR_ENCODE_OVERFLOW in sc_guard->err
// cell_hist / cell_depth: also count the entries per cell at cell_depth (multi-GPU splitter selection) into
// HIST_COPIES copies of 2^(dim*cell_depth) + 1 counters (zeroed by the caller; at most ENC_CELLS_MAX cells)
// pack_box: write id | box_to_packed(code) << BOX_ID_BITS into ids_out (box_out may then be nullptr)
// hist_plan / hist: also count the key digits of every emitted entry for the passes of hist_plan
// (<= SX_HIST_MAX, key digits only) into hist[HIST_COPIES][SX_HIST_MAX][256] (zeroed by the caller)

// exclusive scan of per-block sums (in place), grand total -> sc->total; if aux != nullptr its n
// partials are folded into *sc as well (single workgroup, no atomics)
// init (with aux): *sc starts from *init before the partials are folded in (no separate reset); zero_a / zero_b: 64-bit arrays
// zero-filled by the same launch (the histograms the following emit kernel adds into)
// pub: the final *sc also goes to pinned host memory, followed by the launch's number (struct Published)
hipError_t launch_scan_block_sums(unsigned long long* sums, size_t n, const BlockAux* aux, DevScalars* sc,
                                  hipStream_t stream, const DevScalars* init = nullptr, unsigned long long* zero_a = nullptr,
                                  size_t n_a = 0, unsigned long long* zero_b = nullptr, size_t n_b = 0, const struct Published* pub = nullptr);
// `bytes` (a multiple of 4) of device memory -> pub.dst, then pub.seq -> *pub.flag: a host that spins on the flag has what a D2H
// copy + hipStreamSynchronize would give it (everything enqueued before this launch has completed), a few microseconds sooner
// (src2: a second block of bytes2 bytes, to pub.dst + dst_off2)
hipError_t launch_publish(const void* src, size_t bytes, const struct Published& pub, hipStream_t stream, const void* src2 = nullptr,
                          size_t bytes2 = 0, size_t dst_off2 = 0);

// OR/AND/depth-mask reduction over an existing tree (set_tree / append_entries)
hipError_t launch_tree_reduce(int kind, const uint64_t* keys, const uint32_t* ids, size_t m, DevScalars* sc,
                              hipStream_t stream);

// ---- radix sort (K4, K7): stands in for par_sort_unstable, src/layer.rs:149,516 ---------------
constexpr int RS_BINS = 256;
constexpr int RS_MAX_PASSES = 12;
#ifndef SX_HIST_MAX_DEF
#define SX_HIST_MAX_DEF 4
#endif
constexpr int SX_HIST_MAX = SX_HIST_MAX_DEF;      // digit histograms a producer kernel (encode, scan) can count for the sort that follows

struct RadixPlan {
    int n_passes;
    int shift[RS_MAX_PASSES];
    int from_val[RS_MAX_PASSES];  // 1: digit taken from the u32 payload (id) instead of the u64 key
};
// plan passes over the varying bits only; id passes (if any) come first (LSD order)
RadixPlan make_radix_plan(unsigned long long key_varying, int use_ids, unsigned id_varying);

// workspace layout for one sort: [hist: passes*256 u64][bin_base: passes*256 u64]
// [tile counters: passes u32 (padded)][status: passes * tiles * 256 u64]
size_t radix_workspace_bytes(size_t n, int n_passes);

// Sorts n (key[,val]) records.  keys/vals are ping-pong buffers; *result_buf tells which holds
// the result (0 or 1).  vals0 == nullptr: keys only.
// box_in / box_out (optional, key+u32-payload sorts only): per-record byte packed into the payload's
// bits 27..31 by the first pass and unpacked by the last one into box_out (may alias box_in).
// epoch_io (optional): the caller's launch counter for THIS workspace -- the passes tag their look-back status words with
// it instead of zero-filling the workspace (the caller zero-fills once, when the workspace is allocated, and starts the
// counter at 0).  sc_init (optional): *sc is set to this by the sort's first kernel (saves the caller's own reset).
hipError_t radix_sort(uint64_t* keys0, uint64_t* keys1, uint32_t* vals0, uint32_t* vals1, size_t n,
                      const RadixPlan& plan, void* workspace, size_t workspace_bytes, DevScalars* sc,
                      int* result_buf, hipStream_t stream, void (*mark)(void*, int), void* mark_ctx,
                      int class_hist, int class_sort, const uint8_t* box_in = nullptr, uint8_t* box_out = nullptr,
                      const uint32_t* blockfill = nullptr, size_t extent = 0, const unsigned long long* ready_hist = nullptr,
                      int ready_copies = 1, int unstable_first = 0, uint32_t* epoch_io = nullptr, const DevScalars* sc_init = nullptr,
                      const uint32_t* dev_sizes = nullptr, const struct ScanFinalize* finalize = nullptr,
                      const struct CellRoute* cells = nullptr);
// dev_sizes (with ready_hist only: nothing on the host side reads the records): the sort is enqueued behind a producer
// nobody has waited for -- n and extent are UPPER BOUNDS (they size grids and the workspace), the kernels take the sizes
// from dev_sizes[SIZES_N] / dev_sizes[SIZES_EXTENT] (launch_scan_finalize writes them; both 0 = do nothing)
enum { SIZES_EXTENT = 0, SIZES_N = 1, SIZES_MISS = 2, SIZES_CELL_MISS = 3, SIZES_BIG = 4 /* slices the scan put on its work list */, SIZES_WORDS = 8 };
// finalize (with dev_sizes): the sort's first kernel also does launch_scan_finalize's job (one launch less): it derives the
// sizes from the scan's shard words and writes them to `sizes` (== dev_sizes) before any pass reads them
struct ScanFinalize {
    const uint64_t* shards; size_t out_blocks, n_bound, extent_bound; const DevScalars* sc; uint32_t* sizes;
};
// cells (two key passes, key-only records, unstable_first, counts ready): NO pass keeps a stable order.  The claiming pass
// goes by the HIGH digit (plan.shift[1]) and a second claiming pass spreads every high bin over its 256 CELLS (high digit,
// low digit), whose ranges were laid out beforehand with room to spare (launch_cell_layout: from the cells' sizes in the
// PREVIOUS sort of the same kind).  The result is in keys0, gapped: cell c holds 16 * line_cur[c] records from
// cell_base[c] upwards and frag_cur[c] records from cell_base[c + 1] downwards (launch_cell_count turns the cursors into
// the exact group bounds a finishing kernel writes its dense output by).  A cell that overflowed sets *miss: the result
// is incomplete and the caller sorts again the plain way.
// would a sort of `extent` slots take radix_claim_kernel for its first pass (size threshold and switches, read per call)?
bool radix_claim_applies(size_t extent);
constexpr uint32_t CELL_COUNT = 65536;
struct CellRoute {
    const uint32_t* cell_base;  // [CELL_COUNT + 1], multiples of 16
    uint32_t* line_cur;         // [CELL_COUNT] zero at launch
    uint32_t* frag_cur;         // [CELL_COUNT] zero at launch
    uint32_t* miss;             // one word, zero at launch
    uint4* src;                 // [CELL_COUNT] scratch: launch_cell_count leaves every group's two input pieces here for the finish kernel
    uint64_t* out_keys;         // sorts with a payload: where the gapped result goes (cell_layout_capacity(n) slots each, line-aligned);
    uint32_t* out_vals;         //   key-only sorts leave it in keys0 (both null)
    unsigned long long* status; // [64] scratch of the two cell walks (zero once, when allocated)
    uint32_t tag;               // an even number the caller never repeats for this `status` (launch_cell_layout tags its walk with it, launch_cell_count with tag + 1)
    // the layout folded into the sort's first kernel (radix_sort with ready_hist; one launch less): launch_cell_layout's arguments,
    // layout_prev_bounds == nullptr when the caller has run launch_cell_layout itself
    const uint32_t* layout_prev_bounds; uint32_t layout_g0, layout_groups; size_t layout_limit;
};
// cell_base from the group bounds of the previous sort (prev_bounds[g] = start of group prev_g0 + g, g = 0 .. prev_groups; cell
// index = group index = first id >> 8): capacity = count + 25 % + 64, rounded to whole lines (an empty cell: one line);
// also zeroes the cursors and the miss word.  total_cap (host, optional): upper bound of cell_base[CELL_COUNT] for n records.
// limit: slots of the buffer the cells live in -- a layout that would not fit (prev_bounds not what the caller thought) gives
// every cell the range [0, 0) and sets *miss
hipError_t launch_cell_layout(const uint32_t* prev_bounds, uint32_t prev_g0, uint32_t prev_groups, uint32_t* cell_base, uint32_t* line_cur,
                              uint32_t* frag_cur, uint32_t* miss, size_t limit, unsigned long long* status, uint32_t tag, hipStream_t stream);
inline size_t cell_layout_capacity(size_t n) { return n + n / 4 + (size_t)CELL_COUNT * 80; }
// after the sort: bounds[g] (g = 0 .. n_groups) = exact prefix of the sizes of cells g0 .. g0 + n_groups - 1, the group statistics
// launch_group_finish wants in stats[0..1], and *miss |= 1 when a cell holds more than its range or the total differs from
// dev_sizes[SIZES_N]
// err_word / err_bit (optional): *err_word |= err_bit whenever *miss is or gets set (a caller that reads scalars, not the miss word);
// want_total (optional, host-known record count): a different total is a miss too
hipError_t launch_cell_count(const CellRoute& cells, uint32_t g0, uint32_t n_groups, uint32_t* bounds, uint32_t cap, uint32_t* stats,
                             const uint32_t* dev_sizes, hipStream_t stream, uint32_t* err_word = nullptr, uint32_t err_bit = 0,
                             uint32_t want_total = 0);
// unstable_first: the first pass may order records of equal digit arbitrarily (cheaper ranking).  Only for sorts whose
// result does not depend on the input order of equal keys: a finishing kernel orders what the passes leave tied, or all
// varying bits of distinct records are sorted.  NOT for the key-only stable tree sort that keeps equal keys in id order.
// blockfill / extent (key-only sorts): keys0 is the block-structured output of the expanding scan --
// `extent` slots in blocks of SX_BLOCK, of which only the first blockfill[b] of block b hold a
// record (n records in all).  The histogram and the first pass read around the holes; the result is
// dense.  The workspace must be sized for `extent`.
// ready_hist: n_passes x 256 digit counts of exactly this plan, already on the device (the expanding
// scan counts them while it emits): the histogram kernel is skipped.  ready_copies: the counts are spread
// over that many arrays of SX_HIST_MAX x 256 u64 (summed here).

// ---- multi-GPU bucketing: one stable one-sweep pass with a custom digit (<= 256 buckets) --------
// entries -> owner rank by Morton cell at split_depth (lut on the device), coarse entries
// (depth < split_depth) -> bucket `coarse_bucket`; *d_hist (device, 256 u64) = bucket sizes
hipError_t partition_entries(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* ids_in, uint32_t* ids_out, size_t m,
                             const unsigned char* d_lut, int cell_shift, uint32_t depth_mask, uint32_t split_depth,
                             uint32_t coarse_bucket, void* workspace, size_t workspace_bytes, DevScalars* sc,
                             const unsigned long long** d_hist, hipStream_t stream, const uint8_t* box_in = nullptr,
                             const unsigned long long* ready_hist = nullptr);
// the same pass when the bucket sizes are known on the host (the caller derived them from the cell histogram): the owner table, the
// sizes and the scalars' reset travel as kernel arguments of ONE setup launch (no H2D copies, no memset, no D2D copy); n_cells <=
// PART_SETUP_MAX_CELLS; d_lut: device scratch of n_cells bytes the setup kernel fills
constexpr uint32_t PART_SETUP_MAX_CELLS = 1024;
hipError_t partition_entries_known(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* ids_in, uint32_t* ids_out, size_t m,
                                   const unsigned char* host_lut, uint32_t n_cells, unsigned char* d_lut, const unsigned long long* host_hist,
                                   int cell_shift, uint32_t depth_mask, uint32_t split_depth, uint32_t coarse_bucket, void* workspace,
                                   size_t workspace_bytes, DevScalars* sc, const DevScalars& sc_init, const unsigned long long** d_hist,
                                   hipStream_t stream, const uint8_t* box_in = nullptr);
// device-to-device copy: a kernel for anything above 1 MB (the runtime's copy runs at a third of the memory's rate), hipMemcpyAsync below
hipError_t launch_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t stream);
// ids[i] = packed & BOX_ID_MASK, box[i] = packed >> BOX_ID_BITS (in place on ids)
hipError_t launch_split_box(uint32_t* ids, uint8_t* box, size_t m, hipStream_t stream);
// the inverse: ids[i] |= box_to_packed(box[i]) << BOX_ID_BITS (in place)
hipError_t launch_pack_box(uint32_t* ids, const uint8_t* box, size_t m, hipStream_t stream);
// packed pairs -> owner rank of the first id; bounds[i] = first id owned by rank i+1
// blockfill != nullptr: `in` is the block-structured output of the expanding scan (n = its extent, n_out records exist)
hipError_t partition_pairs(const uint64_t* in, uint64_t* out, size_t n, const uint32_t* bounds, int n_bounds, void* workspace,
                           size_t workspace_bytes, DevScalars* sc, const unsigned long long** d_hist, hipStream_t stream,
                           const uint32_t* blockfill = nullptr, size_t n_out = 0, const unsigned long long* ready_hist = nullptr);
// entries per cell at split_depth; slot n_cells counts the coarse entries
// out[k] = sum over c < n_copies of copies[c * n + k]  (the spread-out histograms of encode_emit)
hipError_t launch_sum_copies(const unsigned long long* copies, int n_copies, uint32_t n, unsigned long long* out, hipStream_t stream);
hipError_t launch_cell_hist(const uint64_t* keys, size_t m, int cell_shift, uint32_t depth_mask, uint32_t split_depth,
                            uint32_t n_cells, unsigned long long* d_hist, hipStream_t stream);

// ---- overlap scan (K5, K6): reference src/layer.rs:550-573 --------------------------------------
constexpr int SCAN_THREADS = 256;
inline size_t scan_blocks(size_t m) { return (m + SCAN_THREADS - 1) / SCAN_THREADS; }

enum : int { FILT_AABB = 0, FILT_SPHERE = 1 };  // shape of the built-in narrow-phase rows
struct ScanParams {
    const unsigned long long* keys;
    const uint32_t* ids;
    size_t m;
    uint32_t depth_mask;    // present depths
    uint32_t coarse_depth;  // shard mode: entries with depth < coarse_depth are "coarse"
    int emit_coarse;        // pairs whose deeper entry is coarse: 0 never emitted here, 1 always, 2 when the deeper entry's id lies in
    uint32_t coarse_id_lo, coarse_id_hi;  // ... [coarse_id_lo, coarse_id_hi] (the rank's first-id range)
    int general;            // ids may repeat: apply the reference's skip rule exactly
    uint8_t* skip;          // [m] general mode only
    const uint8_t* box;     // [m] per-entry cell-box code (BOX_*), or nullptr: emit every raw pair
                            // non-null: a pair is emitted once, from the minimal-key cell of the
                            // deeper object that lies in any cell of the other (exact dedup at source)
    const float* filt_bounds;  // built-in narrow-phase predicate (fused scan only): rows of 2*dim f32 indexed BY ID,
    uint32_t filt_rows;        // keep (id, id_) iff the inclusive boxes overlap (src/geom.rs:112-119); nullptr: keep all
    int filt_shape;            // FILT_AABB, or FILT_SPHERE: rows of dim+1 f32 (centre, radius), examples/main.rs:466-474
};
hipError_t launch_scan_skipflags(int kind, ScanParams p, hipStream_t stream);
hipError_t launch_scan_count(int kind, ScanParams p, uint32_t* counts, unsigned long long* block_sums,
                             hipStream_t stream);
hipError_t launch_scan_emit(int kind, ScanParams p, const uint32_t* counts, const unsigned long long* block_offs,
                            uint64_t* pairs_out, hipStream_t stream);

// keep the packed pairs whose boxes (rows of `bounds` indexed by id) overlap; arbitrary output order;
// sc->total (zeroed by the host) = number kept
hipError_t launch_filter_pairs_aabb(int dim, const uint64_t* in, size_t n, const float* bounds, int shape, uint64_t* out, DevScalars* sc,
                                    hipStream_t stream);

// single-pass scan for the paths that sort the raw stream afterwards: pairs in arbitrary order,
// sc->total (zeroed by the host) = number of pairs produced; if it exceeds out_cap nothing past
// the capacity was written and the host re-runs with a larger buffer
hipError_t launch_scan_fused(int kind, ScanParams p, uint64_t* pairs_out, size_t out_cap, DevScalars* sc, hipStream_t stream);

// Expanding scan (default for distinct ids): the pair set of launch_scan_fused, candidates enumerated
// per ANCESTOR entry as the contiguous range of entries its cell contains.  idctx[m]: per-entry
// id << 32 | suppression context, written by launch_entry_ctx (box == nullptr: nothing is suppressed;
// entries of depth < inactive_below never act as the deeper side).
// Output is BLOCK-STRUCTURED: out_blocks blocks of SX_BLOCK slots; blockfill[g] (zeroed by the
// caller) = records present at the start of block g.  shards (zeroed by the caller): SX_SHARDS block
// cursors followed by SX_SHARDS pair counts; shard s hands out blocks s, s + SX_SHARDS, ...  The
// caller derives  pairs = sum(counts),  extent = max(cursor) * SX_SHARDS * SX_BLOCK  and re-runs with
// more blocks when extent exceeds the capacity.  big_list: scratch for big_cap work items of TWO u64 each;
// KERR_BIGLIST_OVERFLOW in sc->err means the result is incomplete (use scan_fused).
#ifndef SX_BLOCK_DEF
#define SX_BLOCK_DEF 128
#endif
constexpr uint32_t SX_BLOCK = SX_BLOCK_DEF;
constexpr int SX_SHARDS = 64;
#ifndef SX_RESERVE_PAIRS
#define SX_RESERVE_PAIRS 1024
#endif
constexpr uint32_t SX_RESERVE = SX_RESERVE_PAIRS / SX_BLOCK;  // output blocks a wave reserves per atomic
constexpr int SXB_BLOCKS = 2048;    // grid of the work-list kernel
bool scan_can_count(const RadixPlan& plan);  // launch_scan_expand counts the digits of this plan (hist_plan / hist) iff true
hipError_t launch_entry_ctx(int kind, const uint64_t* keys, const uint32_t* ids, const uint8_t* box, size_t m,
                            uint32_t inactive_below, uint64_t* idctx, hipStream_t stream, uint32_t active_lo = 1, uint32_t active_hi = 0);
// active_lo .. active_hi: ids whose entries stay active below inactive_below (lo > hi: none)
// scalars := init, blockfill[0..n_blocks) := 0, shards[0..n_shard_words) := 0, hist[0..n_hist_words) := 0 (hist may be null)
hipError_t launch_scan_prepare(DevScalars* sc, const DevScalars& init, uint32_t* blockfill, size_t n_blocks, uint64_t* shards,
                               uint32_t n_shard_words, unsigned long long* hist, uint32_t n_hist_words, hipStream_t stream);
hipError_t launch_scan_expand(int kind, const uint64_t* keys, const uint64_t* idctx, size_t m, const float* filt_bounds, int filt_shape,
                              uint64_t* pairs_out, size_t out_blocks, uint32_t* blockfill, uint64_t* shards,
                              uint64_t* big_list, uint32_t big_cap, DevScalars* sc, hipStream_t stream,
                              const RadixPlan* hist_plan = nullptr, unsigned long long* hist = nullptr,
                              const uint32_t* owner_bounds = nullptr, int owner_n = -1, int gen = 0, int mixed_scale = 0,
                              const uint16_t* dcnt = nullptr, uint32_t hist_rows = 0xFFFFFFFFu);
// hist_rows: bit q = count row q of hist_plan (a sort that reads one of the digits only: the others stay zero)
constexpr size_t SCAN_PAD_ENTRIES = 128;  // with dcnt: idctx and dcnt must be READABLE this many entries beyond m (the scan's loads are not clamped)
// dcnt (round 5; null: the kernel searches the range ends itself): per entry, depth | range length << 5 -- the number of entries
// behind it that lie in its cell (0x7FF: unknown, searched) -- as launch_tgroup_finish_compact leaves it; not used by gen 1 / 2
// mixed_scale (speed, never results): > 0: the tree has ranges long enough for the work list (the previous scan listed some): the
// kernel instance that keeps six waves per SIMD instead of seven; < 0: the previous scan listed none -- the work-list kernel is
// not launched, a range of up to 4 x 16384 candidates stays with its wave and a longer one sets KERR_BIGLIST_OVERFLOW (the caller
// falls back as for an overflowed list); 0: no expectation (the work-list kernel runs)
// gen (ids that may repeat; idctx from launch_entry_ctx with box == nullptr): 1 = MARK pass (sets the skipped bit of
// every entry whose id an ancestor carries, src/layer.rs:563-565; emits nothing), 2 = EMIT pass (all combos of two
// unskipped entries).  Run 1 then 2; the long-range counter in sc must be zero before each.
// owner_bounds / owner_n (instead of hist_plan): count the pairs per owner rank of the first id into hist[0..255]
// hist_plan / hist: also count the digits (pair >> shift[q]) & 255 of every emitted pair for the passes of
// hist_plan (n_passes <= SX_HIST_MAX, key digits only) into hist[HIST_COPIES][SX_HIST_MAX][256] (zeroed by the caller; the copies
// are summed by radix_sort, ready_copies = HIST_COPIES)
// The sizes of a finished expanding scan, on the device (for a sort enqueued without a host round trip): from the shard
// cursors and counts, sizes[SIZES_EXTENT] = slots of the block-structured output, sizes[SIZES_N] = pairs in it.  Whatever
// a later kernel could not cope with -- more blocks than out_blocks, more pairs than n_bound, a longer extent than
// extent_bound, an overflowed work list -- sets sizes[SIZES_MISS] and leaves both sizes 0: everything enqueued behind
// does nothing, and the host, which reads the sizes back at its one round trip, takes the blocking route.
hipError_t launch_scan_finalize(const uint64_t* shards, size_t out_blocks, size_t n_bound, size_t extent_bound, const DevScalars* sc,
                                uint32_t* sizes, hipStream_t stream);
// Batched Layer::test_box / test_ray: (query << 32 | id) records for every entry whose cell the reference's
// recursion reaches, block-structured like the scan's output (blockfill / shards zeroed by the caller).
// queries (device memory): n_queries rows of 2*dim f32 [min, max] (ray = 0) or 2*dim + 2 f32
// [origin, direction, range_min, range_max] (ray = 1)
// qtab / qtab_levels (optional): slice bounds of the cells of depths 1..qtab_levels (launch_query_table), read by
// the walk instead of bisecting the top of the tree for every query
hipError_t launch_region_query(int kind, int ray, const uint64_t* keys, const uint32_t* ids, size_t m, SysBounds sb,
                               const float* queries, uint32_t n_queries, int has_max_depth, uint32_t max_depth, uint64_t* out,
                               size_t out_blocks, uint32_t* blockfill, uint64_t* shards, hipStream_t stream,
                               const uint32_t* qtab = nullptr, uint32_t qtab_levels = 0);
uint32_t query_table_entries(int kind, uint32_t levels);
hipError_t launch_query_table(int kind, const uint64_t* keys, size_t m, uint32_t levels, uint32_t* qtab, hipStream_t stream);
// Batched Layer::pick_ray with the built-in ray/AABB distance: rays = n_queries rows [origin(dim), direction(dim),
// max_dist]; bounds = n_rows rows of 2*dim f32 indexed by id; out_id = 0xFFFFFFFF (out_dist = inf) when nothing is hit
hipError_t launch_pick_ray(int kind, const uint64_t* keys, const uint32_t* ids, size_t m, SysBounds sb, const float* rays,
                           uint32_t n_queries, int has_max_depth, uint32_t max_depth, const float* bounds, uint32_t n_rows,
                           float* out_dist, uint32_t* out_id, DevScalars* sc, hipStream_t stream);
// merge of two (key, id)-sorted runs A, B (box bytes ride along when bo != nullptr) into ko / io / bo
hipError_t launch_merge_runs(const uint64_t* ka, const uint32_t* ia, const uint8_t* ba, size_t na, const uint64_t* kb,
                             const uint32_t* ib, const uint8_t* bb, size_t nb, uint64_t* ko, uint32_t* io, uint8_t* bo,
                             hipStream_t stream, uint32_t* splits = nullptr);  // splits: merge_runs_split_words(na + nb) words of scratch (the tiles' splits in one small launch of their own)
size_t merge_runs_split_words(size_t n_total);
// dense copy of a block-structured stream (consumers that cannot read around the holes):
// block_offs[g] = exclusive prefix sum of blockfill (launch_scan_block_sums on a u64 copy)
hipError_t launch_compact_blocks(const uint64_t* in, size_t n_blocks, const uint32_t* blockfill,
                                 const unsigned long long* block_offs, uint64_t* out, hipStream_t stream);
hipError_t launch_blockfill_to_u64(const uint32_t* blockfill, size_t n_blocks, unsigned long long* out, hipStream_t stream);

// ---- dedup (K8): reference src/layer.rs:474,517 --------------------------------------------------
constexpr int UQ_THREADS = 256;
constexpr int UQ_ITEMS = 8;
constexpr int UQ_TILE = UQ_THREADS * UQ_ITEMS;
inline size_t uq_blocks(size_t n) { return (n + UQ_TILE - 1) / UQ_TILE; }
hipError_t launch_unique_count(const uint64_t* sorted, size_t n, unsigned long long* block_sums, hipStream_t stream);
// writes each surviving packed pair (id << 32 | id_) as interleaved u32 (id, id_)
hipError_t launch_unique_emit(const uint64_t* sorted, size_t n, const unsigned long long* block_offs,
                              uint32_t* pairs_out, hipStream_t stream);

// Finishing step after a radix sort on the FIRST id only: `in` holds n DISTINCT packed pairs grouped
// by first id (segments in arbitrary internal order); every segment is rank-sorted by second id
// inside an LDS window and written in the ABI layout.  Sets KERR_SEGMENT_TOO_LONG in sc->err when a
// segment does not fit the window (the caller then falls back to the full-width sort).
// window geometry (elements per workgroup + halo on each side), pairs / tree; small windows win: the
// kernels are latency-bound and LDS per workgroup sets the occupancy
#ifndef SEGP_CHUNK_DEF
#define SEGP_CHUNK_DEF 2048
#endif
#ifndef SEGP_HALO_DEF
#define SEGP_HALO_DEF 128
#endif
#ifndef SEGT_CHUNK_DEF
#define SEGT_CHUNK_DEF 1024
#endif
#ifndef SEGT_HALO_DEF
#define SEGT_HALO_DEF 128
#endif
constexpr int SEG_THREADS = 256;
constexpr int SEGP_CHUNK = SEGP_CHUNK_DEF, SEGP_HALO = SEGP_HALO_DEF;   // segment_finish
constexpr int SEGT_CHUNK = SEGT_CHUNK_DEF, SEGT_HALO = SEGT_HALO_DEF;   // tree_segment_finish
constexpr int SEG_SLOW_MAX = 4096;  // longest group the finish kernels rank straight from HBM before giving up
// Grouped finish of the tree sort (entries sorted by key >> gshift only, gshift >= 8):
// bounds[t] = first index whose key >> gshift is >= g0 + t (t = 0..n_groups); stats (optional, 2 words): largest group,
// entries in groups above tgroup_capacity().  launch_tgroup_finish orders every group by (key, id, position) in LDS.
hipError_t launch_tgroup_bounds(const uint64_t* keys, size_t m, uint32_t n_groups, uint64_t g0, int gshift, uint32_t* bounds,
                                uint32_t* stats, hipStream_t stream);
hipError_t launch_tgroup_finish(const uint64_t* keys_in, const uint32_t* ids_in, uint32_t n_groups, const uint32_t* bounds, int gshift,
                                int payload_packed, uint64_t* keys_out, uint32_t* ids_out, uint8_t* box_out, DevScalars* sc,
                                hipStream_t stream, int ctx_kind = -1, uint64_t* idctx_out = nullptr,
                                const uint32_t* stats = nullptr, uint32_t abort_over = 0xFFFFFFFFu, int distinct_id_bits = 0,
                                const struct CellRoute* cells = nullptr);
// cells: keys_in / ids_in are the gapped result of a cell sort (CellRoute::out_keys / out_vals) and the bounds came from
// launch_cell_count; only where tgroup_composite_applies(gshift, distinct_id_bits) (the composite-key kernel reads gapped input)
bool tgroup_composite_applies(int gshift, int distinct_id_bits);
// the same finish for a cell sort of COMPOSITE words (TreeCompact; comp_gshift: the group shift in the composite): reads the gapped
// words, orders every group by plain unsigned compare and writes index values, ids, box bytes and context words
// ids_out / box_out both null: not written (launch_ids_from_ctx derives them from idctx_out when wanted; every box code must be known)
hipError_t launch_ids_from_ctx(const uint64_t* idctx, uint32_t* ids_out, uint8_t* box_out, size_t n, hipStream_t stream);
bool tgroup_compact_applies(int comp_gshift);  // the composite's group shift leaves room for the finish kernel's bin digit
hipError_t launch_tgroup_finish_compact(const uint64_t* words_in, uint32_t n_groups, int comp_gshift, const TreeCompact& fmt, uint64_t* keys_out,
                                        uint32_t* ids_out, uint8_t* box_out, DevScalars* sc, hipStream_t stream, int ctx_kind, uint64_t* idctx_out,
                                        uint16_t* dcnt_out, const uint32_t* stats, uint32_t abort_over, const struct CellRoute& cells);
// dcnt_out (round 5): per entry, depth | range length << 5 -- what launch_scan_expand's dcnt takes
// distinct_id_bits > 0 (with payload_packed): the caller vouches that the ids are DISTINCT and below 2^distinct_id_bits; where
// the rest of the key and the id fit one 64-bit word the composite-key kernel runs (a group above the capacity then sets
// KERR_SEGMENT_TOO_LONG instead of taking the multi-round path)
// stats / abort_over: launch_tgroup_bounds' statistics; more than abort_over entries in oversize groups -> nothing is
// done and KERR_SEGMENT_TOO_LONG is set (the caller sorts full-width instead)
// ctx_kind (index kind) + idctx_out with payload_packed: also writes launch_entry_ctx's output for inactive_below = 0
int tgroup_capacity();
int tgroup_bin_bits();  // key bits below gshift the finish uses for its counting sort (gshift must be >= this)
// n DISTINCT packed pairs sorted by (first id >> gbits): groups g0 .. g0 + n_groups - 1 of 2^gbits first ids each
// (bounds: n_groups + 1 words of scratch).  Every group is put in final order in LDS and written in the ABI layout;
// a group of more than group_finish_capacity() pairs (or an id with very many) sets KERR_SEGMENT_TOO_LONG
hipError_t launch_group_finish(const uint64_t* in, size_t n, uint32_t n_groups, uint32_t g0, int gbits, uint32_t* bounds,
                               uint32_t* pairs_out, DevScalars* sc, hipStream_t stream, uint32_t* stats = nullptr,
                               uint32_t second_max = 0xFFFFFFFFu,  // an upper bound of the second ids (only speed depends on it)
                               const uint32_t* dev_sizes = nullptr,  // n is an upper bound, the count is dev_sizes[SIZES_N]
                               const struct CellRoute* cells = nullptr);  // `in` is the gapped result of a cell sort: bounds come from launch_cell_count (not computed here)
// stats (2 words of scratch): group-size statistics; with more than a quarter of the pairs in oversize groups the finish
// does nothing and sets KERR_SEGMENT_TOO_LONG
int group_finish_capacity();
// the same for pairs that may repeat: the first copy of every pair leaves, compacted inside its group's input range, in
// tmp (ABI layout); counts[0 .. n_groups] (counts[n_groups] = 0) take the groups' unique counts.  After an exclusive scan
// of counts (launch_scan_block_sums), launch_group_compact moves the groups to their final offsets.
hipError_t launch_group_finish_dups(const uint64_t* in, size_t n, uint32_t n_groups, uint32_t g0, int gbits, uint32_t* bounds, uint64_t* tmp,
                                    unsigned long long* counts, DevScalars* sc, hipStream_t stream, uint32_t* stats, uint32_t second_lo,
                                    uint32_t second_hi);
hipError_t launch_group_compact(const uint64_t* tmp, const uint32_t* bounds, const unsigned long long* offs, uint32_t n_groups,
                                uint32_t* pairs_out, hipStream_t stream);
int group_finish_dups_capacity();
int group_finish_dups_max_bits();
hipError_t launch_segment_finish(const uint64_t* in, size_t n, uint32_t* pairs_out, DevScalars* sc, hipStream_t stream);
// same for pairs that may repeat: equal pairs keep their input order, the output stays packed u64 (sorted), ready
// for unique_count / unique_emit.  sorted_out must not alias `in`.
hipError_t launch_segment_finish_dups(const uint64_t* in, size_t n, uint64_t* sorted_out, DevScalars* sc, hipStream_t stream);

// Finishing step of the tree sort after radix passes on the high key bits only: entries are grouped
// by (key >> seg_shift); every group is rank-sorted by (key, id, input position) in an LDS window and
// written out.  payload_packed: the u32 payload carries the box code in bits 27..31 (it is split into
// ids_out / box_out here).  Sets KERR_SEGMENT_TOO_LONG when a group does not fit the window.
hipError_t launch_tree_segment_finish(const uint64_t* keys_in, const uint32_t* ids_in, size_t m, int seg_shift,
                                      int payload_packed, uint64_t* keys_out, uint32_t* ids_out, uint8_t* box_out,
                                      DevScalars* sc, hipStream_t stream);

// interleaved (id, id_) u32 pairs <-> packed u64 (id << 32 | id_)
hipError_t launch_pack_pairs(const uint32_t* pairs, size_t n, uint64_t* packed, hipStream_t stream);
hipError_t launch_unpack_pairs(const uint64_t* packed, size_t n, uint32_t* pairs, hipStream_t stream);

}  // namespace bp
