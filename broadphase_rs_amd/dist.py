"""Multi-GPU `Layer`: objects shard across the GPUs of one node, index entries and raw pairs are
exchanged by range over RCCL (torch.distributed, one process per GPU).

The reference is single-process; its only partitioning idea is the `min_depth` split of the sorted
array at cell boundaries (src/layer.rs:533-546), which works because an entry only ever interacts
with the entries whose cell contains its own.  The same property shards the path across GPUs:

  1. every rank encodes its own objects (extend)                                  [local, HIP]
  2. entries per Morton cell at `split_depth` are counted and all-reduced; contiguous cell ranges
     are assigned to ranks so that entry counts balance                           [allreduce, tiny]
  3. entries move to the owner of their cell (all-to-all); the few entries coarser than
     `split_depth` ("boundary keys": cells that span several ranges) are all-gathered so that
     every rank holds every possible ancestor of its own entries                  [all-to-all + allgather]
  4. every rank sorts and scans its shard; pairs whose deeper entry is a coarse (replicated)
     entry are emitted by rank 0 only                                             [local, HIP]
  5. raw pairs move to the owner of their first id (all-to-all); a pair can be produced on two
     ranks when an object's cells fall into different ranges, so duplicate removal has to happen
     after this exchange; each rank sorts + dedups its id range                   [all-to-all; local, HIP]

The concatenation of the per-rank results in rank order is exactly the single-GPU pair list.
All device work goes through the C ABI (`Layer`); torch supplies tensors and collectives only.
A backend object isolates the compute so that the exchange logic can be exercised with gloo on
CPU in tests (tests/test_dist_cpu.py provides an oracle-backed backend; the product has none).
"""
import numpy as np

from .layer import Index64_3D, Layer


def choose_owner_lut(hist_fine, n_ranks):
    """Contiguous, balanced assignment of Morton cells to ranks: cell c -> owner.  Deterministic in
    its inputs, so every rank computes the same table from the all-reduced histogram."""
    hist = np.asarray(hist_fine, dtype=np.float64)
    total = float(hist.sum())
    lut = np.zeros(len(hist), dtype=np.uint8)
    if total <= 0:
        return lut
    cum = np.cumsum(hist)
    # cell c goes to rank floor(G * (entries before the middle of c) / total)
    mid = cum - hist * 0.5
    owner = np.minimum((mid * n_ranks / total).astype(np.int64), n_ranks - 1)
    owner = np.maximum.accumulate(owner)  # contiguous ranges
    return owner.astype(np.uint8)


def id_range_bounds(max_id, n_ranks):
    """first id owned by rank 1..G-1 when [0, max_id] is cut into G equal ranges"""
    span = (int(max_id) + n_ranks) // n_ranks
    return np.array([min(span * (r + 1), 0xFFFFFFFF) for r in range(n_ranks - 1)], dtype=np.uint32)


M64 = 0xFFFFFFFFFFFFFFFF


def _varying_bits(lo, hi):
    """mask of the bits that can differ among integers in [lo, hi]"""
    if hi <= lo:
        return 0
    x = int(lo) ^ int(hi)
    return (1 << x.bit_length()) - 1


class HipBackend:
    """Device compute of one rank through the C ABI: a `local` layer (encode target) and a `shard`
    layer (sort/scan target)."""

    def __init__(self, index, min_depth, device):
        import torch
        self.torch = torch
        self.index = index
        self.device = torch.device("cuda", device)
        self.local = Layer(index, min_depth, device=device)
        self.shard = Layer(index, min_depth, device=device)

    # -- encode -------------------------------------------------------------------------------------
    def local_clear(self):
        self.local.clear()

    def local_extend(self, system_bounds, bounds, ids):
        self.local.extend(system_bounds, bounds, ids)

    def local_cell_histogram(self, split_depth):
        return self.local.cell_histogram(split_depth)

    def local_can_pack_box(self):
        return self.local.can_pack_box()

    def local_facts(self):
        return self.local.get_facts()

    def shard_begin(self, m):
        """device buffers of the shard layer for m incoming entries: (keys int64 tensor, ids int32 tensor)"""
        k, i = self.shard.begin_entries(m)
        return k.as_torch(), i.as_torch()

    def shard_commit(self, m, facts):
        self.shard.commit_entries(m, facts)

    def local_partition_entries(self, split_depth, lut, n_ranks, pack_box=False):
        counts, keys, ids = self.local.partition_entries_device(split_depth, lut, n_ranks, pack_box)
        return counts, keys.as_torch(), ids.as_torch()

    # -- shard ----------------------------------------------------------------------------------------
    def shard_set_entries(self, chunks, ids_carry_box=False, box_all_known=False):
        """chunks: [(keys int64 tensor, ids int32 tensor)] in an order for which equal index values
        carry non-decreasing ids; ids are distinct across objects."""
        self.shard.clear()
        for keys, ids in chunks:
            if keys.numel():
                self.shard.append_entries_assume(keys.data_ptr(), ids.data_ptr(), int(keys.numel()), True, True,
                                                 ids_carry_box, box_all_known)

    def shard_scan_pairs(self, coarse_depth, emit_coarse):
        """(packed raw pairs, arbitrary order; True if this rank suppressed every duplicate at the source)"""
        raw, unique = self.shard.scan_shard_pairs_device(coarse_depth, emit_coarse)
        return raw.as_torch(), unique

    def shard_par_scan(self, coarse_depth, emit_coarse):
        return self.shard.par_scan_shard_device(coarse_depth, emit_coarse)

    def partition_pairs(self, packed, bounds):
        counts, out = self.shard.partition_pairs_device(packed.data_ptr(), int(packed.numel()), bounds)
        return counts, out.as_torch()

    def shard_scan_partition(self, coarse_depth, emit_coarse, bounds):
        """shard_scan_pairs + partition_pairs in one device step -> (counts, grouped pairs, unique flag)"""
        counts, out, unique = self.shard.scan_shard_partition_device(coarse_depth, emit_coarse, bounds)
        return counts, out.as_torch(), unique

    def sort_unique_packed(self, packed, varying_mask, already_unique=False, first_range=None):
        if already_unique and first_range is not None:
            return self.shard.sort_packed_range_device(packed.data_ptr(), int(packed.numel()), varying_mask, *first_range)
        if already_unique:
            return self.shard.sort_packed_device(packed.data_ptr(), int(packed.numel()), varying_mask)
        return self.shard.sort_unique_packed_device(packed.data_ptr(), int(packed.numel()), varying_mask)

    def empty(self, n, dtype):
        return self.torch.empty(n, dtype=dtype, device=self.device)

    def sync(self):
        """collectives run on torch's streams, the layers on their own: fence between the two"""
        self.torch.cuda.synchronize(self.device)


class _HostAlloc:
    """stand-in for a backend when TorchComm stages an exchange through host memory"""

    def __init__(self, torch):
        self.torch, self.device = torch, "cpu"

    def empty(self, n, dtype):
        return self.torch.empty(n, dtype=dtype)

    def sync(self):
        pass


class TorchComm:
    """collectives of one rank over a torch.distributed process group (nccl = RCCL on ROCm, gloo on CPU)"""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        # gloo moves host memory only: device tensors are staged through the host (rehearsals of the multi-process
        # path on one GPU; the production backend is nccl = RCCL, which takes device tensors as they are)
        self.host_staged = dist.get_backend(group) == "gloo"
        import os
        if os.environ.get("BP_DIST_CHUNK_BYTES"):
            self.CHUNK_BYTES = max(8, int(os.environ["BP_DIST_CHUNK_BYTES"]))

    def all_reduce_sum(self, arr, backend):
        t = self.torch.from_numpy(np.asarray(arr).astype(np.int64)).to(self._small_device(backend))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def all_reduce_max(self, value, backend):
        t = self.torch.tensor([int(value)], dtype=self.torch.int64, device=self._small_device(backend))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _small_device(self, backend):
        return "cpu" if self.host_staged else backend.device

    # The RCCL build bundled with PyTorch-ROCm 2.10 silently moves only HALF of any single
    # send/recv larger than 1 GiB (measured on MI355X: tools/attic/dbg_a2a.py), so every exchange is cut
    # into per-peer pieces of at most CHUNK_BYTES.
    CHUNK_BYTES = 256 << 20

    def all_gather_counts(self, counts, backend):
        """every rank's count vector -> matrix C[src][k] (one small collective instead of one per exchange)"""
        torch, dist = self.torch, self.dist
        # 64-bit fields (key OR / AND) travel as their two's-complement image: torch has no uint64 tensors to reduce
        mine = torch.from_numpy(np.array([int(c) & 0xFFFFFFFFFFFFFFFF for c in counts], dtype=np.uint64).view(np.int64).copy()
                                ).to(self._small_device(backend))
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.group)
        return torch.stack(parts).cpu().numpy()      # ONE device-to-host copy (a copy per part is a sync per rank)

    def all_to_all_v(self, send, send_counts, dtype, backend, recv_counts=None, out=None, sync=True):
        """variable-size all-to-all of a tensor that is already grouped by destination.  sync=False leaves the
        exchange in flight (the caller's next exchange or backend.sync() fences it): back-to-back exchanges overlap"""
        torch, dist = self.torch, self.dist
        send_counts = [int(c) for c in send_counts]
        if self.host_staged and send.is_cuda:  # gloo rehearsal: exchange host copies, then put the result where it belongs
            got = self.all_to_all_v(send.cpu(), send_counts, dtype, _HostAlloc(torch), recv_counts, None, True)
            recv = out[: got.numel()] if out is not None else backend.empty(got.numel(), dtype)
            recv.copy_(got)
            if sync:
                backend.sync()
            return recv
        if recv_counts is None:
            sc = torch.tensor(send_counts, dtype=torch.int64, device=send.device)
            rc = torch.empty_like(sc)
            dist.all_to_all_single(rc, sc, group=self.group)
            recv_counts = [int(c) for c in rc.cpu().tolist()]
        else:
            recv_counts = [int(c) for c in recv_counts]
        recv = out[: sum(recv_counts)] if out is not None else backend.empty(sum(recv_counts), dtype)
        ch = max(1, self.CHUNK_BYTES // send.element_size())
        # the largest single message of ANY rank decides the number of rounds; with the counts matrix
        # known everywhere this needs no collective (callers pass global_max), else one all-reduce
        local_max = max(send_counts + recv_counts)
        gmax = getattr(self, "_global_max_hint", None)
        rounds = (int(gmax) + ch - 1) // ch if gmax is not None else self.all_reduce_max((local_max + ch - 1) // ch, backend)
        soff = np.concatenate([[0], np.cumsum(send_counts)]).astype(np.int64)
        roff = np.concatenate([[0], np.cumsum(recv_counts)]).astype(np.int64)
        if rounds <= 1:
            dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts,
                                   group=self.group)
        else:
            for r in range(rounds):  # all_to_all_single works on every backend (gloo has no list all_to_all)
                isz = [min(c, (r + 1) * ch) - min(c, r * ch) for c in send_counts]
                osz = [min(c, (r + 1) * ch) - min(c, r * ch) for c in recv_counts]
                stage_in = torch.cat([send[int(soff[d]) + min(send_counts[d], r * ch):int(soff[d]) + min(send_counts[d], (r + 1) * ch)]
                                      for d in range(self.world)])
                stage_out = backend.empty(sum(osz), dtype)
                dist.all_to_all_single(stage_out, stage_in, output_split_sizes=osz, input_split_sizes=isz, group=self.group)
                o = 0
                for d in range(self.world):
                    a = int(roff[d]) + min(recv_counts[d], r * ch)
                    recv[a:a + osz[d]] = stage_out[o:o + osz[d]]
                    o += osz[d]
        if sync:
            backend.sync()
        return recv

    def all_gather_v(self, t, dtype, backend, sizes=None):
        """all-gather of per-rank tensors of different lengths, concatenated in rank order"""
        torch, dist = self.torch, self.dist
        if self.host_staged and t.is_cuda:
            got = self.all_gather_v(t.cpu(), dtype, _HostAlloc(torch), sizes)
            out = backend.empty(got.numel(), dtype)
            out.copy_(got)
            backend.sync()
            return out
        if sizes is None:
            n = torch.tensor([t.numel()], dtype=torch.int64, device=t.device)
            got = [torch.empty_like(n) for _ in range(self.world)]
            dist.all_gather(got, n, group=self.group)
            sizes = [int(x.item()) for x in got]
        sizes = [int(x) for x in sizes]
        mx = max(sizes)
        if mx == 0:
            return backend.empty(0, dtype)
        pad = backend.empty(mx, dtype)
        pad[: t.numel()] = t
        if t.numel() < mx:
            pad[t.numel():] = 0
        parts = [backend.empty(mx, dtype) for _ in range(self.world)]
        dist.all_gather(parts, pad, group=self.group)
        out = torch.cat([p[:n_] for p, n_ in zip(parts, sizes)])
        backend.sync()
        return out


class ShardedLayer:
    """`Layer` across the ranks of a torch.distributed process group (weak scaling: every rank
    extends its own objects).  Method names follow the reference; `par_scan_device` returns this
    rank's slice (a contiguous first-id range) of the global sorted, duplicate-free pair list."""

    def __init__(self, index=Index64_3D, min_depth=0, device=0, n_global=None, split_depth=None, group=None,
                 backend=None, comm=None, pair_order="global"):
        import torch
        self.torch = torch
        # "global": rank r returns the r-th first-id range of the reference's sorted list (concatenation in rank
        # order == Layer::par_scan, bit for bit).  "local": every rank returns, sorted and duplicate-free, the
        # pairs found in its own Morton range -- the ranks' lists are disjoint and their union is the reference's
        # set, but no pair crosses the wire.  Only possible while every rank suppresses duplicates at the source
        # (box codes known everywhere); otherwise the call silently takes the "global" route.
        if pair_order not in ("global", "local"):
            raise ValueError("pair_order must be 'global' or 'local'")
        self.pair_order = pair_order
        self.comm = comm if comm is not None else TorchComm(group)
        self.rank, self.world = self.comm.rank, self.comm.world
        self.index = index
        self.backend = backend if backend is not None else HipBackend(index, min_depth, device)
        # cells at split_depth are the unit of ownership: 8^3 = 512 (3-D) / 4^4 = 256 (2-D) cells
        self.split_depth = split_depth if split_depth is not None else (3 if index.dim == 3 else 4)
        self.n_global = n_global
        self.max_id_local = -1
        self.last = {}
        if hasattr(self.backend, "local") and hasattr(self.backend.local, "set_cell_histogram_depth"):
            self.backend.local.set_cell_histogram_depth(self.split_depth)   # extend counts the splitter histogram itself

    @property
    def shard_layer(self):
        return self.backend.shard

    def clear(self):
        self.backend.local_clear()
        self.max_id_local = -1

    def extend(self, system_bounds, bounds, ids):
        self.backend.local_extend(system_bounds, bounds, ids)
        if self.n_global is None and len(ids):
            if hasattr(ids, "max") and hasattr(ids, "is_cuda"):
                self.max_id_local = max(self.max_id_local, int(ids.to(self.torch.int64).max().item()) & 0xFFFFFFFF)
            else:
                self.max_id_local = max(self.max_id_local, int(np.asarray(ids).max()))

    # ---- the pipeline --------------------------------------------------------------------------------------
    def par_scan_device(self):
        import time
        torch, G, be, comm = self.torch, self.world, self.backend, self.comm
        d_s = self.split_depth
        t_prev = [time.perf_counter()]
        stages = {}

        def lap(name):  # host-side stage clock (every backend call is complete on return)
            now = time.perf_counter()
            stages[name] = stages.get(name, 0.0) + (now - t_prev[0]) * 1e3
            t_prev[0] = now
        # 2. balanced contiguous Morton ranges from the all-reduced cell histogram
        hist = be.local_cell_histogram(d_s).astype(np.int64)
        lap("cell_hist")
        # every rank's facts travel with its histogram (ONE small all-gather, summed locally): who can pack box
        # codes into the ids, id ranges, OR/AND of keys and ids -- the receiving shard layers need no re-reduction
        lf = be.local_facts()
        can_pack = bool(lf["flags"] & 4)
        # ... and its entry count: a rank without entries (no objects, or all of them outside the system bounds) must
        # not take part in the AND / min / max reductions below
        fvec = [lf["key_or"], lf["key_and"], lf["id_or"], lf["id_and"], lf["id_min"], lf["id_max"], lf["depth_mask"], lf["flags"],
                int(hist.sum())]
        both = comm.all_gather_counts(list(hist) + fvec, be)
        hist_all = both[:, :len(hist)].sum(axis=0)
        lap("allgather_hist_facts")
        lut = choose_owner_lut(hist_all[:-1], G)
        # 3. entries -> owners (fine), boundary keys -> everyone (coarse).  Packing must be decided
        #    before the partition, by every rank alike (from the gathered flags)
        fm = both[:, len(hist):]
        nonempty = fm[:, 8] > 0                                          # the rank has entries
        flags_all = fm[:, 7]
        pack = bool(np.all((flags_all[nonempty] & 4) != 0)) if nonempty.any() else False
        all_known = pack and bool(np.all((flags_all[nonempty] & 8) != 0))
        counts, keys, ids = be.local_partition_entries(d_s, lut, G, pack)
        n_fine = int(counts[:G].sum())
        lap("partition_entries")
        cm = comm.all_gather_counts(counts, be)                          # cm[src][dst], cm[src][G] = coarse
        recv_counts = cm[:, self.rank]
        coarse_sizes = cm[:, G]
        n_recv, n_coarse = int(recv_counts.sum()), int(coarse_sizes.sum())
        sk, si = be.shard_begin(n_recv + n_coarse)
        comm._global_max_hint = int(cm[:, :G].max())
        comm.all_to_all_v(keys[:n_fine], counts[:G], torch.int64, be, recv_counts, out=sk, sync=False)   # keys and ids in flight together
        comm.all_to_all_v(ids[:n_fine], counts[:G], torch.int32, be, recv_counts, out=si)
        comm._global_max_hint = None
        if n_coarse:
            sk[n_recv:] = comm.all_gather_v(keys[n_fine:n_fine + int(counts[G])], torch.int64, be, coarse_sizes)
            si[n_recv:] = comm.all_gather_v(ids[n_fine:n_fine + int(counts[G])], torch.int32, be, coarse_sizes)
            be.sync()
        lap("exchange_entries")
        # facts of the union.  Chunks arrive in rank order and every rank's entries are in its own
        # append order, so the order / distinctness facts hold iff they hold locally everywhere and
        # the ranks' id ranges ascend with the rank (fine and coarse entries never share an index value)
        ne = np.flatnonzero(nonempty)
        ascending = bool(np.all(fm[ne[:-1], 5] < fm[ne[1:], 4])) if len(ne) > 1 else True
        facts = {
            "key_or": int(np.bitwise_or.reduce(fm[ne, 0])) & M64 if len(ne) else 0,
            "key_and": int(np.bitwise_and.reduce(fm[ne, 1])) & M64 if len(ne) else 0,
            "id_or": int(np.bitwise_or.reduce(fm[ne, 2])) if len(ne) else 0, "id_and": int(np.bitwise_and.reduce(fm[ne, 3])) if len(ne) else 0,
            "id_min": int(fm[ne, 4].min()) if len(ne) else 0, "id_max": int(fm[ne, 5].max()) if len(ne) else 0,
            "depth_mask": int(np.bitwise_or.reduce(fm[ne, 6])) if len(ne) else 0,
            "flags": (1 if (ascending and bool(np.all((flags_all[ne] & 1) != 0))) else 0)
                     | (2 if (ascending and bool(np.all((flags_all[ne] & 2) != 0))) else 0)
                     | (4 if pack else 0) | (8 if all_known else 0),
        }
        be.shard_commit(n_recv + n_coarse, facts)
        lap("shard_commit")
        # 4. local sort + scan (single pass); coarse-coarse pairs only once.  With box codes every pair
        #    is emitted by exactly one entry on exactly one rank, so no duplicate ever crosses the wire.
        #    The raw pairs go straight to the owner of their first id (5.): scan and bucketing are one
        #    backend step where the backend can fuse them.
        if self.n_global is not None:
            max_id = int(self.n_global) - 1
        else:
            max_id = max(comm.all_reduce_max(self.max_id_local, be), 0)
        bounds = id_range_bounds(max_id, G)
        if self.pair_order == "local" and pack and all_known and hasattr(be, "shard_par_scan"):
            # every rank knows all its box codes, so every pair is emitted exactly once, on one rank
            out, unique = be.shard_par_scan(d_s, self.rank == 0)
            lap("sort_scan_sort_pairs")
            if bool(np.all(comm.all_gather_counts([0 if unique else 1], be)[:, 0] == 0)):
                self.last = {"entries_local": int(counts.sum()), "entries_shard": n_recv + n_coarse, "coarse": n_coarse,
                             "raw_pairs": len(out), "raw_pairs_recv": 0, "pairs": len(out), "id_range": (0, max_id),
                             "box_codes": True, "unique_at_source": True, "pair_order": "local",
                             "stage_ms": {k: round(v, 3) for k, v in stages.items()}}
                return out
            # a shard could not vouch for uniqueness after all: take the exchange route (re-scans)
        if hasattr(be, "shard_scan_partition"):
            pcounts, grouped, unique = be.shard_scan_partition(d_s, self.rank == 0, bounds)
            lap("sort_scan_partition")
        else:
            raw, unique = be.shard_scan_pairs(d_s, self.rank == 0)
            lap("sort_and_scan")
            pcounts, grouped = be.partition_pairs(raw, bounds)
            lap("partition_pairs")
        n_raw = int(np.asarray(pcounts).sum())
        # the final step may skip dedup only if EVERY rank suppressed its duplicates at the source (a
        # wrong assumption here would corrupt the result, not just slow it): the flag rides with the counts
        mine = 0 if (unique and pack and all_known) else 1
        pm_f = comm.all_gather_counts(list(np.asarray(pcounts, dtype=np.int64)) + [mine], be)
        pm = pm_f[:, :G]
        unique = bool(np.all(pm_f[:, G] == 0))
        comm._global_max_hint = int(pm.max())
        rp = comm.all_to_all_v(grouped, pcounts, torch.int64, be, pm[:, self.rank])
        comm._global_max_hint = None
        lap("exchange_pairs")
        lo = 0 if self.rank == 0 else int(bounds[self.rank - 1])
        hi = max_id if self.rank == G - 1 else int(bounds[self.rank]) - 1
        varying = (_varying_bits(lo, max(hi, lo)) << 32) | _varying_bits(0, max_id)
        out = be.sort_unique_packed(rp, varying, unique, (lo, max(hi, lo)))         # this rank's first-id range
        lap("sort_pairs")
        self.last = {"entries_local": int(counts.sum()), "entries_shard": n_recv + n_coarse,
                     "coarse": n_coarse, "raw_pairs": n_raw, "raw_pairs_recv": int(rp.numel()),
                     "pairs": len(out), "id_range": (lo, hi), "box_codes": bool(pack), "unique_at_source": bool(unique),
                     "pair_order": "global", "stage_ms": {k: round(v, 3) for k, v in stages.items()}}
        return out

    # reference names
    def par_scan(self):
        out = self.par_scan_device()
        return out.to_numpy() if hasattr(out, "to_numpy") else out

    def scan(self):
        return self.par_scan()
