"""world_size > 1 on CPU (gloo): the exchange logic of broadphase_rs_amd/dist.py (owner table,
all-to-all of entries, all-gather of boundary keys, all-to-all of raw pairs, per-range dedup) with
the oracle standing in for the device kernels, checked against a single-process oracle scan of
the whole scene.  The oracle-backed backend exists only here, as the checker's compute."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from broadphase_rs_amd import scenes
from broadphase_rs_amd.dist import ShardedLayer, choose_owner_lut, id_range_bounds
from broadphase_rs_amd.layer import Index64_3D


class _Result:
    def __init__(self, pairs):
        self.pairs = pairs

    def __len__(self):
        return len(self.pairs)

    def to_numpy(self):
        return self.pairs


class OracleBackend:
    """numpy + oracle implementation of the backend interface of dist.HipBackend (tests only)"""

    def __init__(self, kind, min_depth):
        self.kind, self.min_depth = kind, min_depth
        self.device = torch.device("cpu")
        self.local = oracle.OracleLayer(kind, min_depth)
        self.shard = oracle.OracleLayer(kind, min_depth)
        dim, _bits, db, ab = oracle.KINDS[kind]
        self.dim, self.db, self.top = dim, db, dim * ab + db

    def local_clear(self):
        self.local.clear()

    def local_extend(self, sysb, bounds, ids):
        self.local.extend(sysb, bounds, ids)

    def _cells(self, keys, split_depth):
        depth = keys & np.uint64((1 << self.db) - 1)
        cell = keys >> np.uint64(self.top - self.dim * split_depth)
        return depth < split_depth, cell.astype(np.int64)

    def local_cell_histogram(self, split_depth):
        n_cells = 1 << (self.dim * split_depth)
        coarse, cell = self._cells(self.local.keys, split_depth)
        hist = np.zeros(n_cells + 1, dtype=np.uint64)
        hist[:n_cells] = np.bincount(cell[~coarse], minlength=n_cells)
        hist[n_cells] = coarse.sum()
        return hist

    def local_can_pack_box(self):
        return False, False

    def local_facts(self):
        keys, ids = self.local.keys, self.local.ids
        if len(keys) == 0:
            # what bp_layer_get_facts reports for an empty layer: neutral but representable AND fields
            return {"key_or": 0, "key_and": 0, "id_or": 0, "id_and": 0, "id_min": 0xFFFFFFFF, "id_max": 0,
                    "depth_mask": 0, "flags": 3}
        asc = bool(np.all(np.diff(ids.astype(np.int64)) >= 0))
        return {"key_or": int(np.bitwise_or.reduce(keys)), "key_and": int(np.bitwise_and.reduce(keys)),
                "id_or": int(np.bitwise_or.reduce(ids)), "id_and": int(np.bitwise_and.reduce(ids)),
                "id_min": int(ids.min()), "id_max": int(ids.max()),
                "depth_mask": int(np.bitwise_or.reduce(np.uint64(1) << (keys & np.uint64((1 << self.db) - 1)))),
                "flags": (1 if asc else 0) | 2}

    def shard_begin(self, m):
        self._sk = torch.zeros(m, dtype=torch.int64)
        self._si = torch.zeros(m, dtype=torch.int32)
        return self._sk, self._si

    def shard_commit(self, m, facts):
        assert not (facts["flags"] & 4)
        self.shard.clear()
        if m:
            self.shard.append_entries(self._sk.numpy().view(np.uint64)[:m], self._si.numpy().view(np.uint32)[:m])

    def local_partition_entries(self, split_depth, lut, n_ranks, pack_box=False):
        assert not pack_box
        keys, ids = self.local.keys, self.local.ids
        coarse, cell = self._cells(keys, split_depth)
        dest = np.where(coarse, n_ranks, np.asarray(lut)[np.minimum(cell, len(lut) - 1)]).astype(np.int64)
        order = np.argsort(dest, kind="stable")
        counts = np.bincount(dest, minlength=n_ranks + 1).astype(np.uint64)
        return (counts, torch.from_numpy(keys[order].view(np.int64).copy()), torch.from_numpy(ids[order].view(np.int32).copy()))

    def shard_set_entries(self, chunks, ids_carry_box=False, box_all_known=False):
        assert not ids_carry_box
        self.shard.clear()
        for k, i in chunks:
            if k.numel():
                self.shard.append_entries(k.numpy().view(np.uint64), i.numpy().view(np.uint32))

    def shard_scan_pairs(self, coarse_depth, emit_coarse):
        return self.shard_scan_raw(coarse_depth, emit_coarse), False

    def shard_scan_raw(self, coarse_depth, emit_coarse):
        raw, _ = self.shard.scan_raw()
        if len(raw) and not emit_coarse:
            keys, ids = self.shard.keys, self.shard.ids
            depth_of = dict(zip(ids.tolist(), (keys & np.uint64((1 << self.db) - 1)).tolist()))
            keep = np.array([depth_of[a] >= coarse_depth for a in raw[:, 0].tolist()], dtype=bool)
            raw = raw[keep]
        packed = (raw[:, 0].astype(np.uint64) << np.uint64(32)) | raw[:, 1].astype(np.uint64)
        return torch.from_numpy(packed.view(np.int64).copy())

    def partition_pairs(self, packed, bounds):
        p = packed.numpy().view(np.uint64)
        owner = np.searchsorted(np.asarray(bounds, dtype=np.uint64), p >> np.uint64(32), side="right")
        order = np.argsort(owner, kind="stable")
        counts = np.bincount(owner, minlength=len(bounds) + 1).astype(np.uint64)
        return counts, torch.from_numpy(p[order].view(np.int64).copy())

    def sort_unique_packed(self, packed, varying_mask, already_unique=False, first_range=None):
        assert not already_unique
        p = np.unique(packed.numpy().view(np.uint64))
        if first_range is not None and len(p):   # the range the caller promises must hold
            assert first_range[0] <= int(p[0] >> np.uint64(32)) and int(p[-1] >> np.uint64(32)) <= first_range[1]
        if varying_mask:  # the mask must cover every bit that differs between the pairs
            assert len(p) == 0 or int(np.bitwise_or.reduce(p ^ p[0])) & ~int(varying_mask) == 0
        return _Result(np.stack([(p >> np.uint64(32)).astype(np.uint32), (p & np.uint64(0xFFFFFFFF)).astype(np.uint32)], axis=1))

    def empty(self, n, dtype):
        return torch.empty(n, dtype=dtype)

    def sync(self):
        pass


def _scene(world, n_per_rank, seed, with_giants):
    parts = [scenes.gen_boxes_shard(world * n_per_rank, r, world, seed=seed) for r in range(world)]
    if with_giants:  # a few boxes far coarser than split_depth: boundary keys on every rank
        for r, (sysb, b, _ids) in enumerate(parts):
            L = float(sysb[1][0])
            b[0] = [0, 0, 0, L, L, L]
            b[1] = [0.1 * L * (r + 1) / world, 0, 0, 0.1 * L * (r + 1) / world + 0.4 * L, 0.5 * L, 0.45 * L]
            b[2] = [0.3 * L, 0.3 * L, 0.3 * L, 0.62 * L, 0.62 * L, 0.62 * L]
    return parts


def _emptied(parts, empty):
    """empty = {rank: "none" | "oob"}: that rank extends nothing / only objects outside the system bounds"""
    out = []
    for r, (sysb, b, ids) in enumerate(parts):
        mode = (empty or {}).get(r)
        if mode == "none":
            b, ids = b[:0], ids[:0]
        elif mode == "oob":
            b = b + np.float32(4.0) * np.float32(sysb[1][0])     # every box beyond the system bounds: skipped silently
        out.append((sysb, b, ids))
    return out


def _worker(rank, world, port, n_per_rank, seed, with_giants, split_depth, q, pair_order="global", empty=None):
    if split_depth == 2:
        os.environ["BP_DIST_CHUNK_BYTES"] = "4096"   # force the multi-round (chunked) exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        parts = _emptied(_scene(world, n_per_rank, seed, with_giants), empty)
        sysb, bounds, ids = parts[rank]
        lay = ShardedLayer(Index64_3D, 0, n_global=None, split_depth=split_depth, backend=OracleBackend(2, 0), pair_order=pair_order)
        for _round in range(2):  # reuse across "frames"
            lay.clear()
            lay.extend(sysb, bounds[: len(ids) // 2], ids[: len(ids) // 2])
            lay.extend(sysb, bounds[len(ids) // 2:], ids[len(ids) // 2:])
            out = lay.par_scan()
        q.put((rank, out, lay.last))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# the last case asks for pair_order="local": this backend cannot vouch for uniqueness at the source, so every rank
# must take the exchange route and the result is still the sorted partition
@pytest.mark.parametrize("world,n_per_rank,with_giants,split_depth,pair_order",
                         [(2, 1500, True, 3, "global"), (2, 2500, False, 2, "global"), (3, 800, True, 3, "global"),
                          (2, 1200, False, 3, "local")])
def test_sharded_pipeline_equals_single_process(world, n_per_rank, with_giants, split_depth, pair_order):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_per_rank, 7, with_giants, split_depth, q, pair_order)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, out, info = q.get(timeout=180)
        results[rank] = (out, info)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts = _scene(world, n_per_rank, 7, with_giants)
    ref = oracle.OracleLayer(2, 0)
    for sysb, b, ids in parts:
        ref.extend(sysb, b, ids)
    expect = ref.scan()
    got = np.concatenate([results[r][0] for r in range(world)], axis=0)   # rank order = first-id order
    assert got.shape == expect.shape and np.array_equal(got, expect)
    if with_giants:
        assert all(results[r][1]["coarse"] >= 3 * world for r in range(world))  # boundary keys replicated everywhere
    assert all(results[r][1]["pair_order"] == "global" for r in range(world))
    for r in range(world):                                                   # every rank holds its own id range only
        lo, hi = results[r][1]["id_range"]
        o = results[r][0]
        assert len(o) == 0 or (o[:, 0].min() >= lo and o[:, 0].max() <= hi)


@pytest.mark.parametrize("world,empty,split_depth", [(3, {1: "none"}, 3), (2, {0: "oob"}, 2), (3, {0: "none", 2: "oob"}, 2)])
def test_sharded_pipeline_with_ranks_that_hold_no_entries(world, empty, split_depth):
    """a rank with no objects, or with every object outside the system bounds, contributes neutral facts and empty
    exchanges; split_depth 2 also forces the chunked multi-round exchange (such a rank used to crash on
    key_and = 2^64 - 1 and hang the others)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 1500, 11, True, split_depth, q, "global", empty)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, out, info = q.get(timeout=180)
        results[rank] = (out, info)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    parts = _emptied(_scene(world, 1500, 11, True), empty)
    ref = oracle.OracleLayer(2, 0)
    for sysb, b, ids in parts:
        ref.extend(sysb, b, ids)
    expect = ref.scan()
    assert len(expect) > 0
    got = np.concatenate([results[r][0].reshape(-1, 2) for r in range(world)], axis=0)
    assert got.shape == expect.shape and np.array_equal(got, expect)
    for r, mode in empty.items():
        assert results[r][1]["entries_local"] == 0


def test_owner_table_is_contiguous_and_balanced():
    rng = np.random.default_rng(0)
    hist = rng.integers(0, 1000, size=512)
    for g in (1, 2, 3, 8):
        lut = choose_owner_lut(hist, g)
        assert lut.min() == 0 and lut.max() == g - 1 and (np.diff(lut.astype(int)) >= 0).all()
        loads = np.bincount(lut, weights=hist, minlength=g)
        assert loads.max() <= hist.sum() / g + hist.max()
    assert (choose_owner_lut(np.zeros(64), 4) == 0).all()
    assert id_range_bounds(99, 4).tolist() == [25, 50, 75]
    assert id_range_bounds(0, 2).tolist() == [1]
