"""GPU tests of the sharded pipeline's HIP building blocks.  The pool gives one GPU per box, so
(a) a 1-rank NCCL group runs the production path (TorchComm + HipBackend) end to end, and
(b) N rank-threads in one process run the real kernels of N HipBackends on the same GPU with an
    in-memory stand-in for the collectives -- the multi-rank data flow on device buffers."""
import os
import threading

import numpy as np
import pytest

import oracle
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes
from broadphase_rs_amd.dist import HipBackend, ShardedLayer

pytestmark = pytest.mark.gpu


class ThreadComm:
    """collectives among rank-threads of one process (test stand-in for RCCL)"""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.sh, self.rank, self.world = shared, rank, shared.world

    def _exchange(self, value):
        self.sh.slots[self.rank] = value
        self.sh.barrier.wait()
        vals = list(self.sh.slots)
        self.sh.barrier.wait()
        return vals

    def all_reduce_sum(self, arr, backend):
        return np.sum(self._exchange(np.asarray(arr).astype(np.int64)), axis=0)

    def all_reduce_max(self, value, backend):
        return max(self._exchange(int(value)))

    def all_gather_counts(self, counts, backend):
        return np.stack(self._exchange(np.asarray([int(c) for c in counts], dtype=np.int64)))

    def all_to_all_v(self, send, send_counts, dtype, backend, recv_counts=None, out=None, sync=True):
        import torch
        offs = np.concatenate([[0], np.cumsum([int(c) for c in send_counts])])
        vals = self._exchange([send[int(offs[d]):int(offs[d + 1])].clone() for d in range(self.world)])
        parts = [vals[src][self.rank] for src in range(self.world)]
        n = sum(p.numel() for p in parts)
        res = torch.cat(parts) if n else backend.empty(0, dtype)
        if out is not None:
            out[:n] = res
            res = out[:n]
        backend.sync()
        self.sh.barrier.wait()
        return res

    def all_gather_v(self, t, dtype, backend, sizes=None):
        import torch
        vals = self._exchange(t.clone())
        out = torch.cat(vals) if sum(v.numel() for v in vals) else backend.empty(0, dtype)
        backend.sync()
        self.sh.barrier.wait()
        return out


def _giants(parts, world):
    for r, (sysb, b, _ids) in enumerate(parts):
        L = float(sysb[1][0])
        b[0] = [0, 0, 0, L, L, L]
        b[1] = [0.1 * L * (r + 1) / world, 0, 0, 0.1 * L * (r + 1) / world + 0.4 * L, 0.5 * L, 0.45 * L]
        b[2] = [0.3 * L, 0.3 * L, 0.3 * L, 0.62 * L, 0.62 * L, 0.62 * L]


def _run_rank_threads(parts, world, frames=1, pair_order="global", n_global=None):
    """one ShardedLayer per rank-thread on the same GPU (real kernels, ThreadComm exchanges) -> [(pairs, last, stats)]"""
    import torch
    torch.cuda.init()
    shared = ThreadComm.Shared(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            lay = ShardedLayer(bp.Index64_3D, 0, device=0, backend=HipBackend(bp.Index64_3D, 0, 0), comm=ThreadComm(shared, rank),
                               pair_order=pair_order, n_global=n_global)
            sysb, bounds, ids = parts[rank]
            d_b, d_i = torch.from_numpy(bounds).cuda(), torch.from_numpy(ids.view(np.int32)).cuda()
            for _frame in range(frames):
                lay.clear()
                lay.extend(sysb, d_b, d_i)
                out = lay.par_scan()
            results[rank] = (out, lay.last, lay.shard_layer.stats())
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    return results


def _oracle_pairs(parts):
    ref = oracle.OracleLayer(2, 0)
    for sysb, b, ids in parts:
        ref.extend(sysb, b, ids)
    return ref.par_scan()


@pytest.mark.parametrize("world,n_total,alpha", [(4, 200000, 1.5), (8, 240000, 2.5), (8, 200000, 1.5)])
def test_rank_threads_powerlaw_scene_matches_the_oracle(world, n_total, alpha):
    """configs[4]'s shape at test size: mixed-scale (power-law) boxes sharded over 4 / 8 ranks.  The heavy tail is the
    boundary-key all-gather's real load: every box coarser than split_depth is replicated on every rank."""
    parts = [scenes.gen_boxes_powerlaw_shard(n_total, r, world, seed=21, alpha=alpha) for r in range(world)]
    results = _run_rank_threads(parts, world, frames=2)
    expect = _oracle_pairs(parts)
    got = np.concatenate([results[r][0] for r in range(world)], axis=0)
    assert got.shape == expect.shape and np.array_equal(got, expect)
    coarse = [results[r][1]["coarse"] for r in range(world)]
    # the same boundary keys on every rank; alpha = 1.5 gives a real heavy tail (hundreds of boxes coarser than split_depth)
    assert len(set(coarse)) == 1 and coarse[0] > (400 if alpha < 2.0 else 0)
    for r in range(world):
        assert results[r][2]["flags"] == 3


def test_rank_threads_world8_at_the_true_per_gpu_share_of_10m():
    """BASELINE configs[3] as stated: 10 M Index64_3D objects in total over 8 ranks (1.25 M each), every rank's slice of
    the pair list compared with the oracle's list of the union scene"""
    world, n_total = 8, 10_000_000
    parts = [scenes.gen_boxes_shard(n_total, r, world, seed=0) for r in range(world)]
    results = _run_rank_threads(parts, world, frames=1, n_global=n_total)
    expect = _oracle_pairs(parts)
    got = np.concatenate([results[r][0] for r in range(world)], axis=0)
    assert got.shape == expect.shape and np.array_equal(got, expect)
    loads = [results[r][1]["entries_shard"] for r in range(world)]
    assert max(loads) < 1.2 * (sum(loads) / world)


@pytest.mark.parametrize("world,empty", [(3, {1: "none"}), (3, {0: "oob", 2: "none"})])
def test_rank_threads_with_ranks_that_hold_no_entries(world, empty):
    parts = [scenes.gen_boxes_shard(world * 30000, r, world, seed=13) for r in range(world)]
    _giants(parts, world)
    for r, mode in empty.items():
        sysb, b, ids = parts[r]
        parts[r] = (sysb, b[:0], ids[:0]) if mode == "none" else (sysb, b + np.float32(4.0) * np.float32(sysb[1][0]), ids)
    results = _run_rank_threads(parts, world, frames=2)
    expect = _oracle_pairs(parts)
    got = np.concatenate([np.asarray(results[r][0]).reshape(-1, 2) for r in range(world)], axis=0)
    assert len(expect) > 0 and got.shape == expect.shape and np.array_equal(got, expect)
    for r in empty:
        assert results[r][1]["entries_local"] == 0


@pytest.mark.parametrize("world,n_per_rank,giants", [(2, 40000, True), (4, 25000, True), (3, 30000, False)])
def test_rank_threads_on_one_gpu_match_the_oracle(world, n_per_rank, giants):
    import torch
    torch.cuda.init()
    parts = [scenes.gen_boxes_shard(world * n_per_rank, r, world, seed=3) for r in range(world)]
    if giants:
        _giants(parts, world)
    shared = ThreadComm.Shared(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            lay = ShardedLayer(bp.Index64_3D, 0, device=0, backend=HipBackend(bp.Index64_3D, 0, 0), comm=ThreadComm(shared, rank))
            sysb, bounds, ids = parts[rank]
            for _frame in range(2):
                lay.clear()
                lay.extend(sysb, torch.from_numpy(bounds).cuda(), torch.from_numpy(ids.view(np.int32)).cuda())
                out = lay.par_scan()
            results[rank] = (out, lay.last, lay.shard_layer.stats())
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    ref = oracle.OracleLayer(2, 0)
    for sysb, b, ids in parts:
        ref.extend(sysb, b, ids)
    expect = ref.scan()
    got = np.concatenate([results[r][0] for r in range(world)], axis=0)
    assert got.shape == expect.shape and np.array_equal(got, expect)
    for r in range(world):
        assert results[r][2]["flags"] == 3          # shard layers ran the fast (key-only sort, distinct-id) kernels
        if giants:
            assert results[r][1]["coarse"] >= 3 * world
    loads = [results[r][1]["entries_shard"] for r in range(world)]
    assert max(loads) < 1.5 * (sum(loads) / world) + 1000


@pytest.mark.parametrize("world,n_per_rank,giants", [(3, 30000, False), (2, 40000, True)])
def test_local_pair_order_is_a_disjoint_cover_of_the_reference_list(world, n_per_rank, giants):
    """pair_order="local": no pair crosses the wire; the ranks' sorted lists are disjoint and their union is
    exactly the reference's list (falls back to the exchange when a rank cannot vouch for uniqueness)"""
    import torch
    torch.cuda.init()
    parts = [scenes.gen_boxes_shard(world * n_per_rank, r, world, seed=5) for r in range(world)]
    if giants:
        _giants(parts, world)
    shared = ThreadComm.Shared(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            lay = ShardedLayer(bp.Index64_3D, 0, device=0, backend=HipBackend(bp.Index64_3D, 0, 0), comm=ThreadComm(shared, rank),
                               pair_order="local")
            sysb, bounds, ids = parts[rank]
            lay.extend(sysb, torch.from_numpy(bounds).cuda(), torch.from_numpy(ids.view(np.int32)).cuda())
            results[rank] = (lay.par_scan(), lay.last)
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    ref = oracle.OracleLayer(2, 0)
    for sysb, b, ids in parts:
        ref.extend(sysb, b, ids)
    expect = ref.scan()
    orders = {results[r][1]["pair_order"] for r in range(world)}
    assert len(orders) == 1                                     # every rank took the same route
    packed = []
    for r in range(world):
        out = results[r][0]
        pk = (out[:, 0].astype(np.uint64) << np.uint64(32)) | out[:, 1].astype(np.uint64)
        assert np.all(pk[1:] > pk[:-1])                         # sorted, duplicate-free
        packed.append(pk)
    allp = np.concatenate(packed)
    assert len(allp) == len(expect)                             # disjoint ...
    allp.sort()
    exp = (expect[:, 0].astype(np.uint64) << np.uint64(32)) | expect[:, 1].astype(np.uint64)
    assert np.array_equal(allp, exp)                            # ... and a cover
    if not giants:
        assert orders == {"local"} and all(results[r][1]["raw_pairs_recv"] == 0 for r in range(world))
        assert min(len(pk) for pk in packed) > 0


def test_single_rank_nccl_group_runs_the_production_path():
    import torch
    import torch.distributed as dist
    from broadphase_rs_amd.dist import TorchComm
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        # regression: exchanges above 1 GiB must arrive complete (this RCCL halves single messages > 1 GiB)
        class _Be:
            device = torch.device("cuda", 0)

            def empty(self, n, dtype):
                return torch.empty(n, dtype=dtype, device=self.device)

            def sync(self):
                torch.cuda.synchronize()
        big = torch.arange(150_000_000, dtype=torch.int64, device="cuda") * 7 + 3       # 1.2 GB
        got_big = TorchComm().all_to_all_v(big, [big.numel()], torch.int64, _Be())
        assert torch.equal(got_big, big)
        del big, got_big
        sysb, bounds, ids = scenes.gen_boxes(120000, seed=9)
        lay = ShardedLayer(bp.Index64_3D, 0, device=0, n_global=len(ids))
        lay.clear()
        lay.extend(sysb, torch.from_numpy(bounds).cuda(), torch.from_numpy(ids.view(np.int32)).cuda())
        got = lay.par_scan()
        ref = oracle.OracleLayer(2, 0)
        ref.extend(sysb, bounds, ids)
        expect = ref.scan()
        assert got.shape == expect.shape and np.array_equal(got, expect)
    finally:
        dist.destroy_process_group()
