"""Randomised differential test: random sequences of Layer operations on the HIP layer and on the
oracle side by side.  Targets the bookkeeping that selects kernels (id order / distinctness facts,
box codes, sorted flag, ping-pong buffers) rather than any single kernel."""
import numpy as np
import pytest

import oracle
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes

pytestmark = pytest.mark.gpu


def _tree(gpu, ref, what):
    k, i = gpu.tree()
    assert len(k) == len(ref), "%s: %d entries vs %d" % (what, len(k), len(ref))
    assert np.array_equal(k, ref.keys) and np.array_equal(i, ref.ids), what
    assert gpu.sorted == ref.sorted, what


def _narrow_rows(ref, pool):
    """one narrow-phase row per id (ids may be shared by objects: the table is simply a function of the id)"""
    n_ids = int(ref.ids.max()) + 1 if len(ref) else 1
    return np.ascontiguousarray(pool[np.arange(n_ids) % len(pool)], dtype=np.float32)


def _scan(gpu, ref, what, rng, pool=None):
    exp = ref.scan()
    mode = rng.integers(0, 6 if pool is not None else 4)
    if mode >= 4:   # the narrow phase on the device, inside the scan (a pure function of the two ids: filtering before or after the dedup is the same)
        dim = pool.shape[1] // 2
        rows = _narrow_rows(ref, pool)
        if mode == 4:
            a, b = rows[exp[:, 0]], rows[exp[:, 1]]
            keep = np.ones(len(exp), dtype=bool)
            for ax in range(dim):
                keep &= ~((a[:, ax] > b[:, dim + ax]) | (a[:, dim + ax] < b[:, ax]))       # geom.rs:112-119
            got = np.asarray(gpu.scan_filtered_aabb(rows))
        else:
            sph = np.ascontiguousarray(np.concatenate([(rows[:, :dim] + rows[:, dim:]) * np.float32(0.5),
                                                       ((rows[:, dim:] - rows[:, :dim]).max(axis=1) * np.float32(0.5))[:, None]], axis=1), dtype=np.float32)
            a, b = sph[exp[:, 0]], sph[exp[:, 1]]
            d2 = None
            for ax in range(dim):
                o = (b[:, ax] - a[:, ax]).astype(np.float32)
                d2 = o * o if d2 is None else (d2 + o * o).astype(np.float32)
            keep = ~(np.sqrt(d2, dtype=np.float32) > (a[:, dim] + b[:, dim]).astype(np.float32)) if len(exp) else np.zeros(0, dtype=bool)
            got = np.asarray(gpu.scan_filtered_sphere(sph))
        exp = exp[keep]
        got = got.reshape(-1, 2)
        assert got.shape == exp.shape and np.array_equal(got, exp), what + " (narrow phase %d)" % mode
        assert gpu.sorted and ref.sorted
        return
    if mode == 0:
        got = np.array(gpu.scan())
    elif mode == 1:
        got = np.array(gpu.par_scan())
    elif mode == 2:
        got = gpu.scan_device().to_numpy()
    else:
        got = np.array(gpu.scan_filtered(lambda a, b: True))
    assert got.shape == exp.shape and np.array_equal(got, exp), what
    assert gpu.sorted and ref.sorted


import os
N_SEEDS = int(os.environ.get("BP_FUZZ_SEEDS", "48"))   # BP_FUZZ_SEEDS=400 for a long soak


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_operation_sequences(seed):
    rng = np.random.default_rng(1000 + seed)
    min_depth = int(rng.choice([0, 0, 0, 3, 6]))
    n_pool = int(rng.choice([3000, 8000, 20000, 120000]))
    kind = [bp.Index64_3D, bp.Index64_3D, bp.Index64_2D, bp.Index32_2D][seed % 4 if seed >= 12 else 0]
    if seed % 4 == 3:
        sysb, bounds, _ = scenes.gen_boxes_powerlaw(n_pool, seed=seed, alpha=2.0)
    else:
        sysb, bounds, _ = scenes.gen_boxes(n_pool, seed=seed)
    if kind.dim == 2:
        sysb = np.ascontiguousarray(sysb[:, :2])
        min_depth = min(min_depth, 3)
        bounds = np.ascontiguousarray(bounds[:, [0, 1, 3, 4]])
    layers = [(bp.Layer(kind, min_depth), oracle.OracleLayer(kind.kind, min_depth)) for _ in range(2)]
    next_id = [0]
    log = []

    def extend(which):
        g, r = layers[which]
        n = int(rng.integers(1, n_pool // 3))
        rows = rng.integers(0, n_pool, size=n)
        style = rng.integers(0, 5)
        if style <= 1:      # fresh ascending ids (the fast-path case)
            ids = np.arange(next_id[0], next_id[0] + n, dtype=np.uint32)
            next_id[0] += n
        elif style == 2:    # shuffled but distinct
            ids = rng.permutation(np.arange(next_id[0], next_id[0] + n, dtype=np.uint32))
            next_id[0] += n
        elif style == 3:    # ids reused across objects (several boxes per id)
            ids = rng.integers(0, max(next_id[0], 50), size=n).astype(np.uint32)
        else:               # ascending but restarting from a lower value
            start = int(rng.integers(0, max(next_id[0], 1)))
            ids = np.arange(start, start + n, dtype=np.uint32)
            next_id[0] = max(next_id[0], start + n)
        b = bounds[rows].copy()
        if rng.random() < 0.2:
            b[: max(1, n // 50), b.shape[1] // 2:] += np.float32(1e6)      # some objects fall outside the system bounds
        g.extend(sysb, b, ids); r.extend(sysb, b, ids)
        log.append("extend[%d] n=%d style=%d" % (which, n, style))
        _tree(g, r, " | ".join(log[-6:]))

    for _step in range(int(rng.integers(6, 14))):
      try:
        op = rng.integers(0, 11)
        which = int(rng.integers(0, 2))
        g, r = layers[which]
        if op <= 3:
            extend(which)
        elif op == 4:
            (g.par_sort if rng.random() < 0.5 else g.sort)(); r.sort()
            log.append("sort[%d]" % which)
            _tree(g, r, " | ".join(log[-6:]))
        elif op == 5:
            log.append("scan[%d]" % which)
            _scan(g, r, " | ".join(log[-6:]), rng, bounds)
        elif op == 6:
            g.clear(); r.clear()
            log.append("clear[%d]" % which)
            assert len(g) == 0 and g.sorted
        elif op == 7:
            og, orf = layers[1 - which]
            g.merge(og); r.merge(orf)
            log.append("merge[%d<-%d]" % (which, 1 - which))
            _tree(g, r, " | ".join(log[-6:]))
            assert g.min_depth == r.min_depth
        elif op == 8:
            c = g.clone()
            assert c == g
            layers[which] = (c, r)       # keep working on the clone
            log.append("clone[%d]" % which)
        elif op == 10:       # region queries sort the layer first, like Layer::test (src/layer.rs:260)
            dim = kind.dim
            L = float(sysb[1][0])
            c = rng.uniform(0, L, size=(4, dim)); h = rng.uniform(0.5, L / 8, size=(4, dim))
            boxes = np.concatenate([c - h, c + h], axis=1).astype(np.float32)
            md = [None, 3, 7][int(rng.integers(0, 3))]
            rows = np.array(g.test_box_batch(sysb, boxes, md))
            log.append("test_box[%d]" % which)
            for q in range(4):
                exp = r.test_box(sysb, (boxes[q, :dim], boxes[q, dim:]), md)
                assert np.array_equal(rows[rows[:, 0] == q, 1], exp), " | ".join(log[-6:])
            d = rng.normal(size=dim).astype(np.float32)
            got = g.test_ray(sysb, c[0].astype(np.float32), d, 0.0, L / 4, md)
            assert np.array_equal(got, r.test_ray(sysb, c[0].astype(np.float32), d, 0.0, L / 4, md)), " | ".join(log[-6:])
            assert g.sorted and r.sorted
        else:
            raw_g = np.array(g.scan_raw()).reshape(-1, 2)
            raw_r, _ = r.scan_raw()
            log.append("scan_raw[%d]" % which)
            assert np.array_equal(raw_g, raw_r), " | ".join(log[-6:])
      except bp.BroadphaseError as e:
        raise AssertionError("%s after: %s" % (e, " | ".join(log))) from e
    for which in range(2):
        g, r = layers[which]
        _scan(g, r, "final scan[%d] after %s" % (which, " | ".join(log[-8:])), rng, bounds)


def _frame_scene(rng, n, style):
    """one frame's objects: uniform, shifted, clustered (dense consecutive ids), sparse ids, ids shuffled or repeated"""
    sysb, bounds, ids = scenes.gen_boxes(n, seed=int(rng.integers(0, 1 << 30)))
    L = float(sysb[1][0])
    if style == "clustered":
        k = int(rng.integers(n // 20, n // 4))
        size = (bounds[:k, 3:] - bounds[:k, :3]) * np.float32(rng.choice([0.02, 0.2]))
        lo = (np.float32(rng.uniform(0.1, 0.8)) * L + rng.uniform(0, float(rng.choice([0.0005, 0.004, 0.03])) * L, size=(k, 3))).astype(np.float32)
        bounds[:k] = np.concatenate([lo, lo + size], axis=1).astype(np.float32)
    elif style == "sparse_ids":
        ids = (ids.astype(np.uint64) * np.uint64(rng.choice([3, 17, 300]))).astype(np.uint32)
    elif style == "shuffled_ids":
        ids = rng.permutation(ids)
    elif style == "repeated_ids":
        ids = (ids // np.uint32(2)).astype(np.uint32)
    elif style == "dense_pairs":
        bounds[:, 3:] = bounds[:, :3] + (bounds[:, 3:] - bounds[:, :3]) * np.float32(2.2)
    return sysb, bounds, ids


# BP_FRAME_SEEDS=24 for a soak (~25 s per seed, mostly the oracle)
@pytest.mark.parametrize("seed", range(int(os.environ.get("BP_FRAME_SEEDS", "1"))))
def test_random_frame_sequences_on_one_layer(seed):
    """the sort plans adapt from frame to frame (grouped tree / pair finishes chosen from the previous frame's statistics,
    digit counts made by the producers for the predicted plan, context words written by the sort): whatever the order of
    scene kinds, every frame's tree and pair list must equal the oracle's"""
    rng = np.random.default_rng(7000 + seed)
    gpu, ref = bp.Layer(bp.Index64_3D, 0), oracle.OracleLayer(2, 0)
    styles = ["uniform", "uniform", "clustered", "sparse_ids", "shuffled_ids", "repeated_ids", "dense_pairs"]
    log = []
    for frame in range(7):
        style = str(rng.choice(styles)) if frame > 1 else "uniform"     # two uniform frames first: the grouped paths switch on
        n = 30000 if style == "dense_pairs" else int(rng.choice([30000, 70000]))   # the oracle pays for the pairs
        sysb, bounds, ids = _frame_scene(rng, n, style)
        log.append("%s/%d" % (style, n))
        gpu.clear(); ref.clear()
        if rng.random() < 0.3:                                          # the frame arrives in two extends
            h = n // 3
            gpu.extend(sysb, bounds[:h], ids[:h]); ref.extend(sysb, bounds[:h], ids[:h])
            gpu.extend(sysb, bounds[h:], ids[h:]); ref.extend(sysb, bounds[h:], ids[h:])
        else:
            gpu.extend(sysb, bounds, ids); ref.extend(sysb, bounds, ids)
        if rng.random() < 0.5:
            gpu.par_sort(); ref.sort()
            _tree(gpu, ref, "frames " + " ".join(log))
        exp = ref.scan()
        got = gpu.scan_device().to_numpy() if rng.random() < 0.5 else np.array(gpu.par_scan())
        assert got.shape == exp.shape and np.array_equal(got, exp), "frames " + " ".join(log)
        if rng.random() < 0.3:                                          # a second scan of the same tree (cached context words)
            got = np.array(gpu.scan())
            assert got.shape == exp.shape and np.array_equal(got, exp), "rescan, frames " + " ".join(log)
        if rng.random() < 0.5:                                          # ... and one with the narrow phase on the device (rows per id)
            n_ids = int(ids.max()) + 1
            rows = np.zeros((n_ids, 6), dtype=np.float32)
            rows[ids] = bounds                                          # (repeated ids: the last object's box -- any table is a valid predicate)
            a, b = rows[exp[:, 0]], rows[exp[:, 1]]
            keep = np.ones(len(exp), dtype=bool)
            for ax in range(3):
                keep &= ~((a[:, ax] > b[:, 3 + ax]) | (a[:, 3 + ax] < b[:, ax]))       # geom.rs:112-119
            got = np.asarray(gpu.scan_filtered_aabb(rows)).reshape(-1, 2)
            assert got.shape == exp[keep].shape and np.array_equal(got, exp[keep]), "narrow phase, frames " + " ".join(log)
