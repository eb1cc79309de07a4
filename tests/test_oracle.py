"""CPU tests of the oracle (oracle/bp_oracle.c) against the reference's own known-answer tests,
against the independent Python restatement (oracle/py_oracle.py), and against brute force.

KAT sources (reference file:line):
  * src/index.rs:343-364  decode/encode constants
  * src/index.rs:367-374  round_trip_axis
  * src/geom.rs:696-706   system_bounds round trip
  * tests/test_layer.rs:42-54, 63, 81, 98, 115   is_sorted / strictly-increasing assertions
  * SURVEY.md 8(c) worked example (surveyor-derived, hand-checked; not produced by the Rust)
"""
import numpy as np
import pytest

import oracle
from oracle import py_oracle as P
from broadphase_rs_amd import scenes

K3 = oracle.INDEX64_3D


def test_codec_kat_index_rs_343_364():
    L = oracle.lib()
    # index.rs:356-359
    assert L.orc_encode_axis(K3, 0o1777777 << 13) == 0o0001111111111111111111
    assert L.orc_encode_axis(K3, 0) == 0
    # index.rs:344-351
    assert L.orc_decode_axis(K3, 0o0001111111111111111111) == (0o1777777 << 13)
    assert L.orc_decode_axis(K3, 0o0006666666666666666666) == 0
    # same constants in hex as recorded in SURVEY.md 8(c)
    assert L.orc_encode_axis(K3, 0xFFFFE000) == 0x0049249249249249


def test_codec_round_trip_index_rs_367_374():
    L = oracle.lib()
    rng = np.random.default_rng(0)
    for v in rng.integers(0, 0o2000000, size=10000):
        expected = int(v) << 13
        assert L.orc_decode_axis(K3, L.orc_encode_axis(K3, expected)) == expected


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_codec_matches_naive_bit_spread(kind):
    L = oracle.lib()
    rng = np.random.default_rng(kind + 1)
    vals = [0, 0xFFFFFFFF, 0x80000000, 1, 0x12345678] + [int(v) for v in rng.integers(0, 2**32, size=3000)]
    ab = oracle.KINDS[kind][3]
    for v in vals:
        e = L.orc_encode_axis(kind, v)
        assert e == P.encode_axis(kind, v)
        assert L.orc_decode_axis(kind, e) == (v >> (32 - ab)) << (32 - ab)
        assert L.orc_decode_axis(kind, e) == P.decode_axis(kind, e)


def test_level_masks_survey_8a_row6():
    L = oracle.lib()
    assert L.orc_level_mask(2, 0) == 0
    assert L.orc_level_mask(2, 1) == 0x3800000000000000
    assert L.orc_level_mask(2, 19) == 0x3FFFFFFFFFFFFFE0
    assert L.orc_level_mask(0, 1) == 0xC0000000
    assert L.orc_level_mask(0, 14) == 0xFFFFFFF0
    for kind in (0, 1, 2):
        for d in range(0, oracle.KINDS[kind][3] + 1):
            assert L.orc_level_mask(kind, d) == P.level_mask(kind, d)


def test_to_local_kat_geom_rs_696_706():
    smin, smax = [-64.0] * 3, [64.0] * 3
    lo = oracle.to_local(smin, smax, [-32.0] * 3)
    hi = oracle.to_local(smin, smax, [32.0] * 3)
    assert list(lo) == [0x3FFFFFC0] * 3
    assert list(hi) == [0xBFFFFF00] * 3      # exact 0xBFFFFF40 is not representable in f32
    assert list(oracle.to_global(smin, smax, lo)) == [-32.0] * 3
    assert list(oracle.to_global(smin, smax, hi)) == [32.0] * 3
    assert P.to_local(smin, smax, [-32.0] * 3) == [0x3FFFFFC0] * 3
    assert P.to_local(smin, smax, [32.0] * 3) == [0xBFFFFF00] * 3


def test_to_local_matches_python_f32():
    rng = np.random.default_rng(5)
    smin = np.array([-3.5, 0.0, 10.0], dtype=np.float32)
    smax = np.array([100.25, 7.0, 11.0], dtype=np.float32)
    for _ in range(2000):
        g = (smin + (smax - smin) * rng.random(3, dtype=np.float32)).astype(np.float32)
        assert list(oracle.to_local(smin, smax, g)) == P.to_local(smin, smax, g)
    # saturating cast semantics: NaN -> 0, below -> 0, far above -> u32::MAX
    assert list(oracle.to_local([0.0], [1.0], [float("nan")])) == [0]
    assert list(oracle.to_local([0.0], [1.0], [-5.0])) == [0]
    assert list(oracle.to_local([0.0], [1.0], [2.0])) == [0xFFFFFFFF]


WORKED = dict(
    sysb=([0.0, 0.0, 0.0], [16.0, 16.0, 16.0]),
    bounds=[[1, 1, 1, 2, 2, 2], [1.5, 1.5, 1.5, 3, 3, 3], [0, 0, 0, 16, 16, 16], [7, 7, 7, 9, 9, 9],
            [20, 0, 0, 21, 1, 1]],
    ids=[0, 1, 2, 3, 4],
)


def test_worked_example_survey_8c():
    lay = oracle.OracleLayer(K3, 0)
    lay.extend(WORKED["sysb"], WORKED["bounds"], WORKED["ids"])
    keys, ids = lay.keys, lay.ids
    assert lay.num_invalid == 1 and not lay.sorted
    assert len(keys) == 25
    # id 0: depth 4, 8 cells 0x4 + c * 0x0004_0000_0000_0000 (z outer, y, x inner == ascending here)
    assert [int(k) for k in keys[:8]] == [0x4 + c * 0x0004000000000000 for c in range(8)]
    assert [int(k) for k in keys[8:16]] == [0x3 + c * 0x0020000000000000 for c in range(8)]
    assert int(keys[16]) == 0 and int(ids[16]) == 2
    assert [int(k) for k in keys[17:25]] == [0x07E0000000000003, 0x0EC0000000000003, 0x15A0000000000003,
                                             0x1C80000000000003, 0x2360000000000003, 0x2A40000000000003,
                                             0x3120000000000003, 0x3800000000000003]
    # local coordinates quoted in the table
    assert list(oracle.to_local(*WORKED["sysb"], [1, 1, 1])) == [0x0FFFFFF0] * 3
    assert list(oracle.to_local(*WORKED["sysb"], [2, 2, 2])) == [0x1FFFFFE0] * 3
    assert list(oracle.to_local(*WORKED["sysb"], [16, 16, 16])) == [0xFFFFFF00] * 3
    raw, max_stack = lay.scan_raw()
    assert len(raw) == 32 and max_stack == 3
    assert lay.sorted
    sk, si = lay.keys, lay.ids
    assert (int(sk[0]), int(si[0])) == (0x0, 2)
    assert (int(sk[1]), int(si[1])) == (0x3, 1)
    assert (int(sk[2]), int(si[2])) == (0x4, 0)
    pairs = lay.scan()
    assert pairs.tolist() == [[0, 1], [0, 2], [1, 2], [3, 2]]
    assert lay.par_scan(4).tolist() == pairs.tolist()


def _is_sorted(rows, unique):
    """tests/test_layer.rs:42-54"""
    for a, b in zip(rows[:-1], rows[1:]):
        if a > b or (unique and a == b):
            return False
    return True


@pytest.mark.parametrize("kind,n,min_depth", [(2, 300, 0), (2, 1000, 0), (2, 1000, 3), (0, 500, 4), (1, 500, 0)])
def test_c_oracle_matches_python_restatement(kind, n, min_depth):
    if kind == 2:
        sysb, bounds, ids = scenes.gen_boxes(n, seed=11)
    else:
        sysb, bounds, ids = scenes.gen_circles_2d(n, seed=11)
    c = oracle.OracleLayer(kind, min_depth)
    p = P.PyLayer(kind, min_depth)
    c.extend(sysb, bounds, ids)
    p.extend(sysb, bounds, ids)
    assert list(zip(c.keys.tolist(), c.ids.tolist())) == p.tree       # unsorted order is part of the contract
    c.sort(); p.sort()
    tree = list(zip(c.keys.tolist(), c.ids.tolist()))
    assert tree == p.tree and _is_sorted(tree, unique=False)
    raw_c, ms_c = c.scan_raw()
    raw_p, ms_p = p.scan_raw()
    assert [tuple(r) for r in raw_c.tolist()] == raw_p and ms_c == ms_p
    pairs = [tuple(r) for r in c.scan().tolist()]
    assert pairs == p.scan() and _is_sorted(pairs, unique=True)
    assert [tuple(r) for r in c.par_scan(8).tolist()] == pairs
    assert [tuple(r) for r in c.par_scan(1).tolist()] == pairs


def test_no_false_negatives_vs_brute_force():
    sysb, bounds, ids = scenes.gen_boxes(3000, seed=0)
    lay = oracle.OracleLayer(K3, 0)
    lay.extend(sysb, bounds, ids)
    got = {(min(a, b), max(a, b)) for a, b in lay.scan().tolist()}
    truth = {tuple(r) for r in oracle.brute_force_overlaps(bounds).tolist()}
    assert len(truth) > 100
    assert truth <= got


def test_filter_applied_before_dedup_and_merge_semantics():
    sysb, bounds, ids = scenes.gen_boxes(800, seed=3)
    a = oracle.OracleLayer(K3, 2)
    b = oracle.OracleLayer(K3, 0)
    a.extend(sysb, bounds[:400], ids[:400])
    b.extend(sysb, bounds[400:], ids[400:])
    a.sort()
    a.merge(b)                                   # layer.rs:127-138
    assert a.min_depth == 0 and not a.sorted and len(a) > 800
    full = [tuple(r) for r in a.scan().tolist()]
    even = [tuple(r) for r in a.scan_filtered(lambda x, y: (x + y) % 2 == 0).tolist()]
    assert even == [p for p in full if (p[0] + p[1]) % 2 == 0]
    assert [tuple(r) for r in a.par_scan_filtered(lambda x, y: (x + y) % 2 == 0, 4).tolist()] == even


def test_duplicate_ids_skip_rule_and_edge_boxes():
    sysb = ([0.0] * 3, [64.0] * 3)
    bounds = [[0, 0, 0, 64, 64, 64],      # whole system, depth 0
              [1, 1, 1, 2, 2, 2],          # small box, same id as the first -> skipped everywhere
              [1, 1, 1, 2, 2, 2],          # identical cells, different id
              [5, 5, 5, 5, 5, 5],          # zero-size box -> depth clamp 19
              [float("nan"), 1, 1, 2, 2, 2]]
    ids = [7, 7, 9, 3, 4]
    c, p = oracle.OracleLayer(K3, 0), P.PyLayer(K3, 0)
    c.extend(sysb, bounds, ids); p.extend(sysb, bounds, ids)
    assert list(zip(c.keys.tolist(), c.ids.tolist())) == p.tree
    assert (c.keys[np.asarray(c.ids) == 3] & 31).tolist() == [19]
    raw_c, _ = c.scan_raw()
    raw_p, _ = p.scan_raw()
    assert [tuple(r) for r in raw_c.tolist()] == raw_p
    pairs = [tuple(r) for r in c.scan().tolist()]
    assert pairs == p.scan()
    assert all(a != b for a, b in pairs)
    assert (9, 7) in pairs and (3, 7) in pairs


# ---- region queries: the two restatements of test_impl must agree -----------------
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_region_queries_recursive_c_vs_closed_form_python(kind):
    from broadphase_rs_amd import scenes
    dim = 2 if kind != 2 else 3
    sysb, bounds, ids = scenes.gen_boxes(300, seed=kind)
    if dim == 2:
        sysb = np.ascontiguousarray(sysb[:, :2]); bounds = np.ascontiguousarray(bounds[:, [0, 1, 3, 4]])
    c, p = oracle.OracleLayer(kind, 0), P.PyLayer(kind, 0)
    c.extend(sysb, bounds, ids); p.extend(sysb, bounds, ids)
    rng = np.random.default_rng(kind)
    L = float(sysb[1][0])
    hits = 0
    for q in range(25):
        ctr = rng.uniform(0, L, dim); h = rng.uniform(0.5, L / 6, dim)
        tb = ((ctr - h).astype(np.float32), (ctr + h).astype(np.float32))
        for md in (None, 0, 3, 6):
            got = c.test_box(sysb, tb, md)                       # layer.rs:294-313
            assert got.tolist() == p.test_box(sysb, tb, md)
            assert np.all(got[1:] > got[:-1])                    # results.sort(); results.dedup() (layer.rs:274-275)
        # broadphase guarantee (geom.rs:112-119 inclusive overlap): nothing that truly overlaps is missing
        true = [i for i in range(len(bounds))
                if not any(bounds[i][a] > tb[1][a] or bounds[i][dim + a] < tb[0][a] for a in range(dim))]
        assert set(true) <= set(c.test_box(sysb, tb).tolist())
        org = rng.uniform(-0.2 * L, 1.2 * L, dim).astype(np.float32)
        d = rng.normal(size=dim).astype(np.float32)
        if q % 5 == 0:
            d[0] = 0.0                                           # infinite / NaN plane distances (geom.rs:640-643)
        lo, hi = [(-np.inf, np.inf), (0.0, np.inf), (0.0, L / 3)][q % 3]
        for md in (None, 4):
            got = c.test_ray(sysb, org, d, lo, hi, md)           # layer.rs:327-351
            assert got.tolist() == p.test_ray(sysb, org, d, lo, hi, md)
            hits += len(got)
    assert hits > 0
    # the worked example of SURVEY 8(c): objects 0..3 in [0,16]^3
    if kind == 2:
        w = oracle.OracleLayer(2, 0)
        sb = ([0.0] * 3, [16.0] * 3)
        w.extend(sb, [[1, 1, 1, 2, 2, 2], [1.5, 1.5, 1.5, 3, 3, 3], [0, 0, 0, 16, 16, 16], [7, 7, 7, 9, 9, 9]], [0, 1, 2, 3])
        assert w.test_box(sb, ([1.0] * 3, [2.0] * 3)).tolist() == [0, 1, 2]
        assert w.test_box(sb, ([10.0] * 3, [12.0] * 3), 2).tolist() == [2, 3]
        assert w.test_ray(sb, [0.5, 8.0, 8.0], [1.0, 0.0, 0.0]).tolist() == [2, 3]


def test_pick_ray_oracle_finds_the_nearest_box():
    # Layer::pick_ray (layer.rs:418-446) with the built-in ray/AABB distance: the pruned, ordered traversal must
    # return the same nearest distance as a brute-force slab test over all objects
    from broadphase_rs_amd import scenes
    sysb, bounds, ids = scenes.gen_boxes(1500, seed=9)
    c = oracle.OracleLayer(K3, 0)
    c.extend(sysb, bounds, ids)
    rng = np.random.default_rng(4)
    L = float(sysb[1][0])
    f = np.float32
    hits = 0
    for q in range(120):
        org = rng.uniform(0, L, 3).astype(f); d = rng.normal(size=3).astype(f)
        if q % 6 == 0:
            d[q % 3] = 0
        maxd = f(L if q % 2 else L / 20)
        got = c.pick_ray_aabb(sysb, org, d, maxd, bounds)
        best = None
        for i in range(len(bounds)):
            tmin, tmax, ok = f(0), f(np.inf), True
            for a in range(3):
                if d[a] == 0:
                    if org[a] < bounds[i][a] or org[a] > bounds[i][3 + a]:
                        ok = False
                        break
                else:
                    t0 = f(f(bounds[i][a] - org[a]) / d[a]); t1 = f(f(bounds[i][3 + a] - org[a]) / d[a])
                    if t0 > t1:
                        t0, t1 = t1, t0
                    tmin = max(tmin, t0); tmax = min(tmax, t1)
            if ok and tmin <= tmax and tmin < maxd and (best is None or tmin < best):
                best = tmin
        if got is None:
            assert best is None
        else:
            hits += 1
            assert best is not None and f(got[0]) == best
    assert hits > 30
