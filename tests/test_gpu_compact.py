"""Compact records of the per-frame tree sort (TreeCompact, csrc/bp_kernels.h): once a layer's frames have settled into the
cell route, `extend` writes one 64-bit word per entry -- the index bits that vary, the id, the box code -- and the radix passes
move 8-byte records; the finish expands them into the (index, id) tree.  Everything observable must stay what the reference
produces (src/layer.rs:94-165, 449-520): the unsorted tree in append order when something asks for it before the sort, the
sorted tree, the pair list -- against the oracle, frame by frame, for the three index kinds."""
import numpy as np
import pytest

import oracle
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes

pytestmark = pytest.mark.gpu

KINDS = {0: bp.Index32_2D, 1: bp.Index64_2D, 2: bp.Index64_3D}


def base_scene(kind, n, seed):
    if kind == 2:
        return scenes.gen_boxes(n, seed=seed)
    return scenes.gen_circles_2d(n, seed=seed)


def moved(kind, sysb, base, frame, amount=0.5):
    """every object displaced by up to +- amount x its size per axis (seeded), kept inside the system bounds"""
    dim = 3 if kind == 2 else 2
    rng = np.random.default_rng(1000 + frame)
    size = base[:, dim:] - base[:, :dim]
    lo = np.asarray(sysb[0], dtype=np.float32)[:dim]
    hi = np.asarray(sysb[1], dtype=np.float32)[:dim]
    mn = base[:, :dim] + (rng.random((len(base), dim), dtype=np.float32) - np.float32(0.5)) * np.float32(2 * amount) * size
    mn = np.minimum(np.maximum(mn, lo), hi - size).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([mn, mn + size], axis=1).astype(np.float32))


def check_tree(gpu, ref, what):
    k, i = gpu.tree()
    assert len(k) == len(ref.keys), "%s: %d entries, oracle %d" % (what, len(k), len(ref.keys))
    assert np.array_equal(k, ref.keys), "%s: keys differ (first at %d)" % (what, int(np.flatnonzero(k != ref.keys)[0]))
    assert np.array_equal(i, ref.ids), "%s: ids differ (first at %d)" % (what, int(np.flatnonzero(i != ref.ids)[0]))
    assert gpu.sorted == ref.sorted, what


def check_pairs(got, exp, what):
    got = np.asarray(got)
    assert got.shape == exp.shape, "%s: %s pairs, oracle %s" % (what, got.shape, exp.shape)
    assert np.array_equal(got, exp), what


def warm(gpu, kind, sysb, base, ids, frames, first=0, min_depth=0):
    """frames of the moving scene through clear / extend / par_sort / par_scan, each checked against the oracle"""
    for f in range(first, first + frames):
        b = moved(kind, sysb, base, f)
        ref = oracle.OracleLayer(kind, min_depth)
        gpu.clear()
        gpu.extend(sysb, b, ids); ref.extend(sysb, b, ids)
        gpu.par_sort(); ref.sort()
        check_tree(gpu, ref, "frame %d, par_sort" % f)
        check_pairs(gpu.par_scan(), ref.scan(), "frame %d, par_scan" % f)


@pytest.mark.parametrize("kind,n", [(2, 200000), (1, 250000), (0, 250000)])
def test_compact_records_over_moving_frames(kind, n, monkeypatch):
    monkeypatch.setenv("BP_RADIX_CLAIM_MIN_TILES", "1")   # (the claiming passes, and with them the cell routes, start at ~2 M records by default)
    sysb, base, ids = base_scene(kind, n, seed=7 + kind)
    gpu = bp.Layer(KINDS[kind], 0)
    warm(gpu, kind, sysb, base, ids, 7)
    st = gpu.stats()
    want = ("tree_sorts_in_cells", "tree_sorts_compact", "tree_expands", "extend_reruns", "sort_repairs", "tree_sort_passes")
    # frame 0 takes the classic plan and probes, frame 1 the grouped plan, frame 2 the cell route; from then on extend may write
    # compact records -- and nothing but the sort asks for the tree before it is sorted, so nothing is expanded
    assert st["tree_sorts_compact"] >= 3 and st["tree_expands"] == 0 and st["sort_repairs"] == 0, {k: st[k] for k in want}


def test_compact_records_with_objects_of_more_than_two_cells_per_axis(monkeypatch):
    """min_depth above the natural depth of the larger objects: they span three or more cells on an axis (geom.rs:299-301 warns),
    their box code is unknown, and the emit kernel walks such a round object by object instead of dealing entries to lanes --
    the words, the tree and the pairs stay the oracle's"""
    monkeypatch.setenv("BP_RADIX_CLAIM_MIN_TILES", "1")
    kind, n, min_depth = 2, 200000, 6
    sysb, base, ids = base_scene(kind, n, seed=51)
    gpu = bp.Layer(KINDS[kind], min_depth)
    warm(gpu, kind, sysb, base, ids, 6, min_depth=min_depth)
    st = gpu.stats()
    assert st["tree_sorts_compact"] >= 2 and st["sort_repairs"] == 0, {k: st[k] for k in ("tree_sorts_compact", "tree_sorts_in_cells", "tree_expands", "extend_reruns")}


def test_compact_records_are_expanded_for_every_other_consumer(monkeypatch):
    """once the layer writes compact records, anything but the frame's own sort that touches the tree first -- iter() of the
    unsorted tree, a second extend, merge, clone, equality, raw-order scans -- sees exactly the (index, id) entries in append
    order (src/layer.rs:79-81, 94-138)"""
    monkeypatch.setenv("BP_RADIX_CLAIM_MIN_TILES", "1")
    kind, n = 2, 150000
    sysb, base, ids = base_scene(kind, n, seed=21)
    gpu = bp.Layer(KINDS[kind], 0)
    warm(gpu, kind, sysb, base, ids, 4)
    assert gpu.stats()["tree_sorts_compact"] >= 1

    def frame(f):
        b = moved(kind, sysb, base, f)
        ref = oracle.OracleLayer(kind, 0)
        gpu.clear()
        gpu.extend(sysb, b, ids); ref.extend(sysb, b, ids)
        return b, ref

    # (a) the unsorted tree, then the rest of the frame
    e0 = gpu.stats()["tree_expands"]
    _, ref = frame(10)
    check_tree(gpu, ref, "unsorted tree after a compact extend")
    assert gpu.stats()["tree_expands"] == e0 + 1
    gpu.par_sort(); ref.sort()
    check_tree(gpu, ref, "sorted after expansion")
    check_pairs(gpu.par_scan(), ref.scan(), "scan after expansion")
    warm(gpu, kind, sysb, base, ids, 2, first=11)     # back on the route
    # (b) a second extend appends to the expanded entries
    _, ref = frame(13)
    sysb2, extra, _ = scenes.gen_boxes(5000, seed=99)
    extra = (extra * np.float32(float(sysb[1][0]) / float(sysb2[1][0]) * 0.999)).astype(np.float32)
    extra_ids = (np.arange(5000, dtype=np.uint32) + np.uint32(n + 10))
    gpu.extend(sysb, extra, extra_ids); ref.extend(sysb, extra, extra_ids)
    check_tree(gpu, ref, "second extend behind a compact one")
    check_pairs(gpu.par_scan(), ref.scan(), "scan of both extends")
    warm(gpu, kind, sysb, base, ids, 3, first=14)
    # (c) merge of a static layer into the frame's layer
    _, ref = frame(17)
    stat_gpu = bp.Layer(KINDS[kind], 0); stat_ref = oracle.OracleLayer(kind, 0)
    stat_gpu.extend(sysb, extra, extra_ids); stat_ref.extend(sysb, extra, extra_ids)
    stat_gpu.par_sort(); stat_ref.sort()
    gpu.merge(stat_gpu); ref.merge(stat_ref)
    check_tree(gpu, ref, "merge into a compact tree")
    check_pairs(gpu.par_scan(), ref.scan(), "scan after merge")
    warm(gpu, kind, sysb, base, ids, 3, first=18)
    # (d) clone and equality
    _, ref = frame(21)
    twin = gpu.clone()
    assert twin == gpu
    check_tree(twin, ref, "clone of a compact tree")
    check_pairs(twin.par_scan(), ref.scan(), "scan of the clone")
    check_pairs(gpu.par_scan(), ref.scan(), "scan of the original")
    warm(gpu, kind, sysb, base, ids, 3, first=22)
    # (e) the raw stream in the reference's emission order (the literal scan kernels)
    _, ref = frame(25)
    raw_ref, _ = ref.scan_raw()
    check_pairs(gpu.scan_raw(), raw_ref, "scan_raw straight after a compact extend")
    warm(gpu, kind, sysb, base, ids, 3, first=26)
    # (f) after the compact SORT the id and box arrays exist only as context words until someone asks: the literal scan kernels,
    # a clone and the tree accessors ask after the frame's own scan
    _, ref = frame(29)
    check_pairs(gpu.par_scan(), ref.scan(), "scan on the compact route")
    raw_ref, _ = ref.scan_raw()
    check_pairs(gpu.scan_raw(), raw_ref, "scan_raw after a compact sort (ids from the context words)")
    _, ref = frame(30)
    check_pairs(gpu.par_scan(), ref.scan(), "scan on the compact route")
    twin = gpu.clone()
    ref.sort()
    check_tree(twin, ref, "clone after a compact sort")
    check_tree(gpu, ref, "tree after a compact sort")
    check_pairs(twin.par_scan(), ref.scan(), "scan of the clone")


def test_a_frame_that_leaves_the_compact_format_runs_again_on_the_plain_path(monkeypatch):
    """the format is chosen from the PREVIOUS frame's depths and ids (one level of head room on the deep side where 64 bits allow):
    a frame with much smaller objects (deeper cells) or wider ids does not fit it -- extend must notice from the frame's own
    facts and run again, and the results stay the oracle's"""
    monkeypatch.setenv("BP_RADIX_CLAIM_MIN_TILES", "1")
    kind, n = 2, 200000
    sysb, base, ids = base_scene(kind, n, seed=33)
    gpu = bp.Layer(KINDS[kind], 0)
    warm(gpu, kind, sysb, base, ids, 4)
    st0 = gpu.stats()
    assert st0["tree_sorts_compact"] >= 1
    # deeper: a handful of objects shrunk by 2^-6 (six levels below anything the last frame had)
    b = moved(kind, sysb, base, 40)
    c = (b[:50, :3] + b[:50, 3:]) * np.float32(0.5)
    h = (b[:50, 3:] - b[:50, :3]) * np.float32(0.5 / 64)
    b[:50] = np.concatenate([c - h, c + h], axis=1)
    ref = oracle.OracleLayer(kind, 0)
    gpu.clear(); gpu.extend(sysb, b, ids); ref.extend(sysb, b, ids)
    check_pairs(gpu.par_scan(), ref.scan(), "frame with deeper cells")
    check_tree(gpu, ref, "frame with deeper cells, tree")
    st1 = gpu.stats()
    assert st1["extend_reruns"] == st0["extend_reruns"] + 1, (st0["extend_reruns"], st1["extend_reruns"])
    warm(gpu, kind, sysb, base, ids, 4, first=41)
    # wider ids: the same objects under ids that need three more bits (still distinct, ascending, below 2^27)
    st2 = gpu.stats()
    wide = (ids.astype(np.uint64) * 8 + 5).astype(np.uint32)
    b = moved(kind, sysb, base, 50)
    ref = oracle.OracleLayer(kind, 0)
    gpu.clear(); gpu.extend(sysb, b, wide); ref.extend(sysb, b, wide)
    check_pairs(gpu.par_scan(), ref.scan(), "frame with wider ids")
    check_tree(gpu, ref, "frame with wider ids, tree")
    assert gpu.stats()["extend_reruns"] == st2["extend_reruns"] + 1


def test_ids_above_2_27_keep_the_duplicate_suppression_on_the_compact_route(monkeypatch):
    """ids >= 2^27 have no spare bits for the box code, so through the (index, id) sort the codes are lost and the scan falls back
    to the reference's emit-every-combo + dedup (twice the raw pairs).  In a compact word the code has its own field and the id
    field holds id - min(id): once the route engages the raw stream is duplicate-free again, whatever the ids' magnitude"""
    monkeypatch.setenv("BP_RADIX_CLAIM_MIN_TILES", "1")
    kind, n = 2, 200000
    sysb, base, ids = base_scene(kind, n, seed=61)
    big = (ids.astype(np.uint64) * 3 + 3_000_000_000).astype(np.uint32)   # distinct, ascending, all >= 2^27, range 600 k
    assert big.min() >= (1 << 27)
    gpu = bp.Layer(KINDS[kind], 0)
    raw = []
    for f in range(7):
        b = moved(kind, sysb, base, f)
        ref = oracle.OracleLayer(kind, 0)
        gpu.clear()
        gpu.extend(sysb, b, big); ref.extend(sysb, b, big)
        check_pairs(gpu.par_scan(), ref.scan(), "big ids, frame %d" % f)
        check_tree(gpu, (ref.sort(), ref)[1], "big ids, frame %d, tree" % f)
        st = gpu.stats()
        raw.append((st["raw_pairs"], st["pairs"]))
    st = gpu.stats()
    assert st["tree_sorts_compact"] >= 3, {k: st[k] for k in ("tree_sorts_compact", "tree_sorts_in_cells", "extend_reruns", "tree_expands")}
    assert raw[0][0] > raw[0][1] and raw[-1][0] == raw[-1][1], raw   # first frame: every combo; on the route: each pair once
    # round 5: the pair sort's cell route numbers its cells by bits 8 .. 23 of the first id -- ids of any magnitude whose top byte is constant
    assert st["sorts_in_cells"] >= 1, {k: st[k] for k in ("sorts_in_cells", "sorts_ahead", "sorts_ahead_missed")}


def test_compact_switch_off_gives_the_same_lists():
    """BP_NO_COMPACT_TREE (read once per process: child process) takes the route out; same frames, same lists, no compact sort"""
    import os, subprocess, sys
    code = (
        "import numpy as np, oracle, broadphase_rs_amd as bp\n"
        "from broadphase_rs_amd import scenes\n"
        "sysb, base, ids = scenes.gen_boxes(200000, seed=11)\n"
        "size = base[:, 3:] - base[:, :3]; side = np.float32(sysb[1][0])\n"
        "gpu = bp.Layer(bp.Index64_3D, 0)\n"
        "for f in range(6):\n"
        "    rng = np.random.default_rng(100 + f)\n"
        "    mn = np.minimum(np.maximum(base[:, :3] + (rng.random((len(ids), 3), dtype=np.float32) - np.float32(0.5)) * size, 0), side - size).astype(np.float32)\n"
        "    b = np.concatenate([mn, mn + size], axis=1).astype(np.float32)\n"
        "    ref = oracle.OracleLayer(2, 0); gpu.clear(); gpu.extend(sysb, b, ids); ref.extend(sysb, b, ids)\n"
        "    got = np.array(gpu.par_scan()); exp = ref.scan()\n"
        "    assert got.shape == exp.shape and np.array_equal(got, exp), f\n"
        "    gk, gi = gpu.tree(); ref.sort()\n"
        "    assert np.array_equal(gk, ref.keys) and np.array_equal(gi, ref.ids), f\n"
        "st = gpu.stats()\n"
        "assert st['tree_sorts_compact'] == 0 and st['tree_sorts_in_cells'] >= 1, st\n"
        "print('ok')\n")
    env = dict(os.environ, BP_RADIX_CLAIM_MIN_TILES="1", BP_NO_COMPACT_TREE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-2000:])


@pytest.mark.parametrize("seed", list(range(16)))
def test_compact_route_on_random_layers(seed, monkeypatch):
    """random index kind, size, min_depth, id base / stride and object scale; moving frames with a random extra consumer of the tree
    in some of them (unsorted iter, clone, second extend, scan_raw) -- every tree and pair list against the oracle.  Exercises the
    format's corners: no tail level (all objects of one depth), depth 0 entries, 2-D digit tails, id fields of 1 .. 27 bits"""
    monkeypatch.setenv("BP_RADIX_CLAIM_MIN_TILES", "1")
    rng = np.random.default_rng(9000 + seed)
    kind = int(rng.integers(0, 3))
    n = int(rng.integers(70000, 260000))
    min_depth = int(rng.choice([0, 0, 0, 3, 5, 6]))
    sysb, base, ids = base_scene(kind, n, seed=300 + seed)
    dim = 3 if kind == 2 else 2
    shape = int(rng.integers(0, 4))
    if shape == 1:      # every object the same size: one depth, no digit tail
        c = (base[:, :dim] + base[:, dim:]) * np.float32(0.5)
        h = np.float32(0.5 * float(np.median(base[:, dim:] - base[:, :dim])))
        base = np.concatenate([c - h, c + h], axis=1).astype(np.float32)
    elif shape == 2:    # a few objects as large as the system (depth 0 / 1 entries), the rest as generated
        k = 5
        lo = np.asarray(sysb[0], dtype=np.float32)[:dim]; hi = np.asarray(sysb[1], dtype=np.float32)[:dim]
        base[:k, :dim] = lo + (hi - lo) * np.float32(0.01)
        base[:k, dim:] = lo + (hi - lo) * np.float32(0.97)
    stride = int(rng.choice([1, 1, 3, 17]))
    id_base = int(rng.choice([0, 0, 1 << 20, (1 << 27) + 5, 3_000_000_000]))
    ids = (np.arange(n, dtype=np.uint64) * stride + id_base).astype(np.uint32)
    gpu = bp.Layer(KINDS[kind], min_depth)
    for f in range(7):
        b = moved(kind, sysb, base, f, amount=0.3) if shape != 2 else base.copy()
        if shape == 2 and f:   # (the giants stay, the rest moves)
            b[5:] = moved(kind, sysb, base[5:], f, amount=0.3)
        ref = oracle.OracleLayer(kind, min_depth)
        gpu.clear()
        gpu.extend(sysb, b, ids); ref.extend(sysb, b, ids)
        what = "seed %d kind %d n %d min_depth %d shape %d ids %d+%d*i frame %d" % (seed, kind, n, min_depth, shape, id_base, stride, f)
        op = int(rng.integers(0, 6)) if f >= 3 else 0
        if op == 1:
            check_tree(gpu, ref, what + " unsorted")
        elif op == 2:
            twin = gpu.clone()
            check_tree(twin, ref, what + " clone")
        elif op == 3:
            m = 2000
            eb = moved(kind, sysb, base[:m], 100 + f, amount=0.3)
            eids = (np.arange(m, dtype=np.uint64) * stride + id_base + n * stride).astype(np.uint32)
            gpu.extend(sysb, eb, eids); ref.extend(sysb, eb, eids)
        elif op == 4:
            raw_ref, _ = ref.scan_raw()
            check_pairs(gpu.scan_raw(), raw_ref, what + " scan_raw")
        gpu.par_sort(); ref.sort()
        check_tree(gpu, ref, what + " par_sort")
        check_pairs(gpu.par_scan(), ref.scan(), what + " par_scan")


def test_a_compact_sort_that_fails_its_deferred_check_is_repaired(monkeypatch):
    """the repair branch of a failed COMPACT sort (settle_sort with sort_pend.compact: the first pass's composite words are
    expanded in place, the full-width sort runs on plain entries) had no test.  Uniform frames choose the compact format; then a third
    of the objects moves into one corner between two frames, cells laid out from the previous sort's group sizes overflow there
    (KERR_SEGMENT_TOO_LONG in the sort's scalars), and the scan that ran behind the unfinished sort -- on depth / range-length words
    the finish never wrote for the groups it gave up -- is void: the deferred check repairs the tree and scans again.  Tree and pairs
    equal the oracle's in every frame."""
    monkeypatch.setenv("BP_RADIX_CLAIM_MIN_TILES", "1")
    n = 300000
    sysb, bounds, ids = scenes.gen_boxes(n, seed=91)
    size = bounds[:, 3:] - bounds[:, :3]
    side = np.float32(sysb[1][0])
    rng = np.random.default_rng(92)
    gpu = bp.Layer(bp.Index64_3D, 0)
    before = None
    for frame in range(7):
        mn = bounds[:, :3].copy()
        if frame in (3, 4):   # a third of the objects into the lower corner (an eighth of the volume): groups there grow ~ 3.5 x
            pick = rng.random(n) < 0.33
            mn[pick] *= np.float32(0.5)
        mn = np.minimum(np.maximum(mn + (rng.random((n, 3), dtype=np.float32) - np.float32(0.5)), np.float32(0)), side - size).astype(np.float32)
        cur = np.ascontiguousarray(np.concatenate([mn, mn + size], axis=1).astype(np.float32))
        if frame == 3:
            before = gpu.stats()
        ref = oracle.OracleLayer(2, 0)
        gpu.clear()
        gpu.extend(sysb, cur, ids); ref.extend(sysb, cur, ids)
        gpu.par_sort(); ref.par_sort()
        # (the scan first: it is the consumer that ran behind the sort whose check is still pending)
        check_pairs(gpu.scan_device().to_numpy(), ref.par_scan(), "bunching scene, frame %d, par_scan" % frame)
        check_tree(gpu, ref, "bunching scene, frame %d, par_sort" % frame)
    st = gpu.stats()
    keys = ("tree_sorts_compact", "tree_sorts_in_cells", "sort_repairs", "tree_expands", "extend_reruns")
    assert before["tree_sorts_compact"] >= 1 and before["sort_repairs"] == 0, {k: before[k] for k in keys}
    # the bunched frame's sort was a compact one (the extend before it wrote composite words) and its check failed
    assert st["sort_repairs"] >= before["sort_repairs"] + 1 and st["tree_expands"] >= before["tree_expands"] + 1, ({k: before[k] for k in keys}, {k: st[k] for k in keys})


def test_a_frame_loop_without_scans_still_learns_its_sort_plan(monkeypatch):
    """clear + extend + par_sort, frame after frame, with nothing that reads the tree's order in between: the deferred check of a sort
    whose tree is dropped is not needed -- but what it says about the next sort's plan is, or the loop stays on a layer's first plan
    for ever (`clear` fetches the dropped sort's scalars).  The trees must equal the oracle's whichever plan sorted them, and a frame
    whose groups outgrow the LDS finish (a tight cluster) must switch the grouped plan off again for the frames behind it."""
    monkeypatch.setenv("BP_RADIX_CLAIM_MIN_TILES", "1")
    kind, n = 2, 200000
    sysb, base, ids = base_scene(kind, n, seed=11)
    gpu = bp.Layer(KINDS[kind], 0)
    for f in range(5):
        b = moved(kind, sysb, base, f)
        gpu.clear(); gpu.extend(sysb, b, ids); gpu.par_sort()
    st = gpu.stats()
    assert st["tree_sorts_compact"] >= 2 and st["tree_sorts_in_cells"] >= 2, {k: st[k] for k in ("tree_sorts_compact", "tree_sorts_in_cells", "tree_plan_flips")}
    ref = oracle.OracleLayer(kind, 0)
    ref.extend(sysb, moved(kind, sysb, base, 4), ids); ref.sort()
    check_tree(gpu, ref, "the last frame of a loop without scans")
    # a frame that does not fit the grouped finish: dropped unseen, learned from all the same
    L = float(sysb[1][0])
    tight = base.copy()
    k = n // 2
    lo = (np.float32(0.4) * L + np.random.default_rng(5).uniform(0, 0.0008 * L, size=(k, 3))).astype(np.float32)
    tight[:k] = np.concatenate([lo, lo + (base[:k, 3:] - base[:k, :3]) * np.float32(0.02)], axis=1).astype(np.float32)
    flips = gpu.stats()["tree_plan_flips"]
    gpu.clear(); gpu.extend(sysb, tight, ids); gpu.par_sort()
    gpu.clear(); gpu.extend(sysb, tight, ids); gpu.par_sort()
    ref2 = oracle.OracleLayer(kind, 0)
    ref2.extend(sysb, tight, ids); ref2.sort()
    check_tree(gpu, ref2, "a clustered frame behind one whose failed sort was dropped by clear")
    assert gpu.stats()["tree_plan_flips"] > flips
