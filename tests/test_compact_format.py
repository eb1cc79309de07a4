"""The compact tree record (TreeCompact, csrc/bp_kernels.h) restated in numpy: word = (prefix << tbits | tail) << low | id << 5 | code
with the levels below the shallowest depth present as digits 1 + octant in base 2^dim + 1.  Checked on the oracle's entries
(CPU only): the word order is the (index, id) order of src/layer.rs:149, and word -> (index, id, code) is exact.  The GPU twin
(tests/test_gpu_compact.py) checks the kernels that implement this format against the oracle end to end."""
import numpy as np
import pytest

import oracle
from broadphase_rs_amd import scenes

KIND = {0: (2, 4, 14), 1: (2, 5, 29), 2: (3, 5, 19)}   # dim, depth bits, axis bits (src/index.rs:293-295)


def fmt(kind, dmin, dmax, idbits):
    dim, db, ab = KIND[kind]
    T = dmax - dmin
    span = (2 ** dim + 1) ** T
    tbits = max(0, (span - 1).bit_length())
    total = dim * dmin + tbits + idbits + 5
    return dict(pshift=db + dim * (ab - dmin), tbits=tbits, T=T, low=idbits + 5, idbits=idbits, dmin=dmin, dmax=dmax, total=total, dim=dim, db=db)


def pack(f, key, ids, code):
    key = key.astype(np.uint64)
    dim, base = f["dim"], 2 ** f["dim"] + 1
    depth = (key & np.uint64((1 << f["db"]) - 1)).astype(np.int64)
    nlev = depth - f["dmin"]
    tail = np.zeros(len(key), dtype=np.uint64)
    for k in range(1, f["T"] + 1):
        o = (key >> np.uint64(f["pshift"] - dim * k)) & np.uint64((1 << dim) - 1)
        tail = tail * np.uint64(base) + np.where(k <= nlev, o + np.uint64(1), np.uint64(0))
    prefix = key >> np.uint64(f["pshift"])
    return (((prefix << np.uint64(f["tbits"])) | tail) << np.uint64(f["low"])) | (ids.astype(np.uint64) << np.uint64(5)) | code.astype(np.uint64)


def unpack(f, w):
    dim, base = f["dim"], 2 ** f["dim"] + 1
    kp = w >> np.uint64(f["low"])
    t = kp & np.uint64((1 << f["tbits"]) - 1)
    mort = np.zeros(len(w), dtype=np.uint64)
    nlev = np.zeros(len(w), dtype=np.uint64)
    for k in range(f["T"]):
        d = t % np.uint64(base)
        t = t // np.uint64(base)
        mort |= np.where(d > 0, d - np.uint64(1), np.uint64(0)) << np.uint64(dim * k)
        nlev += (d > 0).astype(np.uint64)
    key = ((kp >> np.uint64(f["tbits"])) << np.uint64(f["pshift"])) | (mort << np.uint64(f["pshift"] - dim * f["T"])) | (np.uint64(f["dmin"]) + nlev)
    return key, ((w >> np.uint64(5)) & np.uint64((1 << f["idbits"]) - 1)).astype(np.uint32), (w & np.uint64(31)).astype(np.uint32)


@pytest.mark.parametrize("kind,n,min_depth", [(2, 20000, 0), (2, 3000, 6), (1, 20000, 0), (0, 20000, 4)])
def test_word_order_is_index_id_order_and_round_trips(kind, n, min_depth):
    sysb, b, ids = scenes.gen_boxes(n, seed=5) if kind == 2 else scenes.gen_circles_2d(n, seed=5)
    ref = oracle.OracleLayer(kind, min_depth)
    ref.extend(sysb, b, ids)
    keys, eids = ref.keys.copy(), ref.ids.copy()
    depth = (keys & np.uint64((1 << KIND[kind][1]) - 1)).astype(np.int64)
    rng = np.random.default_rng(1)
    code = rng.integers(0, 32, len(keys)).astype(np.uint32)
    f = fmt(kind, int(depth.min()), int(depth.max()), max(1, int(eids.max()).bit_length()))
    assert f["total"] <= 64, f
    w = pack(f, keys, eids, code)
    k2, i2, c2 = unpack(f, w)
    assert np.array_equal(k2, keys) and np.array_equal(i2, eids) and np.array_equal(c2, code)
    by_word = np.argsort(w, kind="stable")
    by_entry = np.lexsort((eids, keys))
    assert np.array_equal(keys[by_word], keys[by_entry]) and np.array_equal(eids[by_word], eids[by_entry])
    # head room: a format one level wider on either side orders and round-trips the same entries
    f2 = fmt(kind, max(0, int(depth.min()) - 1), min(KIND[kind][2], int(depth.max()) + 1), f["idbits"])
    if f2["total"] <= 64:
        w2 = pack(f2, keys, eids, code)
        assert np.array_equal(np.argsort(w2, kind="stable"), by_word)
        k3, i3, c3 = unpack(f2, w2)
        assert np.array_equal(k3, keys) and np.array_equal(i3, eids) and np.array_equal(c3, code)


def test_the_headline_scene_fits_64_bits_only_with_the_digit_tail():
    """10 M objects: depths 7 .. 11 (16 entries at depth 11), ids below 2^24: Morton bits to level 11 + a depth field take
    33 + 3 + 24 + 5 = 65 bits; prefix to level 6 or 7 + base-9 digits take 63"""
    assert fmt(2, 7, 11, 24)["total"] == 63 and fmt(2, 6, 11, 24)["total"] == 63
    assert 3 * 11 + 3 + 24 + 5 == 65


def test_digit_shortcut_of_the_emit_pass():
    """encode_emit_compact_kernel counts a radix digit ONCE per object where the object's first and far-corner cell agree on it
    (csrc/bp_kernels.hip): correct because the per-axis key deltas live in disjoint Morton lanes -- every cell is the first cell XOR a
    subset of the deltas, so a digit the far corner shares is shared by all.  Cross-check on random 2 x 2 x 2 boxes that straddle digit
    boundaries: per-object counts (cell count added to the first cell's digit when the two ends agree, else per entry) == per-entry counts."""
    rng = np.random.default_rng(7)
    dim, db, ab = KIND[2]
    for trial in range(300):
        depth = int(rng.integers(2, 12))
        # first cell coordinates chosen near power-of-two boundaries so that +1 carries across many bits (and digits)
        c = []
        for a in range(3):
            hi_bit = int(rng.integers(1, depth))
            base = (int(rng.integers(0, 1 << (depth - hi_bit))) << hi_bit) | ((1 << hi_bit) - 1)
            c.append(min(base, (1 << depth) - 2))
        two = [int(rng.integers(0, 2)) for _ in range(3)]
        def key_of(x, y, z):
            m = 0
            for k in range(depth):
                m |= (((x >> k) & 1) << (3 * k)) | (((y >> k) & 1) << (3 * k + 1)) | (((z >> k) & 1) << (3 * k + 2))
            return ((m << (3 * (ab - depth))) << db) | depth
        cells = [key_of(c[0] + dx, c[1] + dy, c[2] + dz) for dz in range(two[2] + 1) for dy in range(two[1] + 1) for dx in range(two[0] + 1)]
        first, far = cells[0], cells[-1]
        for shift in (8, 16, 24, 32, 40, 48, 56):
            per_entry = np.bincount([(k >> shift) & 255 for k in cells], minlength=256)
            per_object = np.zeros(256, dtype=np.int64)
            if ((first ^ far) >> shift) & 255 == 0:
                per_object[(first >> shift) & 255] += len(cells)
            else:
                per_object += per_entry
            assert np.array_equal(per_entry, per_object), (trial, shift, [hex(k) for k in cells])
