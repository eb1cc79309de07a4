"""Second, independent restatement of the reference path in pure Python (small cases only).

TEST INFRASTRUCTURE ONLY (same rules as oracle/__init__.py).  Written separately from
bp_oracle.c -- naive bit loops instead of the magic-mask codec, Python ints instead of
fixed-width arithmetic, numpy.float32 scalars for the quantisation -- so that a transcription
error in either restatement shows up as a disagreement in tests/test_oracle.py.

Reference: src/index.rs (codec, :65-295), src/geom.rs (:40-61, :104-128, :148-163, :183-305),
src/layer.rs (:84-165, :449-573).
"""
import numpy as np

KINDS = {0: (2, 32, 4, 14), 1: (2, 64, 5, 29), 2: (3, 64, 5, 19)}  # index.rs:293-295


def encode_axis(kind, v):
    """Spread the top AXIS_BITS of a u32 so bit k lands at bit dim*k (index.rs:155-172,193-207)."""
    dim, _bits, _db, ab = KINDS[kind]
    a = (v & 0xFFFFFFFF) >> (32 - ab)
    out = 0
    for k in range(ab):
        out |= ((a >> k) & 1) << (dim * k)
    return out


def decode_axis(kind, o):
    dim, _bits, _db, ab = KINDS[kind]
    a = 0
    for k in range(ab):
        a |= ((o >> (dim * k)) & 1) << k
    return (a << (32 - ab)) & 0xFFFFFFFF


def level_mask(kind, depth):
    dim, _bits, db, ab = KINDS[kind]
    if depth <= 0:
        return 0
    return ((1 << (dim * depth)) - 1) << (dim * ab + db - dim * depth)


def make_index(kind, depth, origin):
    """Index::default().set_depth(depth).set_origin(origin)  (index.rs:106-112, 230-250)"""
    dim, bits, db, ab = KINDS[kind]
    d = min(depth, ab)
    morton = 0
    for axis in range(dim):
        morton |= encode_axis(kind, origin[axis]) << axis
    idx = (morton << db) | (d & ((1 << db) - 1))
    return idx & ((1 << bits) - 1)


def depth_of(kind, idx):
    return idx & ((1 << KINDS[kind][2]) - 1)


def overlaps(kind, a, b):
    d = min(depth_of(kind, a), depth_of(kind, b))
    return ((a ^ b) & level_mask(kind, d)) == 0


def f32_as_u32(f):
    f = np.float32(f)
    if not (f > 0):
        return 0
    if f >= np.float32(4294967296.0):
        return 0xFFFFFFFF
    return int(f)


def to_local(sys_min, sys_max, g):
    """geom.rs:148-163, evaluated with IEEE binary32 scalars"""
    out = []
    rng = np.float32(0xFFFFFF00)
    with np.errstate(all="ignore"):
        for i in range(len(g)):
            size = np.float32(sys_max[i]) - np.float32(sys_min[i])
            v = (np.float32(g[i]) - np.float32(sys_min[i])) / size * rng + np.float32(0.0)
            out.append(f32_as_u32(v))
    return out


def contains(sys_min, sys_max, bmin, bmax):
    for i in range(len(bmin)):
        if np.float32(sys_min[i]) > np.float32(bmin[i]) or np.float32(sys_max[i]) < np.float32(bmax[i]):
            return False
    return True


def indices(kind, lmin, lmax, min_depth):
    """geom.rs:189-238 / 247-304 -> list of indices in reference emission order"""
    dim, _bits, _db, ab = KINDS[kind]
    max_axis = 0
    for i in range(dim):
        max_axis = max(max_axis, ((lmax[i] - lmin[i]) + 1) & 0xFFFFFFFF)
    ext = (max_axis - 1) & 0xFFFFFFFF
    depth = 32 - ext.bit_length()  # leading_zeros
    if min_depth is not None and depth < min_depth:
        depth = min_depth
    depth = min(depth, ab)
    if depth == 0:
        return [0]
    step = 1 << (32 - depth)
    lo = [v & ~(step - 1) & 0xFFFFFFFF for v in lmin]
    hi = [v & ~(step - 1) & 0xFFFFFFFF for v in lmax]

    def axis_range(a):
        vals, x = [], lo[a]
        while True:
            vals.append(x)
            if x >= hi[a]:
                break
            x += step
        return vals

    out = []
    if dim == 3:
        for z in axis_range(2):
            for y in axis_range(1):
                for x in axis_range(0):
                    out.append(make_index(kind, depth, (x, y, z)))
    else:
        for y in axis_range(1):
            for x in axis_range(0):
                out.append(make_index(kind, depth, (x, y)))
    return out


class PyLayer:
    def __init__(self, kind=2, min_depth=0):
        self.kind, self.min_depth = kind, min_depth
        self.tree, self.sorted = [], True

    def clear(self):
        self.tree, self.sorted = [], True

    def extend(self, system_bounds, bounds, ids):
        dim = KINDS[self.kind][0]
        smin, smax = system_bounds
        for row, id_ in zip(np.asarray(bounds, dtype=np.float32).reshape(-1, 2 * dim), ids):
            bmin, bmax = row[:dim], row[dim:]
            if not contains(smin, smax, bmin, bmax):
                continue
            lmin, lmax = to_local(smin, smax, bmin), to_local(smin, smax, bmax)
            for idx in indices(self.kind, lmin, lmax, self.min_depth):
                self.tree.append((idx, int(id_)))
            self.sorted = False

    def merge(self, other):
        self.min_depth = min(self.min_depth, other.min_depth)
        self.tree.extend(other.tree)
        self.sorted = False

    def sort(self):
        if not self.sorted:
            self.tree.sort()
            self.sorted = True

    def scan_raw(self, filt=None):
        """scan_impl (layer.rs:550-573) over the whole sorted tree; raw emission order"""
        self.sort()
        stack, out, max_stack = [], [], 0
        for index, id_ in self.tree:
            while stack and not overlaps(self.kind, index, stack[-1][0]):
                stack.pop()
            if any(sid == id_ for _, sid in stack):
                continue
            for _, sid in stack:
                if id_ != sid and (filt is None or filt(id_, sid)):
                    out.append((id_, sid))
            stack.append((index, id_))
            max_stack = max(max_stack, len(stack))
        return out, max_stack

    def scan(self, filt=None):
        raw, _ = self.scan_raw(filt)
        return sorted(set(raw))

    def test_box(self, system_bounds, test_bounds, max_depth=None):
        """Layer::test_box (layer.rs:294-313, test_impl :167-239) in CLOSED FORM, unlike the recursive
        restatement in bp_oracle.c: the recursion reaches an entry iff the float bounds of every cell on the
        path from the root down to level min(depth, max_depth) overlap the test box (should_test,
        geom.rs:409-412); those bounds come from repeated centre splits (geom.rs:383-407 / 418-446) with
        cgmath's midpoint = min + (max - min) / 2 in f32."""
        self.sort()
        dim, _bits, db, ab = KINDS[self.kind]
        f = np.float32
        smin = [f(v) for v in system_bounds[0]]
        smax = [f(v) for v in system_bounds[1]]
        tmin = [f(v) for v in test_bounds[0]]
        tmax = [f(v) for v in test_bounds[1]]

        def ov(mn, mx):
            return not any(mn[a] > tmax[a] or mx[a] < tmin[a] for a in range(dim))

        out = set()
        for index, id_ in self.tree:
            d = depth_of(self.kind, index)
            levels = d if max_depth is None else min(d, max_depth)
            morton = index >> db
            mn, mx = list(smin), list(smax)
            ok = ov(mn, mx)
            for lvl in range(1, levels + 1):
                if not ok:
                    break
                child = (morton >> (dim * (ab - lvl))) & ((1 << dim) - 1)
                for a in range(dim):
                    c = f(mn[a] + f(f(mx[a] - mn[a]) / f(2.0)))
                    if child & (1 << a):
                        mn[a] = c
                    else:
                        mx[a] = c
                ok = ov(mn, mx)
            if ok:
                out.add(id_)
        return sorted(out)

    def test_ray(self, system_bounds, origin, direction, range_min=-np.inf, range_max=np.inf, max_depth=None):
        """Layer::test_ray (layer.rs:327-351) in closed form: replay the path of every entry with
        RayTestGeometry::with_system_bounds / subdivide / should_test (geom.rs:511-537, 617-661, 686-688)."""
        self.sort()
        dim, _bits, db, ab = KINDS[self.kind]
        f = np.float32
        smin = [f(v) for v in system_bounds[0]]
        smax = [f(v) for v in system_bounds[1]]
        o = [f(v) for v in origin]
        dr = [f(v) for v in direction]
        inf = f(np.inf)

        def fmax(a, b):  # f32::max: a NaN operand loses
            return b if np.isnan(a) else (a if np.isnan(b) else max(a, b))

        def fmin(a, b):
            return b if np.isnan(a) else (a if np.isnan(b) else min(a, b))

        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            r0, r1 = f(range_min), f(range_max)
            for a in range(dim):
                d0 = f(f(smin[a] - o[a]) / dr[a])
                d1 = f(f(smax[a] - o[a]) / dr[a])
                if not dr[a] > 0:
                    d0, d1 = d1, d0
                if np.isfinite(d0):
                    r0 = fmax(r0, d0)
                if np.isfinite(d1):
                    r1 = fmin(r1, d1)
            out = set()
            for index, id_ in self.tree:
                d = depth_of(self.kind, index)
                levels = d if max_depth is None else min(d, max_depth)
                morton = index >> db
                mn, mx, lo, hi = list(smin), list(smax), r0, r1
                ok = bool(lo < hi and lo < inf)
                for lvl in range(1, levels + 1):
                    if not ok:
                        break
                    child = (morton >> (dim * (ab - lvl))) & ((1 << dim) - 1)
                    ctr = [f(mn[a] + f(f(mx[a] - mn[a]) / f(2.0))) for a in range(dim)]
                    for a in range(dim):
                        side = bool(child & (1 << a))
                        dist = f(f(ctr[a] - o[a]) / dr[a])
                        if np.isfinite(dist):
                            if bool(dr[a] > 0) != side:
                                hi = fmin(hi, dist)
                            else:
                                lo = fmax(lo, dist)
                        elif bool(o[a] > ctr[a]) != side:
                            lo, hi = inf, -inf
                    for a in range(dim):
                        if child & (1 << a):
                            mn[a] = ctr[a]
                        else:
                            mx[a] = ctr[a]
                    ok = bool(lo < hi and lo < inf)
                if ok:
                    out.add(id_)
        return sorted(out)
