"""Synthetic workloads for the Layer extend -> sort -> scan path (numpy only, deterministic).

`gen_boxes` restates the reference's own generator recipe (utils/src/gen_test_data.rs:98-155 with the
parameters of tests/gen_test_scenes.py:12-14): density 0.001 boxes/unit^3, per-axis size U[1,10),
side L = cbrt(N/density) + avg_size, system bounds [0,L]^3, min ~ U[0, L-size), max = min+size,
ids 0..N-1.  `gen_boxes` draws from numpy's PCG64 (fast at 10^7 .. 10^8 objects): statistically -- not bitwise --
the reference's scenes.  `gen_boxes_reference` is the BIT-EXACT generator: the reference draws from ChaCha20
(rand 0.7 / rand_chacha 0.2 seeded through rand_core 0.5 `seed_from_u64`), restated here from the crates' published
algorithms; its output files hash to the Git-LFS oids of the reference's tests/data/inputs (tests/test_reference_goldens.py,
tools/gen_validation_data.py --gen-inputs).
"""
import struct

import numpy as np


def gen_boxes(count, seed=0, density=0.001, size_range=(1.0, 10.0)):
    """-> (system_bounds (2,3) f32, bounds (N,6) f32 [min xyz, max xyz], ids (N,) u32)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    lo, hi = np.float32(size_range[0]), np.float32(size_range[1])
    avg = (lo + hi) / np.float32(2)
    linear = np.float32(np.cbrt(np.float32(count) / np.float32(density))) + avg
    size = lo + (hi - lo) * rng.random((count, 3), dtype=np.float32)
    mn = (linear - size) * rng.random((count, 3), dtype=np.float32)
    mx = mn + size
    bounds = np.concatenate([mn, mx], axis=1).astype(np.float32)
    sysb = np.array([[0, 0, 0], [linear, linear, linear]], dtype=np.float32)
    return sysb, bounds, np.arange(count, dtype=np.uint32)


def gen_boxes_shard(count_global, rank, world, seed=0, density=0.001, size_range=(1.0, 10.0)):
    """Rank `rank`'s share of one global gen_boxes scene of count_global objects: the same system
    bounds on every rank (side from the GLOBAL count), objects drawn independently per rank, ids
    [rank * n, (rank + 1) * n).  Statistically the global scene sharded by object index."""
    n = count_global // world
    rng = np.random.Generator(np.random.PCG64([seed, rank, world]))
    lo, hi = np.float32(size_range[0]), np.float32(size_range[1])
    avg = (lo + hi) / np.float32(2)
    linear = np.float32(np.cbrt(np.float32(count_global) / np.float32(density))) + avg
    size = lo + (hi - lo) * rng.random((n, 3), dtype=np.float32)
    mn = (linear - size) * rng.random((n, 3), dtype=np.float32)
    mx = mn + size
    bounds = np.concatenate([mn, mx], axis=1).astype(np.float32)
    sysb = np.array([[0, 0, 0], [linear, linear, linear]], dtype=np.float32)
    return sysb, bounds, (np.arange(n, dtype=np.uint32) + np.uint32(rank * n))


def gen_boxes_powerlaw(count, seed=0, density=0.001, alpha=2.5, cap_frac=0.25):
    """Mixed-scale scene (BASELINE.json configs[4]): same placement as gen_boxes, isotropic size
    ~ Pareto(x_m=1, alpha) capped at cap_frac * L (deep-stack stress)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    linear = np.float32(np.cbrt(np.float32(count) / np.float32(density))) + np.float32(5.5)
    u = rng.random(count, dtype=np.float32)
    size = np.minimum((1.0 / np.maximum(1.0 - u, 1e-7)) ** (1.0 / alpha), cap_frac * linear).astype(np.float32)
    size3 = np.repeat(size[:, None], 3, axis=1)
    mn = (linear - size3) * rng.random((count, 3), dtype=np.float32)
    mx = mn + size3
    bounds = np.concatenate([mn, mx], axis=1).astype(np.float32)
    sysb = np.array([[0, 0, 0], [linear, linear, linear]], dtype=np.float32)
    return sysb, bounds, np.arange(count, dtype=np.uint32)


def gen_boxes_powerlaw_shard(count_global, rank, world, seed=0, density=0.001, alpha=2.5, cap_frac=0.25):
    """Rank `rank`'s share of one global mixed-scale scene (configs[4]: power-law sizes, sharded by object index):
    system bounds from the GLOBAL count, objects drawn independently per rank, ids [rank * n, (rank + 1) * n)."""
    n = count_global // world
    rng = np.random.Generator(np.random.PCG64([seed, rank, world, 4]))
    linear = np.float32(np.cbrt(np.float32(count_global) / np.float32(density))) + np.float32(5.5)
    u = rng.random(n, dtype=np.float32)
    size = np.minimum((1.0 / np.maximum(1.0 - u, 1e-7)) ** (1.0 / alpha), cap_frac * linear).astype(np.float32)
    size3 = np.repeat(size[:, None], 3, axis=1)
    mn = (linear - size3) * rng.random((n, 3), dtype=np.float32)
    mx = mn + size3
    bounds = np.concatenate([mn, mx], axis=1).astype(np.float32)
    sysb = np.array([[0, 0, 0], [linear, linear, linear]], dtype=np.float32)
    return sysb, bounds, (np.arange(n, dtype=np.uint32) + np.uint32(rank * n))


def gen_circles_2d(count, seed=0, width=1280.0, height=720.0):
    """BASELINE.json configs[0]: the example's scene (examples/main.rs:314, 415-417, 163-169):
    r = exp(U[0.5,2.0)), AABB = centre +- r, system bounds [-1, max(w,h)]^2, Index32_2D, min_depth 4."""
    rng = np.random.Generator(np.random.PCG64(seed))
    r = np.exp(0.5 + 1.5 * rng.random(count, dtype=np.float32)).astype(np.float32)
    cx = r + (np.float32(width) - 2 * r) * rng.random(count, dtype=np.float32)
    cy = r + (np.float32(height) - 2 * r) * rng.random(count, dtype=np.float32)
    bounds = np.stack([cx - r, cy - r, cx + r, cy + r], axis=1).astype(np.float32)
    m = np.float32(max(width, height))
    sysb = np.array([[-1, -1], [m, m]], dtype=np.float32)
    return sysb, bounds, np.arange(count, dtype=np.uint32)


# ---- the reference's own generator, bit for bit (utils/src/gen_test_data.rs:98-155 as driven by tests/gen_test_scenes.py:12-33) ----
_M32 = 0xFFFFFFFF
# f32::cbrt of count / density is the one platform-dependent step (libm's cbrtf): off by this many ulps from numpy's for the scene
# sizes the reference ships (found once against the files' sha256, tests/golden/reference_lfs_oids.json); 0 for any other size
REFERENCE_CBRT_ULPS = {100: -1, 300: 0, 1000: 0, 3000: 0, 10000: 1, 30000: 0, 100000: 0}


def _seed_from_u64(state):
    """rand_core 0.5 SeedableRng::seed_from_u64: one PCG32 output per 4-byte chunk of the 32-byte seed"""
    mul, inc = 6364136223846793005, 11634580027462260723
    out = b""
    for _ in range(8):
        state = (state * mul + inc) & 0xFFFFFFFFFFFFFFFF
        xorshifted = (((state >> 18) ^ state) >> 27) & _M32
        rot = state >> 59
        out += struct.pack("<I", ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & _M32)
    return out


def _chacha20_words(key_words, n_words):
    """the first n_words u32 outputs of rand_chacha 0.2's ChaChaRng (ChaCha20, 64-bit block counter, stream 0), all blocks at once"""
    nb = (n_words + 15) // 16
    ctr = np.arange(nb, dtype=np.uint64)
    st = np.empty((16, nb), dtype=np.uint32)
    for i, c in enumerate((0x61707865, 0x3320646E, 0x79622D32, 0x6B206574)):
        st[i] = c
    for i in range(8):
        st[4 + i] = key_words[i]
    st[12] = (ctr & np.uint64(_M32)).astype(np.uint32)
    st[13] = (ctr >> np.uint64(32)).astype(np.uint32)
    st[14] = 0
    st[15] = 0
    x = st.copy()

    def rotl(v, n):
        return (v << np.uint32(n)) | (v >> np.uint32(32 - n))

    def qr(a, b, c, d):
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7)
    with np.errstate(over="ignore"):
        for _ in range(10):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        x += st
    return x.T.reshape(-1)[:n_words]


def _gen_range_f32(words, pos, low, high):
    """rand 0.7 UniformFloat<f32>::sample_single on the word stream: 23 mantissa bits -> [1, 2) - 1, * scale + low; a result that
    rounding pushed onto `high` shrinks the scale by an ulp and draws again.  -> (value, next position)"""
    f = np.float32
    low, high = f(low), f(high)
    scale = f(high - low)
    while True:
        bits = (int(words[pos]) >> 9) | 0x3F800000
        pos += 1
        v01 = f(np.frombuffer(struct.pack("<I", bits), dtype=np.float32)[0] - f(1.0))
        res = f(f(v01 * scale) + low)
        if res < high:
            return res, pos
        scale = np.nextafter(scale, f(0.0), dtype=np.float32)


def gen_boxes_reference(count, seed=0, density=0.001, size_range=(1.0, 10.0), cbrt_ulps=None):
    """The reference's gen_boxes, bit for bit -> (system_bounds (2,3) f32, bounds (N,6) f32, ids (N,) u32).
    cbrt_ulps: correction of the platform's cbrtf (None: REFERENCE_CBRT_ULPS for the sizes the reference ships, else 0)."""
    f = np.float32
    if cbrt_ulps is None:
        cbrt_ulps = REFERENCE_CBRT_ULPS.get(int(count), 0) if (seed, density, tuple(size_range)) == (0, 0.001, (1.0, 10.0)) else 0
    s0, s1 = f(size_range[0]), f(size_range[1])
    avg = f(f(s0 + s1) / f(2.0))
    lin = f(np.cbrt(f(f(count) / f(density))))
    for _ in range(abs(cbrt_ulps)):
        lin = np.nextafter(lin, f(np.inf if cbrt_ulps > 0 else -np.inf), dtype=np.float32)
    linear = f(lin + avg)
    key = struct.unpack("<8I", _seed_from_u64(seed))
    sysb = np.array([[0, 0, 0], [linear, linear, linear]], dtype=np.float32)
    ids = np.arange(count, dtype=np.uint32)
    # vectorised: valid when no draw is rejected (a rejection would shift the stream) -- checked
    w = _chacha20_words(key, 6 * count).reshape(count, 6)
    v01 = (((w >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - f(1.0)).astype(np.float32)
    size = (v01[:, :3] * f(s1 - s0) + s0).astype(np.float32)
    scale = (linear - size).astype(np.float32)                       # high - low with low = 0
    mn = (v01[:, 3:] * scale + f(0.0)).astype(np.float32)
    if np.all(size < s1) and np.all(mn < scale):
        return sysb, np.concatenate([mn, (mn + size).astype(np.float32)], axis=1).astype(np.float32), ids
    # a rejected draw somewhere: replay the stream draw by draw (more words than 6 per object may be consumed)
    words = _chacha20_words(key, 6 * count + 4096)
    pos, rows = 0, []
    for _ in range(count):
        sz = []
        for _k in range(3):
            v, pos = _gen_range_f32(words, pos, s0, s1)
            sz.append(v)
        lo3 = []
        for k in range(3):
            v, pos = _gen_range_f32(words, pos, f(0.0), f(linear - sz[k]))
            lo3.append(v)
        rows.append(lo3 + [f(lo3[k] + sz[k]) for k in range(3)])
    return sysb, np.array(rows, dtype=np.float32), ids
