"""Synthetic workloads for the Layer extend -> sort -> scan path (numpy only, deterministic).

`gen_boxes` restates the reference's own generator recipe (utils/src/gen_test_data.rs:98-155 with the
parameters of tests/gen_test_scenes.py:12-14): density 0.001 boxes/unit^3, per-axis size U[1,10),
side L = cbrt(N/density) + avg_size, system bounds [0,L]^3, min ~ U[0, L-size), max = min+size,
ids 0..N-1.  The reference draws from ChaCha20 (rand 0.7 / rand_chacha 0.2), which cannot be
reproduced without a Rust toolchain; this module uses numpy's PCG64 with a fixed seed instead, so
scenes are statistically -- not bitwise -- the reference's.
"""
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
