"""ctypes binding of the CPU oracle (oracle/bp_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package (broadphase_rs_amd), which fails
loudly when its HIP library is missing instead of falling back to this.

Parity pinning status: see bp_oracle.h -- PINNED end to end: run on the reference's own 10 000-object test
scene (regenerated bit-exactly, authenticated by the Git-LFS sha256) it reproduces the reference's three
validation goldens byte for byte (tests/test_reference_goldens.py); the codec/quantisation KATs are pinned too.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BP_ORACLE_ASAN=1 (with LD_PRELOAD=libasan.so): the sanitizer build of the same source, for `make -C oracle _build/liboracle_asan.so`
_ASAN = os.environ.get("BP_ORACLE_ASAN") == "1"
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle_asan.so" if _ASAN else "liboracle.so")

INDEX32_2D, INDEX64_2D, INDEX64_3D = 0, 1, 2
KINDS = {  # src/index.rs:293-295: (dim, bits, depth_bits, axis_bits)
    INDEX32_2D: (2, 32, 4, 14),
    INDEX64_2D: (2, 64, 5, 29),
    INDEX64_3D: (3, 64, 5, 19),
}


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("bp_oracle.c", "bp_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "_build/liboracle_asan.so" if _ASAN else "all"])
    return _LIB_PATH


_lib = None

FILTER_FN = C.CFUNCTYPE(C.c_int, C.c_uint32, C.c_uint32, C.c_void_p)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    u32p, u64p, f32p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_float)
    sz, szp = C.c_size_t, C.POINTER(C.c_size_t)
    L.orc_encode_axis.restype = C.c_uint64
    L.orc_encode_axis.argtypes = [C.c_int, C.c_uint32]
    L.orc_decode_axis.restype = C.c_uint32
    L.orc_decode_axis.argtypes = [C.c_int, C.c_uint64]
    L.orc_level_mask.restype = C.c_uint64
    L.orc_level_mask.argtypes = [C.c_int, C.c_uint32]
    L.orc_clamp_depth.restype = C.c_uint32
    L.orc_clamp_depth.argtypes = [C.c_int, C.c_uint32]
    L.orc_depth.restype = C.c_uint32
    L.orc_depth.argtypes = [C.c_int, C.c_uint64]
    L.orc_set_depth.restype = C.c_uint64
    L.orc_set_depth.argtypes = [C.c_int, C.c_uint64, C.c_uint32]
    L.orc_set_origin.restype = C.c_uint64
    L.orc_set_origin.argtypes = [C.c_int, C.c_uint64, u32p]
    L.orc_origin.restype = None
    L.orc_origin.argtypes = [C.c_int, C.c_uint64, u32p]
    L.orc_same_cell_at_depth.restype = C.c_int
    L.orc_same_cell_at_depth.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint32]
    L.orc_overlaps.restype = C.c_int
    L.orc_overlaps.argtypes = [C.c_int, C.c_uint64, C.c_uint64]
    L.orc_contains.restype = C.c_int
    L.orc_contains.argtypes = [C.c_uint32, f32p, f32p, f32p, f32p]
    L.orc_to_local.restype = None
    L.orc_to_local.argtypes = [C.c_uint32, f32p, f32p, f32p, u32p]
    L.orc_to_global.restype = None
    L.orc_to_global.argtypes = [C.c_uint32, f32p, f32p, u32p, f32p]
    L.orc_indices_depth.restype = C.c_uint32
    L.orc_indices_depth.argtypes = [C.c_int, u32p, u32p, C.c_int, C.c_uint32]
    L.orc_indices_at_depth.restype = C.c_uint64
    L.orc_indices_at_depth.argtypes = [C.c_int, u32p, u32p, C.c_uint32, u64p, C.c_uint64]
    L.orc_layer_new.restype = C.c_void_p
    L.orc_layer_new.argtypes = [C.c_int, C.c_uint32]
    L.orc_layer_free.restype = None
    L.orc_layer_free.argtypes = [C.c_void_p]
    L.orc_layer_clear.restype = None
    L.orc_layer_clear.argtypes = [C.c_void_p]
    L.orc_layer_extend.restype = C.c_int
    L.orc_layer_extend.argtypes = [C.c_void_p, f32p, f32p, f32p, u32p, sz]
    L.orc_layer_merge.restype = C.c_int
    L.orc_layer_merge.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_layer_append_entries.restype = C.c_int
    L.orc_layer_append_entries.argtypes = [C.c_void_p, u64p, u32p, sz]
    L.orc_layer_sort.restype = None
    L.orc_layer_sort.argtypes = [C.c_void_p]
    L.orc_layer_par_sort.restype = None
    L.orc_layer_par_sort.argtypes = [C.c_void_p, C.c_int]
    L.orc_layer_scan.restype = u32p
    L.orc_layer_scan.argtypes = [C.c_void_p, szp]
    L.orc_layer_scan_filtered.restype = u32p
    L.orc_layer_scan_filtered.argtypes = [C.c_void_p, FILTER_FN, C.c_void_p, szp]
    L.orc_layer_par_scan.restype = u32p
    L.orc_layer_par_scan.argtypes = [C.c_void_p, C.c_int, szp]
    L.orc_layer_par_scan_filtered.restype = u32p
    L.orc_layer_par_scan_filtered.argtypes = [C.c_void_p, C.c_int, FILTER_FN, C.c_void_p, szp]
    L.orc_layer_scan_raw.restype = u32p
    L.orc_layer_scan_raw.argtypes = [C.c_void_p, szp, szp]
    L.orc_layer_len.restype = sz
    L.orc_layer_len.argtypes = [C.c_void_p]
    L.orc_layer_keys.restype = u64p
    L.orc_layer_keys.argtypes = [C.c_void_p]
    L.orc_layer_ids.restype = u32p
    L.orc_layer_ids.argtypes = [C.c_void_p]
    L.orc_layer_sorted.restype = C.c_int
    L.orc_layer_sorted.argtypes = [C.c_void_p]
    L.orc_layer_min_depth.restype = C.c_uint32
    L.orc_layer_min_depth.argtypes = [C.c_void_p]
    L.orc_layer_num_invalid.restype = sz
    L.orc_layer_num_invalid.argtypes = [C.c_void_p]
    L.orc_layer_test_box.restype = C.c_long
    L.orc_layer_test_box.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p, C.c_int, C.c_uint32, u32p, sz]
    L.orc_layer_test_ray.restype = C.c_long
    L.orc_layer_test_ray.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p, C.c_float, C.c_float, C.c_int, C.c_uint32, u32p, sz]
    L.orc_layer_pick_ray_aabb.restype = C.c_int
    L.orc_layer_pick_ray_aabb.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p, C.c_float, C.c_int, C.c_uint32, f32p, sz,
                                          C.POINTER(C.c_float), u32p]
    L.orc_brute_force_overlaps.restype = sz
    L.orc_brute_force_overlaps.argtypes = [C.c_uint32, f32p, sz, u32p, sz]
    L.orc_hw_threads.restype = C.c_int
    L.orc_hw_threads.argtypes = []
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def to_local(sys_min, sys_max, g):
    sys_min, sys_max, g = _f32(sys_min), _f32(sys_max), _f32(g)
    out = np.zeros(len(g), dtype=np.uint32)
    lib().orc_to_local(len(g), _ptr(sys_min, C.c_float), _ptr(sys_max, C.c_float), _ptr(g, C.c_float),
                       _ptr(out, C.c_uint32))
    return out


def to_global(sys_min, sys_max, l):
    sys_min, sys_max = _f32(sys_min), _f32(sys_max)
    l = np.ascontiguousarray(l, dtype=np.uint32)
    out = np.zeros(len(l), dtype=np.float32)
    lib().orc_to_global(len(l), _ptr(sys_min, C.c_float), _ptr(sys_max, C.c_float), _ptr(l, C.c_uint32),
                        _ptr(out, C.c_float))
    return out


def indices(kind, lmin, lmax, min_depth=None):
    """Bounds<Point<u32>>::indices(min_depth) -> (depth, np.uint64[])  geom.rs:189-200/247-258"""
    lmin = np.ascontiguousarray(lmin, dtype=np.uint32)
    lmax = np.ascontiguousarray(lmax, dtype=np.uint32)
    L = lib()
    d = L.orc_indices_depth(kind, _ptr(lmin, C.c_uint32), _ptr(lmax, C.c_uint32),
                            0 if min_depth is None else 1, 0 if min_depth is None else int(min_depth))
    n = L.orc_indices_at_depth(kind, _ptr(lmin, C.c_uint32), _ptr(lmax, C.c_uint32), d, None, 0)
    out = np.zeros(int(n), dtype=np.uint64)
    L.orc_indices_at_depth(kind, _ptr(lmin, C.c_uint32), _ptr(lmax, C.c_uint32), d, _ptr(out, C.c_uint64), n)
    return int(d), out


def brute_force_overlaps(bounds):
    bounds = _f32(bounds)
    n, w = bounds.shape
    dim = w // 2
    L = lib()
    cnt = L.orc_brute_force_overlaps(dim, _ptr(bounds, C.c_float), n, None, 0)
    out = np.zeros((int(cnt), 2), dtype=np.uint32)
    L.orc_brute_force_overlaps(dim, _ptr(bounds, C.c_float), n, _ptr(out, C.c_uint32), cnt)
    return out


class OracleLayer:
    """Mirror of the reference Layer<Index, u32> (src/layer.rs:42-68) over the C oracle."""

    def __init__(self, kind=INDEX64_3D, min_depth=0):
        self.kind = kind
        self._L = lib()
        self._h = self._L.orc_layer_new(kind, min_depth)
        if not self._h:
            raise MemoryError("orc_layer_new failed")
        self._cb_keepalive = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_layer_free(self._h)
            self._h = None

    @property
    def dim(self):
        return KINDS[self.kind][0]

    def clear(self):
        self._L.orc_layer_clear(self._h)

    def extend(self, system_bounds, bounds, ids):
        sys_min, sys_max = _f32(system_bounds[0]), _f32(system_bounds[1])
        bounds = _f32(bounds).reshape(-1, 2 * self.dim)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        assert len(ids) == len(bounds)
        rc = self._L.orc_layer_extend(self._h, _ptr(sys_min, C.c_float), _ptr(sys_max, C.c_float),
                                      _ptr(bounds, C.c_float), _ptr(ids, C.c_uint32), len(ids))
        if rc:
            raise MemoryError("orc_layer_extend failed")

    def merge(self, other):
        if self._L.orc_layer_merge(self._h, other._h):
            raise MemoryError("orc_layer_merge failed")

    def append_entries(self, keys, ids):
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        if self._L.orc_layer_append_entries(self._h, _ptr(keys, C.c_uint64), _ptr(ids, C.c_uint32), len(keys)):
            raise MemoryError("orc_layer_append_entries failed")

    def sort(self):
        self._L.orc_layer_sort(self._h)

    def par_sort(self, threads=None):
        self._L.orc_layer_par_sort(self._h, threads or self._L.orc_hw_threads())

    def _pairs(self, p, n):
        n = n.value
        if n == 0:
            return np.zeros((0, 2), dtype=np.uint32)
        return np.ctypeslib.as_array(p, shape=(n, 2)).copy()

    def scan(self):
        n = C.c_size_t(0)
        return self._pairs(self._L.orc_layer_scan(self._h, C.byref(n)), n)

    def scan_filtered(self, fn):
        n = C.c_size_t(0)
        cb = FILTER_FN(lambda a, b, _u: 1 if fn(a, b) else 0)
        return self._pairs(self._L.orc_layer_scan_filtered(self._h, cb, None, C.byref(n)), n)

    def par_scan(self, threads=None):
        n = C.c_size_t(0)
        return self._pairs(self._L.orc_layer_par_scan(self._h, threads or self._L.orc_hw_threads(), C.byref(n)), n)

    def par_scan_filtered(self, fn, threads=None):
        n = C.c_size_t(0)
        cb = FILTER_FN(lambda a, b, _u: 1 if fn(a, b) else 0)
        return self._pairs(self._L.orc_layer_par_scan_filtered(
            self._h, threads or self._L.orc_hw_threads(), cb, None, C.byref(n)), n)

    def scan_raw(self):
        """(raw pairs in scan_impl emission order, max stack depth)"""
        n, ms = C.c_size_t(0), C.c_size_t(0)
        p = self._L.orc_layer_scan_raw(self._h, C.byref(n), C.byref(ms))
        return self._pairs(p, n), ms.value

    def test_box(self, system_bounds, test_bounds, max_depth=None):
        """Layer::test_box (layer.rs:294-313): sorted unique ids whose cells overlap test_bounds = (min, max)"""
        sys_min, sys_max = _f32(system_bounds[0]), _f32(system_bounds[1])
        tmin, tmax = _f32(test_bounds[0]), _f32(test_bounds[1])
        out = np.zeros(max(len(self), 1), dtype=np.uint32)
        n = self._L.orc_layer_test_box(self._h, _ptr(sys_min, C.c_float), _ptr(sys_max, C.c_float), _ptr(tmin, C.c_float),
                                       _ptr(tmax, C.c_float), 0 if max_depth is None else 1,
                                       0 if max_depth is None else int(max_depth), _ptr(out, C.c_uint32), len(out))
        if n < 0:
            raise RuntimeError("orc_layer_test_box failed (%d)" % n)
        return out[:n].copy()

    def test_ray(self, system_bounds, origin, direction, range_min=-np.inf, range_max=np.inf, max_depth=None):
        """Layer::test_ray (layer.rs:327-351): sorted unique ids of the cells the ray passes through"""
        sys_min, sys_max = _f32(system_bounds[0]), _f32(system_bounds[1])
        o, d = _f32(origin), _f32(direction)
        out = np.zeros(max(len(self), 1), dtype=np.uint32)
        n = self._L.orc_layer_test_ray(self._h, _ptr(sys_min, C.c_float), _ptr(sys_max, C.c_float), _ptr(o, C.c_float),
                                       _ptr(d, C.c_float), float(range_min), float(range_max), 0 if max_depth is None else 1,
                                       0 if max_depth is None else int(max_depth), _ptr(out, C.c_uint32), len(out))
        if n < 0:
            raise RuntimeError("orc_layer_test_ray failed (%d)" % n)
        return out[:n].copy()

    def pick_ray_aabb(self, system_bounds, origin, direction, max_dist, bounds_by_id, max_depth=None):
        """Layer::pick_ray (layer.rs:418-446) with the built-in ray/AABB distance: (dist, id) or None"""
        sys_min, sys_max = _f32(system_bounds[0]), _f32(system_bounds[1])
        o, d = _f32(origin), _f32(direction)
        b = _f32(bounds_by_id).reshape(-1, 2 * self.dim)
        dist, idv = C.c_float(), C.c_uint32()
        rc = self._L.orc_layer_pick_ray_aabb(self._h, _ptr(sys_min, C.c_float), _ptr(sys_max, C.c_float), _ptr(o, C.c_float),
                                             _ptr(d, C.c_float), float(max_dist), 0 if max_depth is None else 1,
                                             0 if max_depth is None else int(max_depth), _ptr(b, C.c_float), len(b),
                                             C.byref(dist), C.byref(idv))
        if rc < 0:
            raise RuntimeError("orc_layer_pick_ray_aabb failed (%d)" % rc)
        return (dist.value, idv.value) if rc else None

    def __len__(self):
        return self._L.orc_layer_len(self._h)

    @property
    def keys(self):
        n = len(self)
        if n == 0:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(self._L.orc_layer_keys(self._h), shape=(n,)).copy()

    @property
    def ids(self):
        n = len(self)
        if n == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(self._L.orc_layer_ids(self._h), shape=(n,)).copy()

    @property
    def sorted(self):
        return bool(self._L.orc_layer_sorted(self._h))

    @property
    def min_depth(self):
        return self._L.orc_layer_min_depth(self._h)

    @property
    def num_invalid(self):
        return self._L.orc_layer_num_invalid(self._h)
