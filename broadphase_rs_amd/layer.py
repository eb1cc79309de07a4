"""ctypes mirror of the reference's `Layer<Index, ID>` / `LayerBuilder` (src/layer.rs:42-68, 619-696)
over the C ABI of include/bp_layer.h.  Same method names, argument meaning and error behaviour:
out-of-bounds objects are skipped silently, results are borrowed until the next mutating call,
`par_*` are aliases of the serial names (one device implementation).

numpy arrays are host inputs; anything exposing `data_ptr()` and `is_cuda` (a torch tensor on the
GPU) is passed through as a device pointer without a copy.
"""
import ctypes as C
import os
import subprocess
from collections import namedtuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BP_LIBRARY: load another build of the same library (A/B measurements of kernel variants: `make OUT=../lib_ab/x EXTRA=-D...`)
_LIB = os.environ.get("BP_LIBRARY") or os.path.join(_HERE, "lib", "libbroadphase_amd.so")

Bounds = namedtuple("Bounds", ["min", "max"])  # src/geom.rs:84-87 (inclusive)


class _IndexKind:
    def __init__(self, name, kind, dim, bits, depth_bits, axis_bits):
        self.name, self.kind, self.dim, self.bits = name, kind, dim, bits
        self.depth_bits, self.axis_bits = depth_bits, axis_bits

    def __repr__(self):
        return self.name


# src/index.rs:293-295
Index32_2D = _IndexKind("Index32_2D", 0, 2, 32, 4, 14)
Index64_2D = _IndexKind("Index64_2D", 1, 2, 64, 5, 29)
Index64_3D = _IndexKind("Index64_3D", 2, 3, 64, 5, 19)
_KINDS = {0: Index32_2D, 1: Index64_2D, 2: Index64_3D}

KERNEL_CLASSES = ("encode", "tree_hist", "tree_sort", "scan_count", "scan_emit", "pair_hist", "pair_sort",
                  "dedup", "misc", "tree_finish")


class BroadphaseError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (status, message))
        self.status = status


class _Facts(C.Structure):
    _fields_ = [("key_or", C.c_uint64), ("key_and", C.c_uint64), ("id_or", C.c_uint32), ("id_and", C.c_uint32),
                ("id_min", C.c_uint32), ("id_max", C.c_uint32), ("depth_mask", C.c_uint32), ("flags", C.c_uint32)]


FACT_ORDER_OK, FACT_IDS_DISTINCT, FACT_IDS_CARRY_BOX, FACT_BOX_ALL_KNOWN = 1, 2, 4, 8


class _Stats(C.Structure):
    _fields_ = [("objects_valid", C.c_uint64), ("objects_invalid", C.c_uint64), ("entries", C.c_uint64),
                ("raw_pairs", C.c_uint64), ("pairs", C.c_uint64), ("raw_candidates", C.c_uint64),
                ("tree_sort_passes", C.c_uint32),
                ("pair_sort_passes", C.c_uint32), ("depth_mask", C.c_uint32), ("flags", C.c_uint32),
                ("host_syncs", C.c_uint64), ("extend_reruns", C.c_uint32), ("tree_plan_flips", C.c_uint32),
                ("pair_plan_flips", C.c_uint32), ("sort_repairs", C.c_uint32), ("sorts_ahead", C.c_uint32),
                ("sorts_ahead_missed", C.c_uint32), ("sorts_in_cells", C.c_uint32),
                ("tree_sorts_in_cells", C.c_uint32), ("tree_sorts_compact", C.c_uint32), ("tree_expands", C.c_uint32)]


FILTER_FN = C.CFUNCTYPE(C.c_int, C.c_uint32, C.c_uint32, C.c_void_p)

_lib = None

# every symbol include/bp_layer.h declares
EXPORTS = [
    "bp_status_string", "bp_layer_last_error", "bp_layer_create", "bp_layer_destroy", "bp_layer_clone",
    "bp_layer_equals", "bp_layer_clear", "bp_layer_extend", "bp_layer_extend_host", "bp_layer_extend_device",
    "bp_layer_merge", "bp_layer_sort", "bp_layer_par_sort", "bp_layer_scan", "bp_layer_par_scan",
    "bp_layer_scan_device", "bp_layer_scan_filtered", "bp_layer_par_scan_filtered", "bp_layer_scan_raw",
    "bp_layer_tree", "bp_layer_tree_device", "bp_layer_set_tree", "bp_layer_append_entries", "bp_layer_len",
    "bp_layer_is_sorted", "bp_layer_min_depth", "bp_layer_index_kind", "bp_layer_device",
    "bp_layer_scan_shard_device", "bp_layer_sort_unique_pairs_device", "bp_layer_stats",
    "bp_layer_profile_enable", "bp_layer_profile_select", "bp_layer_profile_sample", "bp_layer_profile_reset", "bp_layer_profile_get", "bp_version",
    "bp_layer_append_entries_assume", "bp_layer_cell_histogram", "bp_layer_partition_entries_device",
    "bp_layer_scan_raw_shard_device", "bp_layer_partition_pairs_device", "bp_layer_scan_shard_partition_device",
    "bp_layer_set_coarse_emit_range",
    "bp_layer_set_cell_histogram_depth",
    "bp_layer_test_box_batch", "bp_layer_test_box_batch_device", "bp_layer_test_ray_batch", "bp_layer_test_ray_batch_device",
    "bp_layer_pick_ray_aabb_batch",
    "bp_layer_sort_unique_packed_device",
    "bp_layer_can_pack_box", "bp_layer_partition_entries_packed_device", "bp_layer_scan_shard_pairs_device",
    "bp_layer_sort_packed_device", "bp_layer_scan_filtered_aabb", "bp_layer_scan_filtered_aabb_device",
    "bp_layer_scan_filtered_sphere", "bp_layer_scan_filtered_sphere_device", "bp_layer_par_scan_shard_device", "bp_layer_sort_packed_range_device",
    "bp_layer_get_facts", "bp_layer_begin_entries", "bp_layer_commit_entries",
    "bp_layer_wait_stream", "bp_layer_stream", "bp_layer_reset_plan_hints",
]


def library_path():
    return _LIB


def build_library(force=False):
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in os.listdir(src_dir)]
    srcs.append(os.path.join(_HERE, "..", "include", "bp_layer.h"))
    if (not force and os.path.exists(_LIB) and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in srcs)):
        return _LIB
    subprocess.check_call(["make", "-C", src_dir, "-s", "all"])
    return _LIB


def _preload_hip_runtime():
    """A process can host only ONE HIP runtime.  PyTorch-ROCm wheels bundle their own
    libamdhip64 (SONAME libamdhip64.so.7, the same as /opt/rocm's), so if torch is installed and
    no HIP runtime is mapped yet, map torch's copy first: the library's DT_NEEDED then resolves
    to it by SONAME and a later `import torch` in this process shares the same runtime.  Without
    torch the library's RPATH (/opt/rocm/lib) is used.  BP_HIP_RUNTIME=system skips this."""
    if os.environ.get("BP_HIP_RUNTIME", "") == "system":
        return
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return
    except OSError:
        pass
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library():
    """dlopen the in-tree HIP library.  Fails loudly: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    _preload_hip_runtime()
    if not os.path.exists(_LIB):
        raise BroadphaseError("BP_ERR_NO_LIBRARY", "HIP library %s is missing -- run "
                              "`python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists)" % _LIB)
    L = C.CDLL(_LIB)
    vp, u32p, u64p, f32p, szp = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_float), \
        C.POINTER(C.c_size_t)
    ip = C.POINTER(C.c_int)
    L.bp_version.restype = C.c_char_p
    L.bp_status_string.restype = C.c_char_p
    L.bp_status_string.argtypes = [C.c_int]
    L.bp_layer_last_error.restype = C.c_char_p
    L.bp_layer_last_error.argtypes = [vp]
    L.bp_layer_create.argtypes = [C.POINTER(vp), C.c_int, C.c_uint32, C.c_size_t, C.c_size_t, C.c_int]
    L.bp_layer_destroy.argtypes = [vp]
    L.bp_layer_clone.argtypes = [vp, C.POINTER(vp)]
    L.bp_layer_equals.argtypes = [vp, vp, ip]
    L.bp_layer_clear.argtypes = [vp]
    for name in ("bp_layer_extend", "bp_layer_extend_host", "bp_layer_extend_device"):
        getattr(L, name).argtypes = [vp, f32p, f32p, vp, vp, C.c_size_t]
    L.bp_layer_merge.argtypes = [vp, vp]
    L.bp_layer_sort.argtypes = [vp]
    L.bp_layer_par_sort.argtypes = [vp]
    for name in ("bp_layer_scan", "bp_layer_par_scan", "bp_layer_scan_device", "bp_layer_scan_raw"):
        getattr(L, name).argtypes = [vp, C.POINTER(vp), szp]
    for name in ("bp_layer_scan_filtered", "bp_layer_par_scan_filtered"):
        getattr(L, name).argtypes = [vp, FILTER_FN, vp, C.POINTER(vp), szp]
    L.bp_layer_tree.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), szp, ip]
    L.bp_layer_tree_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), szp, ip]
    L.bp_layer_set_tree.argtypes = [vp, C.c_uint32, vp, vp, C.c_size_t, C.c_int]
    L.bp_layer_append_entries.argtypes = [vp, vp, vp, C.c_size_t]
    L.bp_layer_len.argtypes = [vp, szp]
    L.bp_layer_is_sorted.argtypes = [vp, ip]
    L.bp_layer_min_depth.argtypes = [vp, u32p]
    L.bp_layer_index_kind.argtypes = [vp, ip]
    L.bp_layer_device.argtypes = [vp, ip]
    L.bp_layer_scan_shard_device.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(vp), szp]
    L.bp_layer_sort_unique_pairs_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp), szp]
    L.bp_layer_append_entries_assume.argtypes = [vp, vp, vp, C.c_size_t, C.c_uint32]
    L.bp_layer_cell_histogram.argtypes = [vp, C.c_uint32, u64p, C.c_size_t]
    L.bp_layer_partition_entries_device.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint32, u64p,
                                                    C.POINTER(vp), C.POINTER(vp)]
    L.bp_layer_scan_raw_shard_device.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(vp), szp]
    L.bp_layer_set_coarse_emit_range.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.bp_layer_can_pack_box.argtypes = [vp, ip]
    L.bp_layer_partition_entries_packed_device.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint8), C.c_size_t, C.c_uint32,
                                                           C.c_int, u64p, C.POINTER(vp), C.POINTER(vp)]
    L.bp_layer_scan_shard_pairs_device.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(vp), szp, ip]
    L.bp_layer_par_scan_shard_device.argtypes = [vp, C.c_uint32, C.c_int, C.POINTER(vp), szp, ip]
    L.bp_layer_sort_packed_device.argtypes = [vp, vp, C.c_size_t, C.c_uint64, C.POINTER(vp), szp]
    L.bp_layer_sort_packed_range_device.argtypes = [vp, vp, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(vp), szp]
    L.bp_layer_get_facts.argtypes = [vp, C.POINTER(_Facts)]
    L.bp_layer_begin_entries.argtypes = [vp, C.c_size_t, C.POINTER(vp), C.POINTER(vp)]
    L.bp_layer_commit_entries.argtypes = [vp, C.c_size_t, C.POINTER(_Facts)]
    L.bp_layer_scan_filtered_aabb.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp), szp]
    L.bp_layer_scan_filtered_aabb_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp), szp]
    L.bp_layer_scan_filtered_sphere.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp), szp]
    L.bp_layer_scan_filtered_sphere_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(vp), szp]
    L.bp_layer_partition_pairs_device.argtypes = [vp, vp, C.c_size_t, u32p, C.c_uint32, u64p, C.POINTER(vp)]
    L.bp_layer_scan_shard_partition_device.argtypes = [vp, C.c_uint32, C.c_int, u32p, C.c_uint32, u64p, C.POINTER(vp),
                                                       C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    L.bp_layer_sort_unique_packed_device.argtypes = [vp, vp, C.c_size_t, C.c_uint64, C.POINTER(vp), szp]
    L.bp_layer_stats.argtypes = [vp, C.POINTER(_Stats)]
    L.bp_layer_profile_enable.argtypes = [vp, C.c_int]
    L.bp_layer_profile_reset.argtypes = [vp]
    L.bp_layer_pick_ray_aabb_batch.argtypes = [vp, f32p, f32p, vp, C.c_size_t, C.c_int, C.c_uint32, vp, C.c_size_t, vp, vp]
    for fn in (L.bp_layer_test_box_batch, L.bp_layer_test_box_batch_device, L.bp_layer_test_ray_batch,
               L.bp_layer_test_ray_batch_device):
        fn.argtypes = [vp, f32p, f32p, vp, C.c_size_t, C.c_int, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.bp_layer_reset_plan_hints.argtypes = [vp]
    L.bp_layer_wait_stream.argtypes = [vp, vp]
    L.bp_layer_stream.argtypes = [vp, C.POINTER(vp)]
    L.bp_layer_profile_select.argtypes = [vp, C.c_uint32]
    L.bp_layer_profile_sample.argtypes = [vp, C.c_uint32]
    L.bp_layer_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), u64p]
    for name in EXPORTS:
        if name not in ("bp_version", "bp_status_string", "bp_layer_last_error"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _is_device_tensor(x):
    return hasattr(x, "data_ptr") and bool(getattr(x, "is_cuda", False))


class DeviceArray:
    """Borrowed view of device memory owned by a Layer (valid until its next mutating call)."""

    def __init__(self, ptr, shape, dtype, owner):
        self.ptr, self.shape, self.dtype, self._owner = ptr, tuple(shape), np.dtype(dtype), owner

    @property
    def nbytes(self):
        return int(np.prod(self.shape)) * self.dtype.itemsize

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr or 0, True), "version": 2}

    def __len__(self):
        return self.shape[0]

    def to_numpy(self):
        """synchronous D2H copy (hipMemcpy through the HIP runtime this process already loaded)"""
        out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes:
            hip = C.CDLL("libamdhip64.so.7")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            rc = hip.hipMemcpy(C.c_void_p(out.ctypes.data), C.c_void_p(self.ptr), out.nbytes, 2)
            if rc != 0:
                raise BroadphaseError("BP_ERR_HIP", "hipMemcpy D2H failed with %d" % rc)
        return out

    def as_torch(self):
        """zero-copy torch view (signed dtype of the same width: torch has no full uint32/uint64)"""
        import torch

        class _View:
            pass
        v = _View()
        signed = {4: "<i4", 8: "<i8"}[self.dtype.itemsize]
        v.__cuda_array_interface__ = {"shape": self.shape, "typestr": signed, "data": (self.ptr or 0, False), "version": 2}
        # the OWNING layer's device, not torch's current one: a view on another device would silently be a copy
        dev = torch.device("cuda", self._owner.device if self._owner is not None else torch.cuda.current_device())
        if int(np.prod(self.shape)) == 0:
            return torch.empty(self.shape, dtype={4: torch.int32, 8: torch.int64}[self.dtype.itemsize], device=dev)
        t = torch.as_tensor(v, device=dev)
        if t.data_ptr() != self.ptr:
            raise BroadphaseError("BP_ERR_INVALID_ARG", "as_torch: torch made a copy instead of a zero-copy view")
        return t


class LayerBuilder:
    """src/layer.rs:619-696"""

    def __init__(self):
        self._min_depth, self._index_capacity, self._collision_capacity, self._test_capacity = 0, 0, 0, 0
        self._device = -1

    @staticmethod
    def new():
        return LayerBuilder()

    def with_min_depth(self, depth):
        self._min_depth = int(depth)
        return self

    def with_index_capacity(self, capacity):
        self._index_capacity = int(capacity)
        return self

    def with_collision_capacity(self, capacity):
        self._collision_capacity = int(capacity)
        return self

    def with_test_capacity(self, capacity):  # accepted, unused: the batched queries size their own scratch
        self._test_capacity = int(capacity)
        return self

    def with_device(self, device):  # not in the reference: which GPU the layer lives on
        self._device = int(device)
        return self

    def build(self, index=Index64_3D):
        return Layer(index, self._min_depth, self._index_capacity, self._collision_capacity, self._device)


class Layer:
    """Layer<Index, u32> (src/layer.rs:42-68): device-resident tree + scan temporaries."""

    def __init__(self, index=Index64_3D, min_depth=0, index_capacity=0, collision_capacity=0, device=-1, _handle=None):
        self._L = load_library()
        self.index = index
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        rc = self._L.bp_layer_create(C.byref(h), index.kind, min_depth, index_capacity, collision_capacity, device)
        if rc != 0:
            raise BroadphaseError(self._L.bp_status_string(rc).decode(),
                                  (self._L.bp_layer_last_error(None) or b"").decode())
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.bp_layer_destroy(h)
            self._h = None

    def _check(self, rc):
        if rc != 0:
            raise BroadphaseError(self._L.bp_status_string(rc).decode(),
                                  (self._L.bp_layer_last_error(self._h) or b"").decode())

    def wait_stream(self, stream_handle):
        """order the layer's work after everything enqueued so far on `stream_handle` (a hipStream_t as int; 0 = the
        legacy default stream) -- the input side of the stream-ordering contract in include/bp_layer.h"""
        self._check(self._L.bp_layer_wait_stream(self._h, C.c_void_p(int(stream_handle))))

    def _fence(self, *objs):
        """device tensors are read on the layer's own stream: fence it behind torch's current stream, where the
        kernels producing those tensors may still be in flight"""
        for o in objs:
            if _is_device_tensor(o):
                import torch
                self.wait_stream(torch.cuda.current_stream(o.device).cuda_stream)
                return

    # ---- reference methods ------------------------------------------------------------------------
    def clear(self):  # src/layer.rs:84-88
        self._check(self._L.bp_layer_clear(self._h))

    def extend(self, system_bounds, bounds, ids):
        """src/layer.rs:94-121.  bounds: (n, 2*dim) f32 rows [min.., max..]; ids: (n,) u32.
        CUDA torch tensors are read in place on the layer's stream (ordered behind torch's current stream): do not write to them
        again before the next call on this layer that returns a result (include/bp_layer.h, STREAM ORDERING)."""
        dim = self.index.dim
        # (the per-frame call: the two corner points go straight into C float arrays; numpy's conversion chain took 5 us)
        fvec = C.c_float * dim
        try:
            c_min, c_max = fvec(*system_bounds[0]), fvec(*system_bounds[1])
        except TypeError:  # nested / oddly shaped input: let numpy flatten it
            c_min = fvec(*np.asarray(system_bounds[0], dtype=np.float32).reshape(dim))
            c_max = fvec(*np.asarray(system_bounds[1], dtype=np.float32).reshape(dim))
        if _is_device_tensor(bounds) or _is_device_tensor(ids):
            if not (_is_device_tensor(bounds) and _is_device_tensor(ids)):
                raise ValueError("bounds and ids must both be device tensors or both be host arrays")
            n = int(ids.numel())
            if int(bounds.numel()) != n * 2 * dim or bounds.element_size() != 4 or ids.element_size() != 4:
                raise ValueError("bounds must be f32 (n, 2*dim) and ids 32-bit (n,)")
            if not (bounds.is_contiguous() and ids.is_contiguous()):
                raise ValueError("device tensors must be contiguous")
            self._fence(bounds, ids)
            self._check(self._L.bp_layer_extend_device(self._h, c_min, c_max, C.c_void_p(bounds.data_ptr()), C.c_void_p(ids.data_ptr()), n))
            return
        b = np.ascontiguousarray(np.asarray(bounds, dtype=np.float32).reshape(-1, 2 * dim))
        i = np.ascontiguousarray(np.asarray(ids, dtype=np.uint32).reshape(-1))
        if len(b) != len(i):
            raise ValueError("bounds and ids differ in length")
        self._check(self._L.bp_layer_extend_host(self._h, c_min, c_max, C.c_void_p(b.ctypes.data), C.c_void_p(i.ctypes.data), len(i)))

    def merge(self, other):  # src/layer.rs:127-138
        self._check(self._L.bp_layer_merge(self._h, other._h))

    def sort(self):  # src/layer.rs:159-165
        self._check(self._L.bp_layer_sort(self._h))

    def par_sort(self):  # src/layer.rs:146-152
        self._check(self._L.bp_layer_par_sort(self._h))

    def _host_pairs(self, p, n):
        n = n.value
        if n == 0:
            return np.zeros((0, 2), dtype=np.uint32)
        arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n, 2))
        return arr  # borrowed: valid until the next mutating call on this layer

    def scan(self):  # src/layer.rs:449-453
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_scan(self._h, C.byref(p), C.byref(n)))
        return self._host_pairs(p, n)

    def par_scan(self):  # src/layer.rs:482-488
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_par_scan(self._h, C.byref(p), C.byref(n)))
        return self._host_pairs(p, n)

    def _filtered(self, fn_name, filt):
        p, n = C.c_void_p(), C.c_size_t()
        cb = FILTER_FN(lambda a, b, _u: 1 if filt(a, b) else 0)
        self._check(getattr(self._L, fn_name)(self._h, cb, None, C.byref(p), C.byref(n)))
        return self._host_pairs(p, n)

    def scan_filtered(self, filt):  # src/layer.rs:460-477
        return self._filtered("bp_layer_scan_filtered", filt)

    def par_scan_filtered(self, filt):  # src/layer.rs:493-520
        return self._filtered("bp_layer_par_scan_filtered", filt)

    def _bounds_arg(self, bounds_by_id, width=None):
        width = width or 2 * self.index.dim
        if _is_device_tensor(bounds_by_id):
            if bounds_by_id.element_size() != 4 or not bounds_by_id.is_contiguous():
                raise ValueError("bounds must be contiguous f32")
            self._fence(bounds_by_id)
            return bounds_by_id.data_ptr(), int(bounds_by_id.numel()) // width, bounds_by_id
        b = np.ascontiguousarray(np.asarray(bounds_by_id, dtype=np.float32).reshape(-1, width))
        return b.ctypes.data, len(b), b

    def scan_filtered_aabb(self, bounds_by_id):
        """scan_filtered(|a, b| bounds[a].overlaps(bounds[b])) with the predicate evaluated on the device"""
        ptr, rows, keep = self._bounds_arg(bounds_by_id)
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_scan_filtered_aabb(self._h, C.c_void_p(ptr), rows, C.byref(p), C.byref(n)))
        return self._host_pairs(p, n)

    def scan_filtered_aabb_device(self, bounds_by_id):
        ptr, rows, keep = self._bounds_arg(bounds_by_id)
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_scan_filtered_aabb_device(self._h, C.c_void_p(ptr), rows, C.byref(p), C.byref(n)))
        return DeviceArray(p.value, (n.value, 2), np.uint32, self)

    def scan_filtered_sphere(self, spheres_by_id):
        """scan_filtered with the example's sphere test (examples/main.rs:459-474) on the device;
        rows (centre..., radius) indexed by id"""
        ptr, rows, keep = self._bounds_arg(spheres_by_id, self.index.dim + 1)
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_scan_filtered_sphere(self._h, C.c_void_p(ptr), rows, C.byref(p), C.byref(n)))
        return self._host_pairs(p, n)

    def scan_filtered_sphere_device(self, spheres_by_id):
        ptr, rows, keep = self._bounds_arg(spheres_by_id, self.index.dim + 1)
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_scan_filtered_sphere_device(self._h, C.c_void_p(ptr), rows, C.byref(p), C.byref(n)))
        return DeviceArray(p.value, (n.value, 2), np.uint32, self)

    # ---- batched region queries (src/layer.rs:294-313 for many boxes at once) -----------------------
    def _test_box_args(self, system_bounds, boxes, max_depth):
        f32p = C.POINTER(C.c_float)
        smin = np.ascontiguousarray(system_bounds[0], dtype=np.float32)
        smax = np.ascontiguousarray(system_bounds[1], dtype=np.float32)
        ptr, rows, keep = self._bounds_arg(boxes)
        return (smin.ctypes.data_as(f32p), smax.ctypes.data_as(f32p), C.c_void_p(ptr), rows,
                0 if max_depth is None else 1, 0 if max_depth is None else int(max_depth)), (smin, smax, keep)

    def test_box_batch(self, system_bounds, boxes, max_depth=None):
        """rows (query index, id), sorted, duplicate-free: group q == test_box(system_bounds, boxes[q], max_depth)"""
        args, _keep = self._test_box_args(system_bounds, boxes, max_depth)
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_test_box_batch(self._h, *args, C.byref(p), C.byref(n)))
        return self._host_pairs(p, n)

    def test_box_batch_device(self, system_bounds, boxes, max_depth=None):
        args, _keep = self._test_box_args(system_bounds, boxes, max_depth)
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_test_box_batch_device(self._h, *args, C.byref(p), C.byref(n)))
        return DeviceArray(p.value, (n.value, 2), np.uint32, self)

    def _query_rows(self, rows, width):
        if _is_device_tensor(rows):
            if rows.element_size() != 4 or not rows.is_contiguous():
                raise ValueError("query rows must be contiguous f32")
            self._fence(rows)
            return rows.data_ptr(), int(rows.numel()) // width, rows
        b = np.ascontiguousarray(np.asarray(rows, dtype=np.float32).reshape(-1, width))
        return b.ctypes.data, len(b), b

    def test_ray_batch(self, system_bounds, rays, max_depth=None, device=False):
        """rays: rows [origin(dim), direction(dim), range_min, range_max]; rows (query index, id) like test_box_batch"""
        f32p = C.POINTER(C.c_float)
        smin = np.ascontiguousarray(system_bounds[0], dtype=np.float32)
        smax = np.ascontiguousarray(system_bounds[1], dtype=np.float32)
        ptr, n_rays, _keep = self._query_rows(rays, 2 * self.index.dim + 2)
        p, n = C.c_void_p(), C.c_size_t()
        fn = self._L.bp_layer_test_ray_batch_device if device else self._L.bp_layer_test_ray_batch
        self._check(fn(self._h, smin.ctypes.data_as(f32p), smax.ctypes.data_as(f32p), C.c_void_p(ptr), n_rays,
                       0 if max_depth is None else 1, 0 if max_depth is None else int(max_depth), C.byref(p), C.byref(n)))
        return DeviceArray(p.value, (n.value, 2), np.uint32, self) if device else self._host_pairs(p, n)

    def pick_ray_aabb_batch(self, system_bounds, rays, bounds_by_id, max_depth=None):
        """Layer::pick_ray (src/layer.rs:418-446) for many rays with the built-in ray/AABB distance.
        rays: rows [origin(dim), direction(dim), max_dist] -> (dist f32[n] (+inf: no hit), id u32[n] (0xFFFFFFFF: no hit))"""
        f32p = C.POINTER(C.c_float)
        smin = np.ascontiguousarray(system_bounds[0], dtype=np.float32)
        smax = np.ascontiguousarray(system_bounds[1], dtype=np.float32)
        rptr, n_rays, _k1 = self._query_rows(rays, 2 * self.index.dim + 1)
        bptr, n_rows, _k2 = self._bounds_arg(bounds_by_id)
        dist = np.empty(n_rays, dtype=np.float32)
        ids = np.empty(n_rays, dtype=np.uint32)
        self._check(self._L.bp_layer_pick_ray_aabb_batch(
            self._h, smin.ctypes.data_as(f32p), smax.ctypes.data_as(f32p), C.c_void_p(rptr), n_rays,
            0 if max_depth is None else 1, 0 if max_depth is None else int(max_depth), C.c_void_p(bptr), n_rows,
            C.c_void_p(dist.ctypes.data), C.c_void_p(ids.ctypes.data)))
        return dist, ids

    def pick_ray_aabb(self, system_bounds, origin, direction, max_dist, bounds_by_id, max_depth=None):
        """(dist, id, point) of the nearest box hit, or None  (src/layer.rs:418-446)"""
        o = np.asarray(origin, np.float32); d = np.asarray(direction, np.float32)
        row = np.concatenate([o, d, np.array([max_dist], np.float32)])
        dist, ids = self.pick_ray_aabb_batch(system_bounds, row[None, :], bounds_by_id, max_depth)
        if ids[0] == 0xFFFFFFFF:
            return None
        return float(dist[0]), int(ids[0]), o + d * dist[0]

    def test_ray(self, system_bounds, origin, direction, range_min=-np.inf, range_max=np.inf, max_depth=None):  # src/layer.rs:327-351
        row = np.concatenate([np.asarray(origin, np.float32), np.asarray(direction, np.float32),
                              np.array([range_min, range_max], np.float32)])
        return np.array(self.test_ray_batch(system_bounds, row[None, :], max_depth)[:, 1])

    def test_box(self, system_bounds, test_bounds, max_depth=None):  # src/layer.rs:294-313
        """sorted duplicate-free ids whose cells overlap test_bounds = (min, max)"""
        box = np.concatenate([np.asarray(test_bounds[0], dtype=np.float32), np.asarray(test_bounds[1], dtype=np.float32)])
        return np.array(self.test_box_batch(system_bounds, box[None, :], max_depth)[:, 1])

    def iter(self):  # src/layer.rs:79-81
        keys, ids = self.tree()
        return zip(keys.tolist(), ids.tolist())

    # ---- device-resident variants and views ---------------------------------------------------------
    def scan_device(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_scan_device(self._h, C.byref(p), C.byref(n)))
        return DeviceArray(p.value, (n.value, 2), np.uint32, self)

    def scan_shard_device(self, coarse_depth, emit_coarse):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_scan_shard_device(self._h, coarse_depth, int(emit_coarse), C.byref(p),
                                                       C.byref(n)))
        return DeviceArray(p.value, (n.value, 2), np.uint32, self)

    def sort_unique_pairs_device(self, pairs_ptr, n):
        p, m = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_sort_unique_pairs_device(self._h, C.c_void_p(pairs_ptr), n, C.byref(p), C.byref(m)))
        return DeviceArray(p.value, (m.value, 2), np.uint32, self)

    def scan_raw(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_scan_raw(self._h, C.byref(p), C.byref(n)))
        return self._host_pairs(p, n)

    def tree(self):
        """(keys u64[m], ids u32[m]) in current order -- the serde view of the persistent state"""
        k, i, m, s = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_int()
        self._check(self._L.bp_layer_tree(self._h, C.byref(k), C.byref(i), C.byref(m), C.byref(s)))
        if m.value == 0:
            return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint32)
        keys = np.ctypeslib.as_array(C.cast(k, C.POINTER(C.c_uint64)), shape=(m.value,)).copy()
        ids = np.ctypeslib.as_array(C.cast(i, C.POINTER(C.c_uint32)), shape=(m.value,)).copy()
        return keys, ids

    def tree_device(self):
        k, i, m, s = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_int()
        self._check(self._L.bp_layer_tree_device(self._h, C.byref(k), C.byref(i), C.byref(m), C.byref(s)))
        return (DeviceArray(k.value, (m.value,), np.uint64, self), DeviceArray(i.value, (m.value,), np.uint32, self))

    def set_tree(self, min_depth, keys, ids, sorted_):
        self._fence(keys, ids)
        if _is_device_tensor(keys):
            kp, ip, m = keys.data_ptr(), ids.data_ptr(), int(keys.numel())
        else:
            keys = np.ascontiguousarray(keys, dtype=np.uint64)
            ids = np.ascontiguousarray(ids, dtype=np.uint32)
            kp, ip, m = keys.ctypes.data, ids.ctypes.data, len(keys)
        self._check(self._L.bp_layer_set_tree(self._h, min_depth, C.c_void_p(kp), C.c_void_p(ip), m, 1 if sorted_ else 0))

    def append_entries(self, keys, ids, count=None):
        self._fence(keys, ids)
        if _is_device_tensor(keys):
            kp, ip, m = keys.data_ptr(), ids.data_ptr(), int(keys.numel())
        elif isinstance(keys, int):
            kp, ip, m = keys, ids, int(count)
        else:
            keys = np.ascontiguousarray(keys, dtype=np.uint64)
            ids = np.ascontiguousarray(ids, dtype=np.uint32)
            kp, ip, m = keys.ctypes.data, ids.ctypes.data, len(keys)
        self._check(self._L.bp_layer_append_entries(self._h, C.c_void_p(kp), C.c_void_p(ip), m))

    # ---- building blocks of the Morton-range sharded pipeline (dist.py) -------------------------------
    def append_entries_assume(self, keys_ptr, ids_ptr, m, order_ok, ids_distinct, ids_carry_box=False, box_all_known=False):
        facts = (1 if order_ok else 0) | (2 if ids_distinct else 0) | (4 if ids_carry_box else 0) | (8 if box_all_known else 0)
        self._check(self._L.bp_layer_append_entries_assume(self._h, C.c_void_p(keys_ptr), C.c_void_p(ids_ptr), m, facts))

    def set_cell_histogram_depth(self, split_depth):
        self._L.bp_layer_set_cell_histogram_depth.argtypes = [C.c_void_p, C.c_uint32]
        self._check(self._L.bp_layer_set_cell_histogram_depth(self._h, int(split_depth)))

    def cell_histogram(self, split_depth):
        n_cells = 1 << (self.index.dim * split_depth)
        hist = np.zeros(n_cells + 1, dtype=np.uint64)
        self._check(self._L.bp_layer_cell_histogram(self._h, split_depth, hist.ctypes.data_as(C.POINTER(C.c_uint64)), n_cells))
        return hist

    def get_facts(self):
        f = _Facts()
        self._check(self._L.bp_layer_get_facts(self._h, C.byref(f)))
        return {name: int(getattr(f, name)) for name, _ in _Facts._fields_}

    def begin_entries(self, m):
        """clear, then device buffers for m entries the caller fills (keys u64[m], ids u32[m])"""
        k, i = C.c_void_p(), C.c_void_p()
        self._check(self._L.bp_layer_begin_entries(self._h, m, C.byref(k), C.byref(i)))
        return DeviceArray(k.value, (m,), np.uint64, self), DeviceArray(i.value, (m,), np.uint32, self)

    def commit_entries(self, m, facts):
        f = _Facts(**{k: int(v) for k, v in facts.items()})
        self._check(self._L.bp_layer_commit_entries(self._h, m, C.byref(f)))

    def can_pack_box(self):
        """(ids can carry box codes, all box codes known)"""
        c = C.c_int()
        self._check(self._L.bp_layer_can_pack_box(self._h, C.byref(c)))
        return bool(c.value & 1), bool(c.value & 2)

    def partition_entries_device(self, split_depth, lut, n_dest, pack_box=False):
        lut = np.ascontiguousarray(lut, dtype=np.uint8)
        counts = np.zeros(n_dest + 1, dtype=np.uint64)
        k, i = C.c_void_p(), C.c_void_p()
        self._check(self._L.bp_layer_partition_entries_packed_device(
            self._h, split_depth, lut.ctypes.data_as(C.POINTER(C.c_uint8)), len(lut), n_dest, 1 if pack_box else 0,
            counts.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(k), C.byref(i)))
        m = len(self)
        return counts, DeviceArray(k.value, (m,), np.uint64, self), DeviceArray(i.value, (m,), np.uint32, self)

    def set_coarse_emit_range(self, id_lo, id_hi):
        """shard scans with emit_coarse = 2: coarse entries act as the deeper side of a pair iff id_lo <= id <= id_hi"""
        self._check(self._L.bp_layer_set_coarse_emit_range(self._h, int(id_lo), int(id_hi)))

    def scan_raw_shard_device(self, coarse_depth, emit_coarse):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_scan_raw_shard_device(self._h, coarse_depth, int(emit_coarse), C.byref(p), C.byref(n)))
        return DeviceArray(p.value, (n.value,), np.uint64, self)

    def scan_shard_pairs_device(self, coarse_depth, emit_coarse):
        """(packed raw pairs in arbitrary order, duplicates-already-suppressed flag)"""
        p, n, u = C.c_void_p(), C.c_size_t(), C.c_int()
        self._check(self._L.bp_layer_scan_shard_pairs_device(self._h, coarse_depth, int(emit_coarse), C.byref(p),
                                                             C.byref(n), C.byref(u)))
        return DeviceArray(p.value, (n.value,), np.uint64, self), bool(u.value)

    def par_scan_shard_device(self, coarse_depth, emit_coarse):
        """(this shard's sorted duplicate-free pairs, no-other-shard-holds-them flag)"""
        p, n, u = C.c_void_p(), C.c_size_t(), C.c_int()
        self._check(self._L.bp_layer_par_scan_shard_device(self._h, coarse_depth, int(emit_coarse), C.byref(p),
                                                           C.byref(n), C.byref(u)))
        return DeviceArray(p.value, (n.value, 2), np.uint32, self), bool(u.value)

    def scan_shard_partition_device(self, coarse_depth, emit_coarse, bounds):
        """scan + bucketing of the raw pairs by owner of the first id -> (counts, grouped packed pairs, unique flag)"""
        bounds = np.ascontiguousarray(bounds, dtype=np.uint32)
        counts = np.zeros(len(bounds) + 1, dtype=np.uint64)
        out, n, u = C.c_void_p(), C.c_size_t(), C.c_int()
        self._check(self._L.bp_layer_scan_shard_partition_device(
            self._h, coarse_depth, int(emit_coarse), bounds.ctypes.data_as(C.POINTER(C.c_uint32)), len(bounds),
            counts.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(out), C.byref(n), C.byref(u)))
        return counts, DeviceArray(out.value, (n.value,), np.uint64, self), bool(u.value)

    def sort_packed_device(self, packed_ptr, n, varying_mask=0):
        p, m = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_sort_packed_device(self._h, C.c_void_p(packed_ptr), n, varying_mask, C.byref(p), C.byref(m)))
        return DeviceArray(p.value, (m.value, 2), np.uint32, self)

    def sort_packed_range_device(self, packed_ptr, n, varying_mask, first_lo, first_hi):
        p, m = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_sort_packed_range_device(self._h, C.c_void_p(packed_ptr), n, varying_mask, int(first_lo),
                                                              int(first_hi), C.byref(p), C.byref(m)))
        return DeviceArray(p.value, (m.value, 2), np.uint32, self)

    def partition_pairs_device(self, packed_ptr, n, bounds):
        bounds = np.ascontiguousarray(bounds, dtype=np.uint32)
        counts = np.zeros(len(bounds) + 1, dtype=np.uint64)
        out = C.c_void_p()
        self._check(self._L.bp_layer_partition_pairs_device(
            self._h, C.c_void_p(packed_ptr), n, bounds.ctypes.data_as(C.POINTER(C.c_uint32)), len(bounds),
            counts.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(out)))
        return counts, DeviceArray(out.value, (n,), np.uint64, self)

    def sort_unique_packed_device(self, packed_ptr, n, varying_mask=0):
        p, m = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_layer_sort_unique_packed_device(self._h, C.c_void_p(packed_ptr), n, varying_mask, C.byref(p), C.byref(m)))
        return DeviceArray(p.value, (m.value, 2), np.uint32, self)

    def clone(self):  # impl Clone, src/layer.rs:595-616
        h = C.c_void_p()
        self._check(self._L.bp_layer_clone(self._h, C.byref(h)))
        return Layer(self.index, _handle=h)

    def __eq__(self, other):  # impl PartialEq, src/layer.rs:576-586
        if not isinstance(other, Layer):
            return NotImplemented
        eq = C.c_int()
        self._check(self._L.bp_layer_equals(self._h, other._h, C.byref(eq)))
        return bool(eq.value)

    def __len__(self):
        m = C.c_size_t()
        self._check(self._L.bp_layer_len(self._h, C.byref(m)))
        return m.value

    @property
    def sorted(self):
        s = C.c_int()
        self._check(self._L.bp_layer_is_sorted(self._h, C.byref(s)))
        return bool(s.value)

    @property
    def min_depth(self):
        d = C.c_uint32()
        self._check(self._L.bp_layer_min_depth(self._h, C.byref(d)))
        return d.value

    @property
    def device(self):
        d = C.c_int()
        self._check(self._L.bp_layer_device(self._h, C.byref(d)))
        return d.value

    def stats(self):
        s = _Stats()
        self._check(self._L.bp_layer_stats(self._h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in _Stats._fields_}

    # ---- measurement ------------------------------------------------------------------------------------
    def reset_plan_hints(self):
        """next frame = the plan of a layer's first frame (same results; for timing the cold plan)"""
        self._check(self._L.bp_layer_reset_plan_hints(self._h))

    def profile_enable(self, on=True):
        self._check(self._L.bp_layer_profile_enable(self._h, 1 if on else 0))

    def profile_select(self, classes=None):
        """restrict kernel timing to the named classes (None: all)"""
        mask = 0xFFFFFFFF if classes is None else sum(1 << KERNEL_CLASSES.index(c) for c in classes)
        self._check(self._L.bp_layer_profile_select(self._h, mask))

    def profile_sample(self, every=1):
        """time only every `every`-th frame (frames are counted by clear())"""
        self._check(self._L.bp_layer_profile_sample(self._h, every))

    def profile_reset(self):
        self._check(self._L.bp_layer_profile_reset(self._h))

    def profile(self):
        """{kernel class: (total device ms, timed regions)} from HIP events on the layer's stream"""
        out = {}
        for cls, name in enumerate(KERNEL_CLASSES):
            ms, n = C.c_double(), C.c_uint64()
            self._check(self._L.bp_layer_profile_get(self._h, cls, C.byref(ms), C.byref(n)))
            out[name] = (ms.value, n.value)
        return out
