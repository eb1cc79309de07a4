"""ctypes mirror of include/bp_sharded.h: the multi-GPU `Layer` whose whole pipeline (partition, exchanges over
RCCL or between the ranks of one process, shard sort + scan, pair exchange) runs behind the C ABI -- what a Rust /
C++ host binds.  `dist.ShardedLayer` is the same pipeline driven from Python over torch.distributed.

    group = Group.rccl(rank, world, uid, device)      # one process per GPU; uid = Group.unique_id() made on rank 0
    group = Group.local(world)                        # `world` ranks inside this process, one host thread each
    sh = CShardedLayer(group, rank, index=Index64_3D, device=device, id_count=n_global)
    sh.clear(); sh.extend(system_bounds, bounds, ids); pairs = sh.par_scan_device()
"""
import ctypes as C

import numpy as np

from .layer import BroadphaseError, DeviceArray, Index64_3D, Layer, _is_device_tensor, load_library

UNIQUE_ID_BYTES = 128
PAIRS_GLOBAL, PAIRS_LOCAL = 0, 1

EXPORTS = [
    "bp_group_unique_id", "bp_group_create_rccl", "bp_group_create_local", "bp_group_destroy", "bp_group_last_error",
    "bp_sharded_create", "bp_sharded_destroy", "bp_sharded_last_error", "bp_sharded_clear", "bp_sharded_extend",
    "bp_sharded_par_scan", "bp_sharded_copy_pairs", "bp_sharded_last_info", "bp_sharded_shard_layer",
]


class _Info(C.Structure):
    _fields_ = [("entries_local", C.c_uint64), ("entries_shard", C.c_uint64), ("coarse", C.c_uint64), ("raw_pairs", C.c_uint64),
                ("raw_pairs_recv", C.c_uint64), ("pairs", C.c_uint64), ("id_lo", C.c_uint32), ("id_hi", C.c_uint32),
                ("box_codes", C.c_uint32), ("pair_order", C.c_uint32), ("stage_ms", C.c_float * 8),
                ("round_trips_layers", C.c_uint32), ("round_trips_exchanges", C.c_uint32)]


_bound = False


def _lib():
    global _bound
    L = load_library()
    if not _bound:
        vp, f32p = C.c_void_p, C.POINTER(C.c_float)
        L.bp_group_unique_id.argtypes = [vp]
        L.bp_group_create_rccl.argtypes = [C.POINTER(vp), C.c_int, C.c_int, vp, C.c_int]
        L.bp_group_create_local.argtypes = [C.POINTER(vp), C.c_int]
        L.bp_group_destroy.argtypes = [vp]
        L.bp_group_last_error.argtypes = [vp]
        L.bp_group_last_error.restype = C.c_char_p
        L.bp_sharded_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64]
        L.bp_sharded_destroy.argtypes = [vp]
        L.bp_sharded_last_error.argtypes = [vp]
        L.bp_sharded_last_error.restype = C.c_char_p
        L.bp_sharded_clear.argtypes = [vp]
        L.bp_sharded_extend.argtypes = [vp, f32p, f32p, vp, vp, C.c_size_t]
        L.bp_sharded_par_scan.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
        L.bp_sharded_copy_pairs.argtypes = [vp, vp, C.c_size_t]
        L.bp_sharded_last_info.argtypes = [vp, C.POINTER(_Info)]
        L.bp_sharded_shard_layer.argtypes = [vp, C.POINTER(vp)]
        for name in EXPORTS:
            if not name.endswith("last_error"):
                getattr(L, name).restype = C.c_int
        _bound = True
    return L


class Group:
    """the ranks of one job and their transport (bp_group_t)"""

    def __init__(self, handle, world, kind):
        self._L, self._h, self.world, self.kind = _lib(), handle, world, kind

    @staticmethod
    def unique_id():
        """rank 0: 128 bytes every rank passes to Group.rccl (ncclGetUniqueId)"""
        L = _lib()
        buf = (C.c_char * UNIQUE_ID_BYTES)()
        rc = L.bp_group_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise BroadphaseError(L.bp_status_string(rc).decode(), (L.bp_group_last_error(None) or b"").decode())
        return bytes(buf)

    @staticmethod
    def rccl(rank, world, unique_id, device):
        L = _lib()
        h = C.c_void_p()
        buf = (C.c_char * UNIQUE_ID_BYTES).from_buffer_copy(bytes(unique_id))
        rc = L.bp_group_create_rccl(C.byref(h), rank, world, C.cast(buf, C.c_void_p), device)
        if rc != 0:
            raise BroadphaseError(L.bp_status_string(rc).decode(), (L.bp_group_last_error(None) or b"").decode())
        return Group(h, world, "rccl")

    @staticmethod
    def local(world):
        L = _lib()
        h = C.c_void_p()
        rc = L.bp_group_create_local(C.byref(h), world)
        if rc != 0:
            raise BroadphaseError(L.bp_status_string(rc).decode(), (L.bp_group_last_error(None) or b"").decode())
        return Group(h, world, "local")

    def destroy(self):
        if self._h:
            self._L.bp_group_destroy(self._h)
            self._h = None


class CShardedLayer:
    """one rank's share of a multi-GPU Layer<Index, u32> (bp_sharded_t).  Every method is collective."""

    def __init__(self, group, rank, index=Index64_3D, min_depth=0, split_depth=0, device=0, id_count=0):
        self._L, self.group, self.rank, self.index = _lib(), group, rank, index
        h = C.c_void_p()
        rc = self._L.bp_sharded_create(C.byref(h), group._h, rank, index.kind, min_depth, split_depth, device, int(id_count))
        if rc != 0:
            raise BroadphaseError(self._L.bp_status_string(rc).decode(), (self._L.bp_group_last_error(group._h) or b"").decode())
        self._h = h
        self.device = device
        lh = C.c_void_p()
        self._check(self._L.bp_sharded_shard_layer(self._h, C.byref(lh)))
        self.shard_layer = Layer(index, _handle=lh)   # borrowed view: the handle belongs to the sharded layer

    def __del__(self):
        sl = getattr(self, "shard_layer", None)
        if sl is not None:
            sl._h = None       # owned by the sharded handle
        h = getattr(self, "_h", None)
        if h:
            self._L.bp_sharded_destroy(h)
            self._h = None

    def _check(self, rc):
        if rc != 0:
            raise BroadphaseError(self._L.bp_status_string(rc).decode(), (self._L.bp_sharded_last_error(self._h) or b"").decode())

    def clear(self):
        self._check(self._L.bp_sharded_clear(self._h))

    def extend(self, system_bounds, bounds, ids):
        dim = self.index.dim
        smin = np.ascontiguousarray(np.asarray(system_bounds[0], dtype=np.float32).reshape(dim))
        smax = np.ascontiguousarray(np.asarray(system_bounds[1], dtype=np.float32).reshape(dim))
        f32p = C.POINTER(C.c_float)
        if _is_device_tensor(bounds):
            import torch
            torch.cuda.current_stream(bounds.device).synchronize()   # the layers read on their own streams
            n = int(ids.numel())
            bp_, ip_ = bounds.data_ptr(), ids.data_ptr()
            keep = (bounds, ids)
        else:
            b = np.ascontiguousarray(np.asarray(bounds, dtype=np.float32).reshape(-1, 2 * dim))
            i = np.ascontiguousarray(np.asarray(ids, dtype=np.uint32).reshape(-1))
            n, bp_, ip_, keep = len(i), b.ctypes.data, i.ctypes.data, (b, i)
        self._check(self._L.bp_sharded_extend(self._h, smin.ctypes.data_as(f32p), smax.ctypes.data_as(f32p),
                                              C.c_void_p(bp_), C.c_void_p(ip_), n))
        del keep

    def par_scan_device(self, pair_order="global"):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._L.bp_sharded_par_scan(self._h, PAIRS_LOCAL if pair_order == "local" else PAIRS_GLOBAL, C.byref(p), C.byref(n)))
        return DeviceArray(p.value, (n.value, 2), np.uint32, self)

    def par_scan(self, pair_order="global"):
        out = self.par_scan_device(pair_order)
        host = np.empty((len(out), 2), dtype=np.uint32)
        self._check(self._L.bp_sharded_copy_pairs(self._h, C.c_void_p(host.ctypes.data), len(out)))
        return host

    scan = par_scan

    @property
    def last(self):
        info = _Info()
        self._check(self._L.bp_sharded_last_info(self._h, C.byref(info)))
        d = {k: int(getattr(info, k)) for k, _ in _Info._fields_ if k != "stage_ms"}
        names = ("hist_facts", "partition_entries", "exchange_entries", "shard_commit", "sort_scan_partition", "exchange_pairs",
                 "sort_pairs", "total")
        d["stage_ms"] = {n: round(float(v), 3) for n, v in zip(names, info.stage_ms)}
        d["pair_order"] = "local" if d["pair_order"] == PAIRS_LOCAL else "global"
        d["id_range"] = (d.pop("id_lo"), d.pop("id_hi"))
        return d
