"""`BR_SCENE` container of the reference's test data (data/src/lib.rs:16-106), reader and writer.

bincode 1.x default encoding: little-endian fixed-width integers, `u64` length prefix for `Vec`,
one tag byte for `Option`/`bool`, structs and tuples as their fields in order.

  Header      signature b"BR_SCENE" (8 bytes, no length) + version (u16, u16)         lib.rs:19-26
  SceneV1_0   system_bounds: 6 x f32 | object_bounds: Vec<(6 x f32, u32)>             lib.rs:28-32
  SceneV1_1   ... + layer: min_depth u32 | tree: (Vec<(u64, u32)>, bool)              lib.rs:34-39, layer.rs:49-50
  SceneV1_2   ... + collisions: Vec<(u32, u32)> | hits: Vec<u32> | nearest: Option<(u32, f32)>   lib.rs:41-49

`ID = u32`, `Index = Index64_3D` (lib.rs:16-17).  Files written here can be read by the crate's
`Scene::load`, and its files by `load`.  The writer is byte-exact: the reference's seven input scenes and its
three validation goldens, rebuilt with it, hash to the SHA-256 of their Git-LFS pointers
(tests/test_reference_goldens.py).
"""
import io
import struct
from dataclasses import dataclass, field

import numpy as np

SIGNATURE = b"BR_SCENE"
VERSION = (1, 2)

_OBJ = np.dtype([("bounds", "<f4", (6,)), ("id", "<u4")])           # (Bounds<Point3<f32>>, ID): 28 bytes
_ENT = np.dtype([("key", "<u8"), ("id", "<u4")])                    # (Index64_3D, ID): 12 bytes
_PAIR = np.dtype([("a", "<u4"), ("b", "<u4")])


@dataclass
class Scene:
    system_bounds: np.ndarray = field(default_factory=lambda: np.zeros((2, 3), np.float32))  # [min, max]
    bounds: np.ndarray = field(default_factory=lambda: np.zeros((0, 6), np.float32))          # rows: min xyz, max xyz
    ids: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    layer_min_depth: int = 0
    layer_keys: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint64))
    layer_ids: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    layer_sorted: bool = True                     # Layer::default(): tree = (vec![], false) -> see parse()
    collisions: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.uint32))
    hits: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    nearest: tuple = None                         # Option<(ID, f32)>
    version: tuple = VERSION

    # ---- reading ---------------------------------------------------------------------------------
    @staticmethod
    def parse(data):
        """Scene::parse (lib.rs:69-88): accepts versions 1.0, 1.1 and 1.2"""
        buf = memoryview(data)
        if bytes(buf[:8]) != SIGNATURE:
            raise ValueError("InvalidSignature(%r)" % bytes(buf[:8]))
        major, minor = struct.unpack_from("<HH", buf, 8)
        if major != VERSION[0] or minor > VERSION[1]:
            raise ValueError("UnsupportedVersion((%d, %d))" % (major, minor))
        pos = 12
        s = Scene(version=(major, minor))
        s.system_bounds = np.frombuffer(buf, "<f4", 6, pos).reshape(2, 3).copy()
        pos += 24
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        objs = np.frombuffer(buf, _OBJ, n, pos)
        pos += n * _OBJ.itemsize
        s.bounds, s.ids = objs["bounds"].copy(), objs["id"].copy()
        if minor >= 1:
            (s.layer_min_depth,) = struct.unpack_from("<I", buf, pos)
            pos += 4
            (m,) = struct.unpack_from("<Q", buf, pos)
            pos += 8
            ent = np.frombuffer(buf, _ENT, m, pos)
            pos += m * _ENT.itemsize
            s.layer_keys, s.layer_ids = ent["key"].copy(), ent["id"].copy()
            s.layer_sorted = bool(buf[pos])
            pos += 1
        else:
            s.layer_sorted = False  # `layer: Default::default()` in From<SceneV1_0> (derive(Default): bool = false)
        if minor >= 2:
            (p,) = struct.unpack_from("<Q", buf, pos)
            pos += 8
            pr = np.frombuffer(buf, _PAIR, p, pos)
            pos += p * _PAIR.itemsize
            s.collisions = np.stack([pr["a"], pr["b"]], axis=1).copy() if p else np.zeros((0, 2), np.uint32)
            (h,) = struct.unpack_from("<Q", buf, pos)
            pos += 8
            s.hits = np.frombuffer(buf, "<u4", h, pos).copy()
            pos += 4 * h
            tag = buf[pos]
            pos += 1
            if tag:
                s.nearest = struct.unpack_from("<If", buf, pos)
                pos += 8
        if pos != len(buf):
            raise ValueError("trailing bytes: parsed %d of %d" % (pos, len(buf)))
        return s

    @staticmethod
    def load(path):
        with open(path, "rb") as f:
            return Scene.parse(f.read())

    # ---- writing ---------------------------------------------------------------------------------
    def assemble(self, version=VERSION):
        """Scene::assemble (lib.rs:97-105); the crate always writes (1, 2), older layouts are for fixtures"""
        out = io.BytesIO()
        out.write(SIGNATURE)
        out.write(struct.pack("<HH", *version))
        out.write(np.asarray(self.system_bounds, "<f4").reshape(6).tobytes())
        n = len(self.ids)
        out.write(struct.pack("<Q", n))
        objs = np.zeros(n, _OBJ)
        objs["bounds"] = np.asarray(self.bounds, "<f4").reshape(n, 6)
        objs["id"] = np.asarray(self.ids, "<u4")
        out.write(objs.tobytes())
        if version[1] >= 1:
            out.write(struct.pack("<I", int(self.layer_min_depth)))
            m = len(self.layer_keys)
            out.write(struct.pack("<Q", m))
            ent = np.zeros(m, _ENT)
            ent["key"], ent["id"] = self.layer_keys, self.layer_ids
            out.write(ent.tobytes())
            out.write(b"\x01" if self.layer_sorted else b"\x00")
        if version[1] >= 2:
            c = np.asarray(self.collisions, "<u4").reshape(-1, 2)
            out.write(struct.pack("<Q", len(c)))
            out.write(np.ascontiguousarray(c).tobytes())
            out.write(struct.pack("<Q", len(self.hits)))
            out.write(np.asarray(self.hits, "<u4").tobytes())
            if self.nearest is None:
                out.write(b"\x00")
            else:
                out.write(b"\x01" + struct.pack("<If", int(self.nearest[0]), float(self.nearest[1])))
        return out.getvalue()

    def save(self, path, version=VERSION):
        with open(path, "wb") as f:
            f.write(self.assemble(version))

    # ---- bridges to the Layer API -------------------------------------------------------------------
    def extend_into(self, layer):
        """input.layer.extend(input.system_bounds, input.object_bounds) -- tests/test_layer.rs:30-32"""
        layer.extend((self.system_bounds[0], self.system_bounds[1]), self.bounds, self.ids)

    def set_layer_from(self, layer):
        keys, ids = layer.tree()
        self.layer_min_depth, self.layer_keys, self.layer_ids, self.layer_sorted = layer.min_depth, keys, ids, layer.sorted
