"""broadphase_rs_amd -- MI355X-native `Layer` extend -> sort -> scan (drop-in for the hot path of
zvxryb/broadphase-rs).

Host-side mirror of the reference's public surface for this path (src/lib.rs:80-82):
`Bounds`, `Layer`, `LayerBuilder`, `Index32_2D`, `Index64_2D`, `Index64_3D`.  All compute runs in
hand-written gfx950 kernels behind the C ABI of include/bp_layer.h (lib/libbroadphase_amd.so).
There is no CPU fallback: constructing a `Layer` without the HIP library or without a GPU raises.
"""
from .layer import (  # noqa: F401
    Bounds,
    BroadphaseError,
    Index32_2D,
    Index64_2D,
    Index64_3D,
    Layer,
    LayerBuilder,
    build_library,
    library_path,
    load_library,
)

__all__ = [
    "Bounds", "BroadphaseError", "Index32_2D", "Index64_2D", "Index64_3D", "Layer", "LayerBuilder",
    "build_library", "library_path", "load_library",
]
