#!/usr/bin/env python3
"""The reference's golden pipeline (utils/src/gen_test_data.rs:761-785) on the MI355X path: load an
input scene, `extend` -> save 0_layer_unsorted, `sort` -> save 1_layer_sorted, `scan` -> save
2_layer_collisions, all in the crate's BR_SCENE (1, 2) layout.

  python tools/gen_validation_data.py <input.br_scene> <out_dir>          (needs a GPU)
  python tools/gen_validation_data.py --gen-inputs <out_dir>              (gen_test_scenes.py analogue, no GPU: the reference's
                                                                           seven input scenes, byte for byte -- their SHA-256 are the
                                                                           Git-LFS oids of tests/data/inputs)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from broadphase_rs_amd import scenes  # noqa: E402
from broadphase_rs_amd.scene_io import Scene  # noqa: E402


def gen_inputs(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    for count in (100, 300, 1000, 3000, 10000, 30000, 100000):   # tests/gen_test_scenes.py:12-33
        # the reference's own generator (ChaCha20 stream, rand 0.7 sampling), restated bit for bit in scenes.gen_boxes_reference
        sysb, bounds, ids = scenes.gen_boxes_reference(count, seed=0, density=0.001, size_range=(1.0, 10.0))
        s = Scene(system_bounds=sysb, bounds=bounds, ids=ids)
        s.save(os.path.join(out_dir, "boxes-seed_0-d_1_1000-s_1_10-n_%06d.br_scene" % count), version=(1, 0))


def gen_validation(input_path, out_dir):
    import broadphase_rs_amd as bp
    os.makedirs(out_dir, exist_ok=True)
    scene = Scene.load(input_path)
    layer = bp.LayerBuilder.new().with_min_depth(scene.layer_min_depth).build(bp.Index64_3D)
    scene.extend_into(layer)
    scene.set_layer_from(layer)
    scene.save(os.path.join(out_dir, "0_layer_unsorted.br_scene"))
    layer.sort()
    scene.set_layer_from(layer)
    scene.save(os.path.join(out_dir, "1_layer_sorted.br_scene"))
    scene.collisions = layer.scan().copy()
    scene.save(os.path.join(out_dir, "2_layer_collisions.br_scene"))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--gen-inputs":
        gen_inputs(sys.argv[2])
    elif len(sys.argv) == 3:
        gen_validation(sys.argv[1], sys.argv[2])
    else:
        raise SystemExit(__doc__)
