"""END-TO-END PIN against the reference's own test data.

The reference keeps its input scenes and its three validation goldens (tests/data/**, consumed by tests/test_layer.rs)
in Git LFS; this tree only has the pointer files: sha256 + size (tests/golden/reference_lfs_oids.json).  The inputs are
regenerated bit-exactly here (tests/golden/ref_scene_gen.py restates the RNG and the sampling of
utils/src/gen_test_data.rs:98-155) and the goldens are re-derived with gen_validation_data's recipe
(utils/src/gen_test_data.rs:761-785: extend -> save, sort -> save, scan -> save).  Equality of the SHA-256 with the
LFS oids proves the regenerated inputs, the unsorted tree, the sorted tree and the collision list are byte-identical to
what the reference produced."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

import oracle
from broadphase_rs_amd import scene_io

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, GOLD)
import ref_scene_gen  # noqa: E402

OIDS = json.load(open(os.path.join(GOLD, "reference_lfs_oids.json")))


def sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.mark.parametrize("count", [100, 300])
def test_generator_reproduces_the_reference_input_scenes(count):
    meta = OIDS["inputs"]["boxes-seed_0-d_1_1000-s_1_10-n_%06d.br_scene" % count]
    sysb, bounds, ids = ref_scene_gen.gen_boxes(count, seed=0, density=0.001, size_range=(1.0, 10.0), cbrt_ulps=meta["cbrt_ulps"])
    blob = ref_scene_gen.scene_v1_0_bytes(sysb, bounds, ids)
    assert len(blob) == meta["size"] and sha(blob) == meta["sha256"]
    # the same bytes through the package's own BR_SCENE writer
    assert scene_io.Scene(system_bounds=sysb, bounds=bounds, ids=ids).assemble((1, 0)) == blob


@pytest.mark.parametrize("count", [100, 300, 1000, 3000, 10000, 30000, 100000])
def test_all_seven_reference_input_scenes_regenerate_to_their_lfs_hash(count):
    # vectorised generator (all ChaCha20 blocks at once); the scalar one above is the cross-check
    meta = OIDS["inputs"]["boxes-seed_0-d_1_1000-s_1_10-n_%06d.br_scene" % count]
    sysb, bounds, ids = ref_scene_gen.gen_boxes_fast(count, cbrt_ulps=meta["cbrt_ulps"])
    blob = ref_scene_gen.scene_v1_0_bytes(sysb, bounds, ids)
    assert len(blob) == meta["size"] and sha(blob) == meta["sha256"]
    # the package's generator (what tools/gen_validation_data.py --gen-inputs calls) with its own cbrt table: the same bytes
    from broadphase_rs_amd import scenes
    s2, b2, i2 = scenes.gen_boxes_reference(count)
    assert scene_io.Scene(system_bounds=s2, bounds=b2, ids=i2).assemble((1, 0)) == blob


@pytest.mark.parametrize("count", [1000, 10000])
def test_committed_input_fixtures_are_the_reference_files(count):
    meta = OIDS["inputs"]["boxes-seed_0-d_1_1000-s_1_10-n_%06d.br_scene" % count]
    blob = open(os.path.join(GOLD, "ref_input_n_%06d.br_scene" % count), "rb").read()
    assert len(blob) == meta["size"] and sha(blob) == meta["sha256"]


def validation_blobs(scene, layer):
    """utils/src/gen_test_data.rs:761-785 on any object with the Layer API.  The committed goldens were written when
    the container was at version 1.1 (files 0, 1: system bounds + layer, no object list) and 1.2 (file 2: whole scene);
    that is what their sizes say (57 + 12 M = 385137 with M = 32090) and what the hashes confirm."""
    sysb = (scene.system_bounds[0], scene.system_bounds[1])
    layer.extend(sysb, scene.bounds, scene.ids)
    out = {}
    s = scene_io.Scene(system_bounds=scene.system_bounds)
    s.layer_min_depth, s.layer_sorted = 0, False
    s.layer_keys, s.layer_ids = layer_tree(layer)
    out["0_layer_unsorted.br_scene"] = s.assemble((1, 1))
    layer.sort()
    s.layer_keys, s.layer_ids = layer_tree(layer)
    s.layer_sorted = True
    out["1_layer_sorted.br_scene"] = s.assemble((1, 1))
    full = scene_io.Scene(system_bounds=scene.system_bounds, bounds=scene.bounds, ids=scene.ids)
    full.layer_min_depth, full.layer_sorted, full.layer_keys, full.layer_ids = 0, True, s.layer_keys, s.layer_ids
    full.collisions = np.asarray(layer.scan(), dtype=np.uint32).reshape(-1, 2)
    out["2_layer_collisions.br_scene"] = full.assemble((1, 2))
    return out


def layer_tree(layer):
    if hasattr(layer, "tree"):
        k, i = layer.tree()
        return np.array(k), np.array(i)
    return layer.keys, layer.ids


def test_oracle_reproduces_the_reference_validation_goldens():
    scene = scene_io.Scene.load(os.path.join(GOLD, "ref_input_n_010000.br_scene"))
    blobs = validation_blobs(scene, oracle.OracleLayer(oracle.INDEX64_3D, 0))
    for name, meta in OIDS["validation"].items():
        assert len(blobs[name]) == meta["size"], name
        assert sha(blobs[name]) == meta["sha256"], "%s differs from the reference's golden" % name


@pytest.mark.gpu
def test_hip_path_reproduces_the_reference_validation_goldens():
    import broadphase_rs_amd as bp
    scene = scene_io.Scene.load(os.path.join(GOLD, "ref_input_n_010000.br_scene"))
    blobs = validation_blobs(scene, bp.Layer(bp.Index64_3D, 0))
    for name, meta in OIDS["validation"].items():
        assert len(blobs[name]) == meta["size"], name
        assert sha(blobs[name]) == meta["sha256"], "%s differs from the reference's golden" % name
    # par_sort / par_scan are the same lists (tests/test_layer.rs:74-124)
    lay = bp.Layer(bp.Index64_3D, 0)
    scene.extend_into(lay)
    lay.par_sort()
    full = scene_io.Scene(system_bounds=scene.system_bounds, bounds=scene.bounds, ids=scene.ids)
    full.set_layer_from(lay)
    full.collisions = np.array(lay.par_scan())
    assert sha(full.assemble((1, 2))) == OIDS["validation"]["2_layer_collisions.br_scene"]["sha256"]


@pytest.mark.gpu
def test_reference_test_layer_rs_mirrored_on_the_hip_path(tmp_path):
    """The five integration tests of the reference (tests/test_layer.rs:25-124), file for file: the validation files are
    rebuilt by the ORACLE (authenticated by their LFS sha256 above), written to disk, loaded back through the BR_SCENE
    reader, and the HIP layer is put through `extend`, `sort`, `par_sort`, `scan`, `par_scan` exactly as the Rust tests do."""
    import broadphase_rs_amd as bp
    inp = scene_io.Scene.load(os.path.join(GOLD, "ref_input_n_010000.br_scene"))
    for name, blob in validation_blobs(inp, oracle.OracleLayer(oracle.INDEX64_3D, 0)).items():
        assert sha(blob) == OIDS["validation"][name]["sha256"]
        (tmp_path / name).write_bytes(blob)
    v0 = scene_io.Scene.load(str(tmp_path / "0_layer_unsorted.br_scene"))
    v1 = scene_io.Scene.load(str(tmp_path / "1_layer_sorted.br_scene"))
    v2 = scene_io.Scene.load(str(tmp_path / "2_layer_collisions.br_scene"))

    def layer_equals(layer, scene):          # impl PartialEq for Layer (src/layer.rs:576-586)
        k, i = layer.tree()
        return (layer.min_depth == scene.layer_min_depth and layer.sorted == scene.layer_sorted and
                np.array_equal(k, scene.layer_keys) and np.array_equal(i, scene.layer_ids))

    def is_sorted(rows, unique):             # tests/test_layer.rs:42-54
        a = np.asarray(rows)
        key = a[:, 0].astype(np.uint64) << np.uint64(32) | a[:, 1].astype(np.uint64) if a.ndim == 2 else a
        return bool(np.all(key[1:] > key[:-1]) if unique else np.all(key[1:] >= key[:-1]))

    # fn extend (:25-40)
    lay = bp.Layer(bp.Index64_3D, 0)
    inp.extend_into(lay)
    assert layer_equals(lay, v0), "Layer::extend() produced unexpected results"
    # fn sort / fn par_sort (:56-90)
    for method in ("sort", "par_sort"):
        lay = bp.Layer(bp.Index64_3D, 0)
        lay.set_tree(v0.layer_min_depth, v0.layer_keys, v0.layer_ids, v0.layer_sorted)
        getattr(lay, method)()
        k, i = lay.tree()
        assert bool(np.all((k[1:] > k[:-1]) | ((k[1:] == k[:-1]) & (i[1:] >= i[:-1])))), "Layer::%s() produced unsorted output" % method
        assert layer_equals(lay, v1), "Layer::%s() produced unexpected results" % method
    # fn scan / fn par_scan (:92-124)
    for method in ("scan", "par_scan"):
        lay = bp.Layer(bp.Index64_3D, 0)
        lay.set_tree(v1.layer_min_depth, v1.layer_keys, v1.layer_ids, v1.layer_sorted)
        actual = np.array(getattr(lay, method)())
        assert is_sorted(actual, True), "Layer::%s() produced unsorted output" % method
        assert np.array_equal(actual, v2.collisions), "Layer::%s() produced unexpected results" % method


@pytest.mark.gpu
@pytest.mark.parametrize("count", [3000, 100000])
def test_hip_path_equals_the_pinned_oracle_on_the_reference_input_scenes(count):
    """the reference's other input scenes (no goldens exist for them): authenticated input, oracle pinned above"""
    import broadphase_rs_amd as bp
    meta = OIDS["inputs"]["boxes-seed_0-d_1_1000-s_1_10-n_%06d.br_scene" % count]
    sysb, bounds, ids = ref_scene_gen.gen_boxes_fast(count, cbrt_ulps=meta["cbrt_ulps"])
    assert sha(ref_scene_gen.scene_v1_0_bytes(sysb, bounds, ids)) == meta["sha256"]
    gpu, ref = bp.Layer(bp.Index64_3D, 0), oracle.OracleLayer(oracle.INDEX64_3D, 0)
    gpu.extend(sysb, bounds, ids); ref.extend(sysb, bounds, ids)
    k, i = gpu.tree()
    assert np.array_equal(k, ref.keys) and np.array_equal(i, ref.ids)
    gpu.par_sort(); ref.par_sort()
    k, i = gpu.tree()
    assert np.array_equal(k, ref.keys) and np.array_equal(i, ref.ids)
    assert np.array_equal(np.array(gpu.par_scan()), ref.par_scan())
