#!/usr/bin/env python3
"""Generates the fixtures in this directory.

kat.json      -- the reference's own in-source known answers, transcribed as DATA (inputs/outputs):
                 src/index.rs:343-364 (encode/decode constants), src/geom.rs:696-706 (system ±64,
                 box ±32 round trip), SURVEY.md 8(c) worked example (surveyor-derived).
scene_*.npz   -- oracle-generated: inputs + unsorted tree + sorted tree + raw pair stream + final
                 pairs, for small seeded scenes and the adversarial cases no reference test covers.
                 (The reference's real goldens under tests/data are Git-LFS stubs, and the crate
                 cannot be built here, so these are NOT reference outputs: they pin the oracle and
                 the HIP path to each other and guard both against regressions.)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from broadphase_rs_amd import scenes  # noqa: E402


def dump(name, kind, min_depth, sysb, bounds, ids):
    lay = oracle.OracleLayer(kind, min_depth)
    lay.extend(sysb, bounds, ids)
    uk, ui = lay.keys, lay.ids
    lay.sort()
    sk, si = lay.keys, lay.ids
    raw, max_stack = lay.scan_raw()
    pairs = lay.scan()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), kind=kind, min_depth=min_depth,
                        sys_min=np.asarray(sysb[0], np.float32), sys_max=np.asarray(sysb[1], np.float32),
                        bounds=np.asarray(bounds, np.float32), ids=np.asarray(ids, np.uint32),
                        unsorted_keys=uk, unsorted_ids=ui, sorted_keys=sk, sorted_ids=si, raw_pairs=raw, pairs=pairs,
                        max_stack=max_stack, num_invalid=lay.num_invalid)
    print(name, len(ids), "objects", len(uk), "entries", len(raw), "raw", len(pairs), "pairs")


def main():
    kat = {
        "index_rs_343_364": {"kind": 2, "encode": [[0o1777777 << 13, 0o0001111111111111111111], [0, 0]],
                             "decode": [[0o0001111111111111111111, 0o1777777 << 13], [0o0006666666666666666666, 0]]},
        "geom_rs_696_706": {"sys_min": [-64.0] * 3, "sys_max": [64.0] * 3, "box_min": [-32.0] * 3, "box_max": [32.0] * 3,
                            "local_min": [0x3FFFFFC0] * 3, "local_max": [0xBFFFFF00] * 3},
        "survey_8c_worked_example": {
            "kind": 2, "min_depth": 0, "sys_min": [0, 0, 0], "sys_max": [16, 16, 16],
            "bounds": [[1, 1, 1, 2, 2, 2], [1.5, 1.5, 1.5, 3, 3, 3], [0, 0, 0, 16, 16, 16], [7, 7, 7, 9, 9, 9], [20, 0, 0, 21, 1, 1]],
            "ids": [0, 1, 2, 3, 4], "entries": 25, "raw_pairs": 32, "max_stack": 3,
            "pairs": [[0, 1], [0, 2], [1, 2], [3, 2]]},
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)
    dump("scene_boxes_100", 2, 0, *scenes.gen_boxes(100, seed=0))
    dump("scene_boxes_3000", 2, 0, *scenes.gen_boxes(3000, seed=0))
    dump("scene_boxes_1000_md5", 2, 5, *scenes.gen_boxes(1000, seed=1))
    dump("scene_circles_2000_i32", 0, 4, *scenes.gen_circles_2d(2000, seed=0))
    dump("scene_circles_1500_i64", 1, 0, *scenes.gen_circles_2d(1500, seed=2))
    sysb, b, _ = scenes.gen_boxes_powerlaw(1500, seed=4, alpha=1.5)
    dump("scene_powerlaw_dupids", 2, 0, sysb, b, np.random.default_rng(4).integers(0, 200, len(b)).astype(np.uint32))
    adv = np.array([[0, 0, 0, 64, 64, 64], [1, 1, 1, 2, 2, 2], [1, 1, 1, 2, 2, 2], [5, 5, 5, 5, 5, 5],
                    [np.nan, 1, 1, 2, 2, 2], [63.5, 63.5, 63.5, 64, 64, 64], [70, 1, 1, 71, 2, 2], [3, 3, 3, 2, 2, 2]], np.float32)
    dump("scene_adversarial", 2, 0, ([0.0] * 3, [64.0] * 3), adv, np.array([7, 7, 9, 3, 4, 5, 6, 8], np.uint32))


if __name__ == "__main__":
    main()
