#!/usr/bin/env python3
"""Pair count of the union scene that `bench.py --gpus W --objects N` shards over W ranks, on ONE GPU (cross-check
of a multi-rank run's config.pairs_total).  usage: python tools/check_shard_union.py <objects_per_rank> <world>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes

n, world = int(sys.argv[1]), int(sys.argv[2])
parts = [scenes.gen_boxes_shard(n * world, r, world, seed=0) for r in range(world)]
sysb = parts[0][0]
bounds = np.concatenate([p[1] for p in parts]); ids = np.concatenate([p[2] for p in parts])
lay = bp.Layer(bp.Index64_3D, 0)
lay.extend(sysb, bounds, ids)
print("objects", len(ids), "pairs", len(lay.scan_device()))
