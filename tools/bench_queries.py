#!/usr/bin/env python3
"""Batched Layer::test_box throughput: Q random boxes against the sorted tree of N uniform
objects, device-resident inputs.  (The CPU figure quoted in DESIGN.md comes from tests: the oracle is test
infrastructure and is not used by tools.)
usage: python tools/bench_queries.py [--objects N] [--queries Q] [--steps K]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--half-extent", type=float, default=10.0, help="query boxes are centre +- U[0.5, 1] * this")
    args = ap.parse_args()
    import torch
    import broadphase_rs_amd as bp
    from broadphase_rs_amd import scenes
    dev = torch.device("cuda", 0)
    sysb, bounds, ids = scenes.gen_boxes(args.objects, seed=0)
    rng = np.random.default_rng(1)
    L = float(sysb[1][0])
    c = rng.uniform(0, L, size=(args.queries, 3))
    h = rng.uniform(0.5, 1.0, size=(args.queries, 3)) * args.half_extent
    boxes = np.concatenate([c - h, c + h], axis=1).astype(np.float32)
    layer = bp.LayerBuilder.new().with_min_depth(0).build(bp.Index64_3D)
    layer.extend(sysb, torch.from_numpy(bounds).to(dev), torch.from_numpy(ids.view(np.int32)).to(dev))
    layer.par_sort()
    d_boxes = torch.from_numpy(boxes).to(dev)
    rows = layer.test_box_batch_device(sysb, d_boxes)          # warm-up (allocations)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = layer.test_box_batch_device(sysb, d_boxes)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    n_rows = len(rows)
    out = {"metric": "batched test_box queries/s", "objects": args.objects, "queries": args.queries, "rows": n_rows,
           "rows_per_query": n_rows / args.queries, "ms_per_batch": dt * 1e3, "queries_per_s": args.queries / dt,
           "rows_per_s": n_rows / dt, "note": "tree resident and sorted; includes the (query, id) sort + dedup"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
