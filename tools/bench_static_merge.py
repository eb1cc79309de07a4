#!/usr/bin/env python3
"""The crate's per-frame pattern (src/layer.rs:123-138): a pre-sorted static layer merged into each frame's dynamic
layer.  usage: python tools/bench_static_merge.py [--static N] [--dynamic M] [--steps K]   (BP_NO_MERGE_SORT=1 for A/B)"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--static", type=int, default=10_000_000)
    ap.add_argument("--dynamic", type=int, default=100_000)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import broadphase_rs_amd as bp
    from broadphase_rs_amd import scenes
    dev = torch.device("cuda", 0)
    sysb, bounds, ids = scenes.gen_boxes(args.static + args.dynamic, seed=0)
    sb = torch.from_numpy(bounds[: args.static]).to(dev); si = torch.from_numpy(ids[: args.static].view(np.int32)).to(dev)
    db = torch.from_numpy(bounds[args.static:]).to(dev); di = torch.from_numpy(ids[args.static:].view(np.int32)).to(dev)
    static = bp.LayerBuilder.new().build(bp.Index64_3D)
    static.extend(sysb, sb, si); static.par_sort()
    dyn = bp.LayerBuilder.new().build(bp.Index64_3D)

    def frame():
        dyn.clear(); dyn.extend(sysb, db, di); dyn.merge(static); dyn.par_sort()
        return dyn.scan_device()
    for _ in range(2):
        out = frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    dyn.profile_reset(); dyn.profile_enable(True); out = frame(); torch.cuda.synchronize(); dyn.profile_enable(False)
    st = dyn.stats()
    print(json.dumps({"static": args.static, "dynamic": args.dynamic, "merge_sort": os.environ.get("BP_NO_MERGE_SORT") is None,
                      "ms_per_frame": dt * 1e3, "pairs": len(out), "raw_pairs": st["raw_pairs"],
                      "kernel_ms": {k: round(v[0], 3) for k, v in dyn.profile().items()}}))


if __name__ == "__main__":
    main()
