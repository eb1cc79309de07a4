#!/usr/bin/env python3
"""Mixed-scale scene (BASELINE.json configs[4] shape on one GPU): step time and per-class kernel times.
usage: python tools/bench_powerlaw.py [--objects N] [--alpha A] [--steps K]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", type=int, default=2_000_000)
    ap.add_argument("--alpha", type=float, default=2.5)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import broadphase_rs_amd as bp
    from broadphase_rs_amd import scenes
    dev = torch.device("cuda", 0)
    sysb, bounds, ids = scenes.gen_boxes_powerlaw(args.objects, seed=0, alpha=args.alpha)
    d_b = torch.from_numpy(bounds).to(dev); d_i = torch.from_numpy(ids.view(np.int32)).to(dev)
    layer = bp.LayerBuilder.new().with_min_depth(0).build(bp.Index64_3D)

    def step():
        layer.clear(); layer.extend(sysb, d_b, d_i); layer.par_sort()
        return layer.scan_device()
    for _ in range(2):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    layer.profile_reset(); layer.profile_enable(True)
    out = step(); torch.cuda.synchronize()
    layer.profile_enable(False)
    st = layer.stats()
    print(json.dumps({"objects": args.objects, "alpha": args.alpha, "ms_per_step": dt * 1e3, "pairs": len(out),
                      "pairs_per_s": len(out) / dt, "entries": st["entries"], "raw_pairs": st["raw_pairs"],
                      "raw_candidates": st["raw_candidates"], "pair_sort_passes": st["pair_sort_passes"],
                      "kernel_ms": {k: round(v[0], 3) for k, v in layer.profile().items()}}))


if __name__ == "__main__":
    main()
