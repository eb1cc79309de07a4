"""Frame latency of clear + extend + par_sort + par_scan for small scenes (GPU only): ms per frame,
host round trips per frame.  usage: python tools/bench_small.py [n ...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes


def run(n, frames=200):
    sysb, bounds, ids = scenes.gen_boxes(n, seed=1)
    d_b = torch.from_numpy(bounds).cuda()
    d_i = torch.from_numpy(ids.view(np.int32)).cuda()
    lay = bp.Layer(bp.Index64_3D, 0)
    for _ in range(5):
        lay.clear(); lay.extend(sysb, d_b, d_i); lay.par_sort(); out = lay.scan_device()
    s0 = lay.stats()["host_syncs"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per = []
    for _ in range(frames):
        t1 = time.perf_counter()
        lay.clear(); lay.extend(sysb, d_b, d_i); lay.par_sort(); out = lay.scan_device()
        per.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    per = np.array(per) * 1e3
    print("    per-frame ms: median %.3f  p90 %.3f  max %.3f" % (np.median(per), np.percentile(per, 90), per.max()))
    syncs = (lay.stats()["host_syncs"] - s0) / frames
    print("n=%8d  pairs=%9d  %.3f ms/frame  %.1f host syncs/frame  %.2f Mpairs/s" % (n, len(out), dt * 1e3, syncs, len(out) / dt / 1e6),
          flush=True)
    lay.profile_enable(True); lay.profile_reset()
    for _ in range(20):
        lay.clear(); lay.extend(sysb, d_b, d_i); lay.par_sort(); out = lay.scan_device()
    st = lay.stats()
    print("    " + "  ".join("%s %.1fus/%d" % (k, v[0] / 20 * 1e3, v[1] // 20) for k, v in lay.profile().items() if v[1]),
          " passes tree %d pair %d" % (st["tree_sort_passes"], st["pair_sort_passes"]), flush=True)


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [1000, 10000, 100000, 1000000]:
        run(n)
