"""Per-kernel-class times of the fused narrow phase frame (clear + extend + par_sort + scan_filtered_aabb_device on the headline
scene), from the layer's own event profile.   python tools/filt_micro.py [n_objects]
A/B: BP_LIBRARY=<variant .so>."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import time
import numpy as np
import torch
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
sysb, bounds, ids = scenes.gen_boxes(n, seed=1)
d_bounds = torch.from_numpy(bounds).to(dev)
d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
L = bp.LayerBuilder.new().with_min_depth(0).with_device(0).build(bp.Index64_3D)


def step(filtered=True):
    L.clear(); L.extend(sysb, d_bounds, d_ids); L.par_sort()
    return L.scan_filtered_aabb_device(d_bounds) if filtered else L.scan_device()


for _ in range(4):
    out = step()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    out = step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / K * 1e3
L.profile_enable(True); L.profile_reset()
for _ in range(K):
    out = step()
torch.cuda.synchronize()
prof = L.profile()
print("filtered frame %.3f ms wall, %d pairs" % (wall, len(out)))
print("  " + "  ".join("%s %.3f" % (k, ms / K) for k, (ms, cnt) in sorted(prof.items()) if ms / K > 0.004))
