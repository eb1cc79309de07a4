"""Per-kernel-class times of the steady frame on the mixed-scale scene (configs[4]'s per-GPU share: power-law sizes), from the layer's
own event profile.   python tools/mixed_micro.py [n_objects] [alpha]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
dev = torch.device("cuda:0")
sysb, bounds, ids = scenes.gen_boxes_powerlaw(n, seed=0, alpha=alpha)
d_bounds = torch.from_numpy(bounds).to(dev)
d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
L = bp.LayerBuilder.new().with_min_depth(0).with_device(0).build(bp.Index64_3D)


def step():
    L.clear(); L.extend(sysb, d_bounds, d_ids); L.par_sort()
    return L.scan_device()


for _ in range(4):
    out = step()
torch.cuda.synchronize()
K = 8
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
st = L.stats()
print("mixed-scale frame %.3f ms wall, %d entries, %d pairs; %s" % (wall, len(L), len(out), {k: st[k] for k in ("tree_sorts_compact", "tree_sorts_in_cells", "sorts_in_cells", "sorts_ahead", "tree_sort_passes", "pair_sort_passes") if k in st}))
print("  " + "  ".join("%s %.3f" % (k, ms / K) for k, (ms, cnt) in sorted(prof.items()) if ms / K > 0.004))
