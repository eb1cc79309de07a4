import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
scale0 = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
step = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
sysb, bounds, ids = scenes.gen_boxes(n, seed=81)
ctr = (bounds[:, :3] + bounds[:, 3:]) * np.float32(0.5)
half = (bounds[:, 3:] - bounds[:, :3]) * np.float32(0.5)
side = np.float32(sysb[1][0])
L = bp.Layer(bp.Index64_3D, 0)
for f in range(6):
    h = half * np.float32(scale0 + step * f)
    cur = np.ascontiguousarray(np.concatenate([np.maximum(ctr - h, np.float32(0)), np.minimum(ctr + h, side)], axis=1).astype(np.float32))
    L.clear(); L.extend(sysb, torch.from_numpy(cur).cuda(), torch.from_numpy(ids.view(np.int32)).cuda()); out = L.scan_device()
    st = L.stats()
    print(f, len(out), {k: st[k] for k in ("sorts_ahead", "sorts_ahead_missed", "sorts_in_cells", "tree_sorts_in_cells", "sort_repairs", "tree_sort_passes", "tree_plan_flips", "entries")}, flush=True)
