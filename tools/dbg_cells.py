import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sysb, bounds, ids = scenes.gen_boxes(n, seed=61)
db, di = torch.from_numpy(bounds).cuda(), torch.from_numpy(ids.view(np.int32)).cuda()
L = bp.Layer(bp.Index64_3D, 0)
for f in range(5):
    L.clear(); L.extend(sysb, db, di); out = L.scan_device()
    st = L.stats()
    print(f, len(out), st["sorts_ahead"], st["sorts_ahead_missed"], flush=True)
