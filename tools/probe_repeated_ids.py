"""frame times of a 10 M-object scene whose ids repeat (every id on two objects; `big`: distinct ids >= 2^27, which
cannot carry box codes): the duplicate-tolerant pair route"""
import numpy as np, time, sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from broadphase_rs_amd import scenes
from broadphase_rs_amd.layer import Layer, Index64_3D
import torch
n = 10_000_000
sysb, bounds, ids = scenes.gen_boxes(n, seed=1)
ids2 = (np.arange(n, dtype=np.uint32) >> 1) if "big" not in sys.argv else (np.arange(n, dtype=np.uint32) + np.uint32(1 << 27))
L = Layer(Index64_3D)
db = torch.from_numpy(bounds).cuda(); di = torch.from_numpy(ids2).cuda()
for f in range(6):
    L.clear(); L.extend(sysb, db, di)
    torch.cuda.synchronize(); t = time.perf_counter()
    p = L.scan_device(); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    st = L.stats()
    print(f, "scan ms %.3f" % (dt * 1e3), len(p), st["raw_pairs"], st["pair_sort_passes"], flush=True)
L.profile_enable(True); L.profile_reset()
for f in range(4):
    L.clear(); L.extend(sysb, db, di); L.scan_device()
torch.cuda.synchronize()
for k, (ms, cnt) in L.profile().items():
    if cnt:
        print("%-14s %8.3f ms/frame  %5.1f regions/frame" % (k, ms / 4, cnt / 4))
