"""kernel classes of a COLD frame (buffers allocated, plan hints forgotten = the plan of a layer's first frame) next to a steady one"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from broadphase_rs_amd import scenes
from broadphase_rs_amd.layer import Layer, Index64_3D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sysb, b, ids = scenes.gen_boxes(n, seed=0)
db, di = torch.from_numpy(b).cuda(), torch.from_numpy(ids.view(np.int32)).cuda()
L = Layer(Index64_3D)


def frame():
    L.clear(); L.extend(sysb, db, di); L.par_sort(); return L.scan_device()


for _ in range(3):
    frame()
for label, cold in (("steady", False), ("cold", True)):
    if cold:
        L.reset_plan_hints()
    torch.cuda.synchronize(); t0 = time.perf_counter(); frame(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if cold:
        L.reset_plan_hints()
    L.profile_reset(); L.profile_enable(True); frame(); torch.cuda.synchronize(); L.profile_enable(False)
    st = L.stats()
    print(label, "%.3f ms" % (dt * 1e3), "tree passes", st["tree_sort_passes"], "pair passes", st["pair_sort_passes"],
          {k: round(v[0], 3) for k, v in L.profile().items() if v[1]})
    for _ in range(2):
        frame()
