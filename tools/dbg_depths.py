import numpy as np, broadphase_rs_amd as bp
from broadphase_rs_amd import scenes
for n in (1000000, 10000000):
    sysb, b, ids = scenes.gen_boxes(n, seed=0)
    L = bp.Layer(bp.Index64_3D, 0)
    L.extend(sysb, b, ids)
    st = L.stats(); f = L.get_facts()
    k, i = L.tree()
    d = (k & 31).astype(np.int64)
    print(n, bin(st["depth_mask"]), np.bincount(d), hex(int(f["key_or"]) if isinstance(f, dict) else 0))
