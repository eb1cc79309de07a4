#!/usr/bin/env python3
"""Size-independent property check at sizes the oracle cannot follow (needs a GPU):
   python tools/check_large.py <objects> [powerlaw]  -> sorted tree, strictly increasing pair list, idempotent scan, and a
   checksum line: run it again with BP_NO_EXPAND_SCAN=1 BP_NO_GROUP_FINISH=1 BP_NO_SEGMENT_FINISH=1 (per-entry ancestor walk,
   full-width sorts: other kernels end to end) and compare"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes

n = int(sys.argv[1])
t0 = time.time()
powerlaw = len(sys.argv) > 2 and sys.argv[2] == "powerlaw"   # BASELINE configs[4]: Pareto(alpha = 2.5) sizes, deep stacks
sysb, bounds, ids = scenes.gen_boxes_powerlaw(n, seed=1, alpha=2.5) if powerlaw else scenes.gen_boxes(n, seed=1)
print("generated %d objects in %.1fs" % (n, time.time() - t0), flush=True)
tb, ti = torch.from_numpy(bounds).cuda(), torch.from_numpy(ids.view(np.int32)).cuda()
del bounds
layer = bp.Layer(bp.Index64_3D, 0)
layer.extend(sysb, tb, ti)
layer.par_sort()
k, i = layer.tree_device()
kt, it = k.as_torch(), i.as_torch()
ok_tree = bool(((kt[1:] > kt[:-1]) | ((kt[1:] == kt[:-1]) & (it[1:] >= it[:-1]))).all().item())
m = kt.numel()
del kt, it
torch.cuda.synchronize()
t0 = time.time()
out = layer.scan_device()
dt = time.time() - t0
p = out.as_torch()                      # (P, 2) int32 view: id, id_
a, b = p[:, 0].to(torch.int64) & 0xFFFFFFFF, p[:, 1].to(torch.int64) & 0xFFFFFFFF
packed = (a << 32) | b
ok_pairs = bool((packed[1:] > packed[:-1]).all().item())
ok_self = bool((a != b).all().item())
st = layer.stats()
checksum = int(packed.sum().item() & 0x7FFFFFFFFFFFFFFF)
del packed, a, b, p
out2 = layer.scan_device()
p2 = out2.as_torch()
a2, b2 = p2[:, 0].to(torch.int64) & 0xFFFFFFFF, p2[:, 1].to(torch.int64) & 0xFFFFFFFF
checksum2 = int((((a2 << 32) | b2).sum().item()) & 0x7FFFFFFFFFFFFFFF)
print("objects %d entries %d raw %d pairs %d scan %.1f ms | tree sorted %s, pairs strictly increasing %s, no self pairs %s, "
      "idempotent %s, tree passes %d pair passes %d" % (n, m, st["raw_pairs"], st["pairs"], dt * 1e3, ok_tree, ok_pairs, ok_self,
                                                    checksum == checksum2 and len(out) == len(out2), st["tree_sort_passes"],
                                                    st["pair_sort_passes"]))
print("CHECKSUM objects %d%s pairs %d sum %d" % (n, " powerlaw" if powerlaw else "", st["pairs"], checksum))
assert ok_tree and ok_pairs and ok_self and checksum == checksum2
