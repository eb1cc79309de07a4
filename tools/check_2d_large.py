#!/usr/bin/env python3
"""2-D index kinds at scale (GPU only): step time + size-independent properties (sorted tree, strictly increasing
pair list, idempotent scan).  usage: python tools/check_2d_large.py [objects]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import broadphase_rs_amd as bp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(0)
L = np.float32(np.sqrt(n / 0.01) + 5.5)          # 0.01 boxes per unit area, sizes U[1, 10) as in the 3-D recipe
size = rng.uniform(1, 10, size=(n, 2)).astype(np.float32)
lo = (rng.uniform(0, 1, size=(n, 2)).astype(np.float32) * (L - size)).astype(np.float32)
bounds = np.concatenate([lo, lo + size], axis=1).astype(np.float32)
sysb = np.array([[0, 0], [L, L]], dtype=np.float32)
ids = np.arange(n, dtype=np.uint32)
tb, ti = torch.from_numpy(bounds).cuda(), torch.from_numpy(ids.view(np.int32)).cuda()
for kind in (bp.Index64_2D, bp.Index32_2D):
    lay = bp.Layer(kind, 0)
    for _ in range(3):
        lay.clear(); lay.extend(sysb, tb, ti); lay.par_sort(); out = lay.scan_device()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        lay.clear(); lay.extend(sysb, tb, ti); lay.par_sort(); out = lay.scan_device()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    k, i = lay.tree_device()
    kt, it = k.as_torch(), i.as_torch()
    if kt.dtype == torch.int32:                   # Index32_2D keys: compare as unsigned
        kt = kt.to(torch.int64) & 0xFFFFFFFF
    ok_tree = bool(((kt[1:] > kt[:-1]) | ((kt[1:] == kt[:-1]) & (it[1:] >= it[:-1]))).all().item())
    p = out.as_torch()
    a, b = p[:, 0].to(torch.int64) & 0xFFFFFFFF, p[:, 1].to(torch.int64) & 0xFFFFFFFF
    packed = (a << 32) | b
    ok_pairs = bool((packed[1:] > packed[:-1]).all().item()) and bool((a != b).all().item())
    cs = int(packed.sum().item() & 0x7FFFFFFFFFFFFFFF)
    out2 = lay.scan_device(); p2 = out2.as_torch()
    cs2 = int(((((p2[:, 0].to(torch.int64) & 0xFFFFFFFF) << 32) | (p2[:, 1].to(torch.int64) & 0xFFFFFFFF)).sum().item()) & 0x7FFFFFFFFFFFFFFF)
    st = lay.stats()
    print("%s: %d objects %d entries %d pairs  %.2f ms/step  %.2f Gpairs/s | tree sorted %s, pairs strictly increasing %s, idempotent %s, passes %d+%d"
          % (kind.name if hasattr(kind, "name") else kind, n, st["entries"], st["pairs"], dt * 1e3, st["pairs"] / dt / 1e9,
             ok_tree, ok_pairs, cs == cs2, st["tree_sort_passes"], st["pair_sort_passes"]), flush=True)
    assert ok_tree and ok_pairs and cs == cs2
