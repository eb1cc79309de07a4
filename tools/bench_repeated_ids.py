"""Frame time of a 10 M-object scene whose raw pair stream cannot be deduplicated at the source:
    python tools/bench_repeated_ids.py        every id on two objects (the reference's skip rule can fire)
    python tools/bench_repeated_ids.py big    distinct ids >= 2^27 (no room for the box codes in the id payload)
One JSON line: ms per frame (clear + extend + par_sort + par_scan, device-resident), pairs, raw pairs, per-class kernel ms."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes

n = 10_000_000
big = "big" in sys.argv[1:]
sysb, b, ids = scenes.gen_boxes(n, seed=0)
ids = (ids + np.uint32(1 << 27)) if big else (ids // 2).astype(np.uint32)
dev = torch.device("cuda", 0)
db = torch.from_numpy(b).to(dev)
di = torch.from_numpy(ids.view(np.int32)).to(dev)
lay = bp.Layer(bp.Index64_3D, 0)


def step():
    lay.clear(); lay.extend(sysb, db, di); lay.par_sort()
    return lay.scan_device()


for _ in range(3):
    out = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
lay.profile_reset(); lay.profile_enable(True); out = step(); torch.cuda.synchronize(); lay.profile_enable(False)
st = lay.stats()
print(json.dumps({"scene": "ids >= 2^27" if big else "two objects per id", "ms": round(dt * 1e3, 3), "pairs": len(out), "raw": st["raw_pairs"],
                  "pair_sort_passes": st["pair_sort_passes"], "flags": st["flags"],
                  "kernel_ms": {k: round(v[0], 3) for k, v in lay.profile().items() if v[1]}}))
