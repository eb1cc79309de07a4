"""Diagnostic: where a scanning wave's time goes (a library built with -DSX_STAMPS: tools/ab.sh build stamps "-DSX_STAMPS").
   BP_LIBRARY=$PWD/broadphase_rs_amd/lib_ab/stamps/libbroadphase_amd.so python tools/sx_stamps.py [n_objects]"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes, layer as _layer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
lib = ctypes.CDLL(os.environ["BP_LIBRARY"])
lib.bp_debug_sx_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
dev = torch.device("cuda:0")
sysb, bounds, ids = scenes.gen_boxes_powerlaw(n, seed=0, alpha=2.5) if os.environ.get("SX_POWERLAW") else scenes.gen_boxes(n, seed=1)
d_bounds = torch.from_numpy(bounds).to(dev)
d_ids = torch.from_numpy(ids.view(np.int32)).to(dev)
L = bp.LayerBuilder.new().with_min_depth(0).with_device(0).build(bp.Index64_3D)
def step():
    L.clear(); L.extend(sysb, d_bounds, d_ids); L.par_sort(); return L.scan_device()
for _ in range(4): step()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
assert lib.bp_debug_sx_stamps(buf, 1) == 0
K = 3
for _ in range(K): out = step()
torch.cuda.synchronize()
assert lib.bp_debug_sx_stamps(buf, 0) == 0
v = [x / K for x in buf]
tot = v[7]
names = ["wait for the chunk's loads + windows to LDS", "range ends (bisect, far ends)", "scan + owner records", "candidate steps (all)", "  of which push2 (stage, hist, block writes)"]
print(f"pairs {len(out)}  chunks {v[5]:.0f}  steps {v[6]:.0f}  ({v[6] / max(v[5], 1):.2f} per chunk)  wave-ticks total {tot:.3e}")
for q, nm in enumerate(names): print(f"{nm:55s} {100 * v[q] / tot:5.1f} %   {v[q] / max(v[5], 1):8.1f} ticks per chunk")
print(f"chunk loop of the slowest wave {buf[8]:.3e}, of the fastest {(~buf[9]) & (2**64 - 1):.3e}, mean {tot / (256 * 5 * 4):.3e} ticks (max / min of the LAST frame)")
print(f"{'outside the chunk loop (prologue, finish, hist flush)':55s} {100 * (tot - v[0] - v[1] - v[2] - v[3]) / tot:5.1f} %")

lib.bp_debug_sx_waves.argtypes = [ctypes.c_void_p]
wv = np.zeros((8192, 4), dtype=np.uint64)
assert lib.bp_debug_sx_waves(wv.ctypes.data) == 0
wv = wv[wv[:, 0] > 0]
t = wv[:, 0].astype(np.float64); chunks = (wv[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.float64); steps = (wv[:, 1] >> np.uint64(32)).astype(np.float64)
hw = wv[:, 2] & np.uint64(0xFFFFFFFF); xcc = (wv[:, 2] >> np.uint64(32)) & np.uint64(0xF)
cu = (hw >> np.uint64(8)) & np.uint64(0xF); se = (hw >> np.uint64(13)) & np.uint64(0x7); simd = (hw >> np.uint64(4)) & np.uint64(3)
start = wv[:, 3].astype(np.float64); start -= start.min()
print(f"waves {len(t)}: loop ticks min {t.min():.3e} p10 {np.percentile(t, 10):.3e} median {np.median(t):.3e} p90 {np.percentile(t, 90):.3e} max {t.max():.3e}  mean {t.mean():.3e}")
print(f"steps per wave min {steps.min():.0f} median {np.median(steps):.0f} max {steps.max():.0f}; corr(ticks, steps) {np.corrcoef(t, steps)[0, 1]:.3f}; start spread max {start.max():.3e}")
end = start + t
print(f"end tick: min {end.min():.3e} median {np.median(end):.3e} max {end.max():.3e}")
for x in range(8):
    m = xcc == x
    if m.any(): print(f"  xcc {x}: waves {m.sum():5d}  mean ticks {t[m].mean():.3e}  max {t[m].max():.3e}  distinct (se, cu) {len(set(zip(se[m].tolist(), cu[m].tolist())))}")
key = xcc * np.uint64(1000) + se * np.uint64(100) + cu
u, cnt = np.unique(key, return_counts=True)
print("waves per CU: min", cnt.min(), "max", cnt.max(), "CUs", len(u))
percu = np.array([t[key == k].mean() for k in u])
print(f"per-CU mean ticks: min {percu.min():.3e} max {percu.max():.3e}; by waves on the CU:", {int(c): float(np.round(percu[cnt == c].mean())) for c in np.unique(cnt)})
