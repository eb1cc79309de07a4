"""Host-side cost of the Python mirror's per-frame calls (what sits between two frames' kernels when the loop is Python):
   PYTHONPATH=. python tools/host_overhead.py"""
import time, numpy as np, torch, ctypes as C
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes

n = 1000000
sysb, b, ids = scenes.gen_boxes(n, seed=0)
dev = torch.device("cuda", 0)
db = torch.from_numpy(b).to(dev); di = torch.from_numpy(ids.view(np.int32)).to(dev)
lay = bp.Layer(bp.Index64_3D, 0)
for _ in range(5):
    lay.clear(); lay.extend(sysb, db, di); lay.par_sort(); out = lay.scan_device()
torch.cuda.synchronize()


def t(label, fn, k=2000):
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    print("%-28s %.2f us" % (label, (time.perf_counter() - t0) / k * 1e6))


t("clear()", lay.clear)
t("_fence(bounds, ids)", lambda: lay._fence(db, di))
t("torch current_stream", lambda: torch.cuda.current_stream(db.device).cuda_stream)
fvec = C.c_float * 3
t("2 x fvec", lambda: (fvec(*sysb[0]), fvec(*sysb[1])))
t("tensor checks", lambda: (int(di.numel()), int(db.numel()), db.element_size(), di.element_size(), db.is_contiguous(), di.is_contiguous(), db.data_ptr(), di.data_ptr()))
t("par_sort() when sorted", lay.par_sort)
# whole frames, host time per call
K = 200
acc = {"clear": 0.0, "extend": 0.0, "par_sort": 0.0, "scan_device": 0.0}
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    a = time.perf_counter(); lay.clear(); b_ = time.perf_counter(); lay.extend(sysb, db, di); c = time.perf_counter(); lay.par_sort(); d = time.perf_counter()
    out = lay.scan_device(); e = time.perf_counter()
    acc["clear"] += b_ - a; acc["extend"] += c - b_; acc["par_sort"] += d - c; acc["scan_device"] += e - d
torch.cuda.synchronize()
print("frame %.1f us; host time inside: %s" % ((time.perf_counter() - t0) / K * 1e6, {k: round(v / K * 1e6, 1) for k, v in acc.items()}))
