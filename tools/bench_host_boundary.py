"""The PCIe-inclusive frame: host buffers in (numpy), host pair list out, as a Rust caller of the C ABI
would see it (GPU only).  usage: python tools/bench_host_boundary.py [objects]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sysb, bounds, ids = scenes.gen_boxes(n, seed=0)
lay = bp.Layer(bp.Index64_3D, 0)
for frame in range(6):
    t0 = time.perf_counter()
    lay.clear()
    lay.extend(sysb, bounds, ids)
    t1 = time.perf_counter()
    lay.par_sort()
    t2 = time.perf_counter()
    pairs = lay.par_scan()
    t3 = time.perf_counter()
    print("frame %d: extend(host) %.2f ms  par_sort %.2f ms  par_scan(host) %.2f ms  total %.2f ms  %.2f Gpairs/s  (%d pairs)"
          % (frame, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, len(pairs) / (t3 - t0) / 1e9, len(pairs)),
          flush=True)
