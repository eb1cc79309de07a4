"""Steady-state frames of one scene for a timeline trace:
    rocprofv3 --kernel-trace --memory-copy-trace -d gpurun_out/trace -o t -- python3 tools/trace_frame.py 1000000
then tools/trace_show.py gpurun_out/trace prints one frame's kernels / copies in start order with the gaps between them."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from broadphase_rs_amd import scenes
from broadphase_rs_amd.layer import Layer, Index64_3D

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sysb, bounds, ids = scenes.gen_boxes(n, seed=0)
db, di = torch.from_numpy(bounds).cuda(), torch.from_numpy(ids).cuda()
L = Layer(Index64_3D)
for f in range(frames):
    L.clear(); L.extend(sysb, db, di); L.scan_device()
torch.cuda.synchronize()
