"""world = 1 rehearsal of the C sharded engine with the shard layer's kernel classes timed (where does the shard stage go?)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from broadphase_rs_amd import scenes, sharded
import broadphase_rs_amd as bp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sysb, b, ids = scenes.gen_boxes(n, seed=0)
db, di = torch.from_numpy(b).cuda(), torch.from_numpy(ids.view(np.int32)).cuda()
g = sharded.Group.rccl(0, 1, sharded.Group.unique_id(), 0)
sh = sharded.CShardedLayer(g, 0, bp.Index64_3D, device=0, id_count=n)
for f in range(6):
    if f == 4:
        sh.shard_layer.profile_enable(True); sh.shard_layer.profile_reset()
    sh.clear(); sh.extend(sysb, db, di); out = sh.par_scan_device()
torch.cuda.synchronize()
print(len(out), sh.last["stage_ms"], "round trips: layers", sh.last["round_trips_layers"], "exchanges", sh.last["round_trips_exchanges"])
for k, (ms, cnt) in sh.shard_layer.profile().items():
    if cnt:
        print("%-14s %8.3f ms/frame  %5.1f regions/frame" % (k, ms / 2, cnt / 2))
print(sh.shard_layer.stats())
