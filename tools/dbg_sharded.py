import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes, sharded
g = sharded.Group.local(1)
sysb, bounds, ids = scenes.gen_boxes(30000, seed=3)
sh = sharded.CShardedLayer(g, 0, bp.Index64_3D, device=0)
sh.clear(); sh.extend(sysb, bounds, ids)
out = sh.par_scan()
print(len(out), sh.last, sh.shard_layer.stats())
lay = bp.Layer(bp.Index64_3D, 0); lay.extend(sysb, bounds, ids); print(len(lay.par_scan()))
