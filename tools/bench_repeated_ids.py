import sys, time, json
sys.path.insert(0,'/root/repo')
import numpy as np, torch
import broadphase_rs_amd as bp
from broadphase_rs_amd import scenes
n=10_000_000
sysb,b,ids=scenes.gen_boxes(n,seed=0)
ids=(ids//2).astype(np.uint32)          # every id owns two boxes: the skip rule can fire
dev=torch.device("cuda",0)
db=torch.from_numpy(b).to(dev); di=torch.from_numpy(ids.view(np.int32)).to(dev)
lay=bp.Layer(bp.Index64_3D,0)
def step():
    lay.clear(); lay.extend(sysb,db,di); lay.par_sort(); return lay.scan_device()
for _ in range(2): out=step()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(3): out=step()
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/3
lay.profile_reset(); lay.profile_enable(True); out=step(); torch.cuda.synchronize(); lay.profile_enable(False)
st=lay.stats()
print(json.dumps({"ms":dt*1e3,"pairs":len(out),"raw":st["raw_pairs"],"flags":st["flags"],"kernel_ms":{k:round(v[0],3) for k,v in lay.profile().items()}}))
