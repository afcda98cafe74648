import os, sys
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,"3d-semantic-segmentation_amd"))
os.environ["VOXPROJ_HEAVY_T"]="100000000"
import numpy as np, torch, voxproj_host
from synthetic_scene import make_scene, make_features_np
dev=torch.device("cuda:0")
s = make_scene(2000, 11, 48, 32, seed=41, room=(5.0, 4.0, 2.4))
C=16
feats = make_features_np(11, 32, 48, C, seed=41)
n_rows=s.n_vox+1
feats_t = torch.from_numpy(feats[None]).to(dev)
occ_t = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w_t = torch.from_numpy(s.c2w).to(dev); intr_t = torch.from_numpy(s.intr[None]).to(dev)
opts=[float(v) for v in s.opts()]; origin=[float(v) for v in s.grid_origin]
def run(splits, pipeline):
    count_t = torch.zeros(n_rows, dtype=torch.int32, device=dev); out_t = torch.zeros(n_rows, C, device=dev)
    ws = voxproj_host.Workspace()
    vm=[c2w_t[a:b].reshape(-1).contiguous() for a,b in splits]
    tot=[]
    for (a,b),v in zip(splits,vm):
        voxproj_host.project_features_raw(feats_t[:, a:b], occ_t, v, intr_t, opts, count_t, out_t, origin, s.voxel_size, workspace=ws, sync=not pipeline, pipeline=pipeline)
        torch.cuda.synchronize(); tot.append(int(count_t.sum()))
    if pipeline: voxproj_host.workspace_status(ws, dev)
    tot.append(int(count_t.sum()))
    return tot
sp=[(0,4),(4,8),(8,9),(9,11),(0,3),(3,11)]
print("plain   ", run(sp, False))
print("pipeline", run(sp, True))
