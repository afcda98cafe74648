import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-semantic-segmentation_amd"))
os.environ["VOXPROJ_DEBUG_EVALS"]="1"
import numpy as np, torch, voxproj_host
from synthetic_scene import make_scene
dev=torch.device("cuda:0")
s=make_scene(200000, 8, 968, 548, seed=0)
V=4; H,W,C=548,968,8
feats=torch.zeros(1,V,H,W,C,device=dev)
occ=torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
cnt=torch.zeros(s.n_vox+1,dtype=torch.int32,device=dev); out=torch.zeros(s.n_vox+1,C,device=dev)
ws=voxproj_host.Workspace()
voxproj_host.project_features_raw(feats, occ, torch.from_numpy(s.c2w[:V]).reshape(-1).to(dev), torch.from_numpy(s.intr[None]).to(dev), [float(v) for v in s.opts()], cnt, out, [float(v) for v in s.grid_origin], s.voxel_size, workspace=ws, sync=True)
h=voxproj_host.hit_image(ws, dev).cpu().numpy()[0]
leap=h>>16; fine=h&0xffff
print("leap mean", leap.mean(), "max", leap.max(), "fine mean", fine.mean(), "max", fine.max())
tot=leap+fine
# per 8x8 tile max
v0=tot[0][:544,:968].reshape(68,8,121,8).max(axis=(1,3))
print("total mean", tot.mean(), "per-wave max mean", v0.mean())
lv=leap[0][:544,:968].reshape(68,8,121,8).max(axis=(1,3)); fv=fine[0][:544,:968].reshape(68,8,121,8).max(axis=(1,3))
print("per-wave max leap", lv.mean(), "fine", fv.mean())
print("hist fine", np.percentile(fine,[10,50,90,99]), "hist leap", np.percentile(leap,[10,50,90,99]))
