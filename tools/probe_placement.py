"""Does the gather's speed depend on where its buffers sit?  One process, same views, the feature pool / outputs /
workspace re-allocated behind dummy allocations of different sizes; prints the mean k_gather launch time for each.
(run on the GPU box: python tools/probe_placement.py)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_features_torch, make_scene  # noqa: E402

dev = torch.device("cuda", 0)
n_vox, n_views, W, H, C = 200000, 300, 968, 548, 512
V, NCALL = 16, 8
s = make_scene(n_vox, n_views, W, H, seed=0)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
vmis = [c2w[i * V:(i + 1) * V].reshape(-1).contiguous() for i in range(NCALL)]


def run(tag, pad_mb, what):
    torch.cuda.empty_cache()
    pads = []
    def pad():
        if pad_mb:
            pads.append(torch.empty(int(pad_mb * (1 << 20)), dtype=torch.uint8, device=dev))
    if "feats" in what:
        pad()
    feats = torch.empty((1, V, H, W, C), dtype=torch.float32, device=dev)
    make_features_torch(V, H, W, C, dev, seed=0, out=feats[0])
    if "out" in what:
        pad()
    count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
    out = torch.zeros(n_vox + 1, C, dtype=torch.float32, device=dev)
    if "ws" in what:
        pad()
    ws = voxproj_host.Workspace()
    for rep in range(2):
        if rep == 1:
            voxproj_host.profile_enable(True)
        for ci in range(NCALL):
            voxproj_host.project_features_raw(feats, occ, vmis[ci], intr, opts, count, out, origin, s.voxel_size,
                                              workspace=ws, sync=False, reuse_accel=(ci > 0 or None))
        torch.cuda.synchronize()
    p = voxproj_host.profile_read()
    voxproj_host.profile_enable(False)
    print(f"{tag:28s} feats@{feats.data_ptr():#x} out@{out.data_ptr():#x} ws@{ws.ptr():#x}  "
          f"gather {p['gather_ms'] / max(p['gather_launches'], 1):.3f} ms/launch  march {p['first_hit_ms'] / NCALL:.3f}", flush=True)
    ws.release()
    del feats, out, count, ws, pads


run("base", 0, ())
run("base again", 0, ())
for mb in (1, 2, 3, 7, 64, 65, 129.5):
    run(f"pad {mb} MB before feats", mb, ("feats",))
for mb in (1, 2, 3, 7, 64.25):
    run(f"pad {mb} MB before out", mb, ("out",))
for mb in (1, 3, 64.25):
    run(f"pad {mb} MB before ws", mb, ("ws",))
