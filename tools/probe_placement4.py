"""Per-slot gather time (V = 1, the SAME camera for every slot, so identical work) across the 32 one-GB slots of the
feature pool, for a few re-allocations: is a slow placement slow everywhere or only in some physical regions?"""
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
POOL = 32
s = make_scene(n_vox, n_views, W, H, seed=0)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
out = torch.zeros(n_vox + 1, C, dtype=torch.float32, device=dev)
ws = voxproj_host.Workspace()
vm = c2w[0:1].reshape(-1).contiguous()

for rnd in range(5):
    feats = None
    torch.cuda.empty_cache()
    feats = torch.empty((1, POOL, H, W, C), dtype=torch.float32, device=dev)
    make_features_torch(POOL, H, W, C, dev, seed=0, out=feats[0])
    res = []
    for sl in range(POOL):
        for rep in range(4):
            voxproj_host.profile_enable(rep > 0)
            voxproj_host.project_features_raw(feats[:, sl:sl + 1], occ, vm, intr, opts, count, out, origin,
                                              s.voxel_size, workspace=ws, sync=False, reuse_accel=(rnd + sl + rep > 0 or None))
            torch.cuda.synchronize()
        p = voxproj_host.profile_read()
        voxproj_host.profile_enable(False)
        res.append(p["gather_ms"] / 3 * 1e3)
    print(f"alloc {rnd}: mean {np.mean(res):.1f} us  per slot: " + " ".join(f"{r:.0f}" for r in res), flush=True)
