import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np, torch, voxproj_host
from synthetic_scene import make_scene, make_features_torch
dev = torch.device("cuda", 0)
for name, (n_vox, n_views, W, H) in {"R2": (200000, 300, 968, 548), "R1": (80000, 100, 484, 274)}.items():
    s = make_scene(n_vox, n_views, W, H, seed=0)
    occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
    intr = torch.from_numpy(s.intr[None]).to(dev)
    vm = [torch.from_numpy(s.c2w[v]).reshape(-1).contiguous().to(dev) for v in range(16)]
    for C in (8, 512):
        feats = torch.empty(16, H, W, C, device=dev); make_features_torch(16, H, W, C, dev, seed=0, out=feats)
        count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev); out = torch.zeros(n_vox + 1, C, device=dev)
        ws = voxproj_host.Workspace()
        for rep in range(3):
            if rep == 1: voxproj_host.profile_enable(True)
            for v in range(16):
                voxproj_host.project_features_raw(feats[v][None, None], occ, vm[v], intr, [float(x) for x in s.opts()], count, out, [float(x) for x in s.grid_origin], s.voxel_size, workspace=ws, sync=True)
        p = voxproj_host.profile_read(); voxproj_host.profile_enable(False)
        print(f"{name} C={C:3d}: march+worklist {p['first_hit_ms'] / 32 * 1e3:.1f} us per one-view call, gather {p['gather_ms'] / 32 * 1e3:.1f} us")
        ws.release()
