"""What of the march is NOT the sample loop: k_first_hit + k_worklist per 60-view call on the R2 scene, on an empty grid of the
same shape (every ray leaps to its end in a few evaluations: ray set-up, image store, launch) and on a grid whose only voxel sits
in a far corner.  python3 tools/march_fixed_cost.py"""
import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np, torch, voxproj_host
from synthetic_scene import make_scene
dev = torch.device("cuda", 0)
n_vox, n_views, W, H, C, V = 200000, 300, 968, 548, 8, 60
s = make_scene(n_vox, n_views, W, H, seed=0)
feats = torch.zeros(1, V, H, W, C, device=dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
vmi = torch.from_numpy(s.c2w[:V]).reshape(-1).contiguous().to(dev)
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev); out = torch.zeros(n_vox + 1, C, device=dev)
for name, occ_np in (("scene", s.occ), ("empty grid", np.zeros_like(s.occ)), ("one far corner voxel", None)):
    if occ_np is None:
        occ_np = np.zeros_like(s.occ); occ_np[0, 0, 0] = 1
    occ = torch.from_numpy(occ_np[None].astype(np.int64)).to(dev)
    ws = voxproj_host.Workspace()
    for rep in range(4):
        if rep == 1:
            voxproj_host.profile_enable(True)          # (enabling resets the recorded spans: once, after the warm-up call)
        voxproj_host.project_features_raw(feats, occ, vmi, intr, [float(v) for v in s.opts()], count, out, [float(v) for v in s.grid_origin], s.voxel_size, workspace=ws, sync=True)
    p = voxproj_host.profile_read(); voxproj_host.profile_enable(False)
    print(f"{name:22s} march+worklist {p['first_hit_ms'] / 3:.3f} ms per {V}-view call  (x5 = {p['first_hit_ms'] / 3 * 5:.2f} ms per 300 views)  gather {p['gather_ms'] / 3:.3f}")
    ws.release()
