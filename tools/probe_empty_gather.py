"""What do the gather's workgroups for UNTOUCHED voxels cost?  The grid is sized for every row (the host does not know how many
voxels a call touches); wavefronts beyond the end of the work list read eight counters and exit.  A view that looks away
from the scene hits nothing: its gather is nothing but such workgroups.  python tools/probe_empty_gather.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_features_torch, make_scene  # noqa: E402

dev = torch.device("cuda", 0)
for name, (n_vox, W, H) in {"R1": (80000, 484, 274), "R2": (200000, 968, 548)}.items():
    C = 512
    s = make_scene(n_vox, 4, W, H, seed=0)
    occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
    intr = torch.from_numpy(s.intr[None]).to(dev)
    opts = [float(v) for v in s.opts()]
    origin = [float(v) for v in s.grid_origin]
    feats = torch.empty((1, 1, H, W, C), dtype=torch.float32, device=dev)
    make_features_torch(1, H, W, C, dev, seed=0, out=feats[0])
    count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
    out = torch.zeros(n_vox + 1, C, device=dev)
    away = s.c2w[:1].copy()
    away[0, :3, 3] += np.array([500.0, 500.0, 500.0], np.float32)          # far outside: every ray misses
    for label, c2w in (("a view of the scene", s.c2w[:1]), ("a view that hits nothing", away)):
        vmi = torch.from_numpy(c2w).reshape(-1).to(dev)
        ws = voxproj_host.Workspace()
        for rep in range(3):
            voxproj_host.profile_enable(rep > 0)
            for _ in range(20):
                voxproj_host.project_features_raw(feats, occ, vmi, intr, opts, count, out, origin, s.voxel_size, workspace=ws, sync=True)
        p = voxproj_host.profile_read()
        voxproj_host.profile_enable(False)
        n = max(p["gather_launches"], 1)
        print(f"{name} one view per call, {label}: gather {p['gather_ms'] / n * 1e3:.1f} us, march {p['first_hit_ms'] / max(p['first_hit_launches'], 1) * 1e3:.1f} us "
              f"({(n_vox + 3) // 4} workgroups in the grid, {int((count > 0).sum().item())} voxels ever touched)", flush=True)
        ws.release()
        count.zero_(); out.zero_()
