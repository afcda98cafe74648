"""How many ray samples does the leaping march evaluate?  With VOXPROJ_DEBUG_EVALS set the library writes, instead of
the first-hit IDs, (leaps << 16) | near-field steps per ray into the hit image; this prints their statistics on the R2
scene (the reference loop takes ~175 samples per ray there).  python tools/dbg_evals.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
os.environ["VOXPROJ_DEBUG_EVALS"] = "1"

import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_scene  # noqa: E402

dev = torch.device("cuda:0")
V, H, W, C = 4, 548, 968, 8
s = make_scene(200000, 8, W, H, seed=0)
feats = torch.zeros(1, V, H, W, C, device=dev)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
count = torch.zeros(s.n_vox + 1, dtype=torch.int32, device=dev)
out = torch.zeros(s.n_vox + 1, C, device=dev)
ws = voxproj_host.Workspace()
voxproj_host.project_features_raw(feats, occ, torch.from_numpy(s.c2w[:V]).reshape(-1).to(dev), torch.from_numpy(s.intr[None]).to(dev),
                                  [float(v) for v in s.opts()], count, out, [float(v) for v in s.grid_origin], s.voxel_size,
                                  workspace=ws, sync=True)
h = voxproj_host.hit_image(ws, dev).cpu().numpy()[0]
leap, fine = h >> 16, h & 0xFFFF
total = leap + fine
print(f"per ray: {total.mean():.2f} samples evaluated = {leap.mean():.2f} leaps (max {leap.max()}) + "
      f"{fine.mean():.2f} near-field steps (max {fine.max()})")


def per_wave_max(a):     # 8x8-pixel tiles = one wavefront each (the wave runs until its slowest lane is done)
    return a[0][:544, :968].reshape(68, 8, 121, 8).max(axis=(1, 3)).mean()


print(f"per wavefront (max over its 64 rays): {per_wave_max(total):.2f} samples = {per_wave_max(leap):.2f} leaps + "
      f"{per_wave_max(fine):.2f} steps")
print("percentiles 10/50/90/99: leaps", np.percentile(leap, [10, 50, 90, 99]), " steps", np.percentile(fine, [10, 50, 90, 99]))
