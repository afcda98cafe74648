"""How many ray samples the march examines, per ray, on the bench scene -- from the diagnostic build of the library
(`make -C 3d-semantic-segmentation_amd/csrc diag` -> tools/libvoxproj_diag.so, never shipped), whose k_first_hit writes per-ray
counters into the hit image instead of voxel IDs:

    VOXPROJ_LIB=tools/libvoxproj_diag.so python tools/march_evals.py [--workload R2|R1] [--views 4]

far   = samples examined in far mode (approximate position, only to bound a skip: ~30 VALU instructions each)
leap  = samples evaluated with the reference's exact arithmetic that allowed a skip (D >= 2)
fine  = samples evaluated exactly next to geometry (D < 2), one by one
The reference loop (VP_FLAG_EXACT_MARCH) evaluates every sample: (dmax - dmin) / (camDir.z * inc) per ray."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="R2")
ap.add_argument("--views", type=int, default=4)
a = ap.parse_args()
assert "diag" in voxproj_host.LIB_PATH, "run with VOXPROJ_LIB=tools/libvoxproj_diag.so"
n_vox, n_views, W, H = {"R2": (200000, 300, 968, 548), "R1": (80000, 100, 484, 274)}[a.workload]
dev = torch.device("cuda", 0)
s = make_scene(n_vox, n_views, W, H, seed=0)
V, C = a.views, 8
views = list(range(0, n_views, max(1, n_views // V)))[:V]
feats = torch.zeros(1, V, H, W, C, device=dev)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
vmi = torch.from_numpy(s.c2w[views]).reshape(-1).contiguous().to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
out = torch.zeros(n_vox + 1, C, device=dev)
ws = voxproj_host.project_features_raw(feats, occ, vmi, intr, [float(v) for v in s.opts()], count, out, [float(v) for v in s.grid_origin],
                                       s.voxel_size, sync=True, extra_flags=1 << 20)
img = voxproj_host.hit_image(ws, dev).cpu().numpy().astype(np.int64)
far, leap, fine = (img >> 20) & 1023, (img >> 10) & 1023, img & 1023
print(f"{a.workload}, views {views}: per ray, mean (max)  far {far.mean():.2f} ({far.max()})  exact with a skip {leap.mean():.2f} ({leap.max()})  "
      f"exact one by one {fine.mean():.2f} ({fine.max()})  -> exact evaluations {(leap + fine).mean():.2f}, examined samples {(far + leap + fine).mean():.2f}")
tot_ray = (far + leap + fine).reshape(-1)
qs = [50, 75, 90, 95, 99, 99.9]
print("examined samples per ray, percentiles " + ", ".join(f"{q}%: {np.percentile(tot_ray, q):.0f}" for q in qs) +
      f";  share of all examined samples spent by rays with > 20: {tot_ray[tot_ray > 20].sum() / tot_ray.sum():.3f}, > 40: {tot_ray[tot_ray > 40].sum() / tot_ray.sum():.3f}"
      f";  rays with > 20: {(tot_ray > 20).mean():.4f}")
fine_ray = fine.reshape(-1)
print("single steps next to geometry per ray, percentiles " + ", ".join(f"{q}%: {np.percentile(fine_ray, q):.0f}" for q in qs) +
      f";  share of all single steps spent by rays with > 8 of them: {fine_ray[fine_ray > 8].sum() / max(1, fine_ray.sum()):.3f}")
# per 8x8 tile (one wavefront): the wavefront iterates until its slowest lane is done
def tiles(x):
    hh, ww = (H // 8) * 8, (W // 8) * 8
    return x[0, :, :hh, :ww].reshape(V, hh // 8, 8, ww // 8, 8)
tot = tiles(far + leap + fine)
tmax = tot.max(axis=(2, 4))
print("per wavefront: max over lanes, percentiles over tiles " + ", ".join(f"{q}%: {np.percentile(tmax, q):.0f}" for q in qs) + f", max {tmax.max()}")
print(f"per wavefront (8x8 tile): max over lanes of examined samples, mean over tiles {tot.max(axis=(2, 4)).mean():.2f}; "
      f"of exact evaluations {tiles(leap + fine).max(axis=(2, 4)).mean():.2f}")
