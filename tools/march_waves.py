"""When do the wavefronts of a ONE-view march start and end?  From the diagnostic build of the library (`make -C
3d-semantic-segmentation_amd/csrc diag` -> tools/libvoxproj_diag.so, never shipped), whose k_first_hit leaves, per wavefront, its
start and end stamps of the 100 MHz clock and its iteration count (its slowest lane's):

    VOXPROJ_LIB=tools/libvoxproj_diag.so python tools/march_waves.py [--workload R2|R1] [--view 0]

(Round 5's tail-split experiment used it with a fourth and fifth word per wavefront: profiles/r05_march_tail_split.log.)"""
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
ap.add_argument("--workload", default="R1")
ap.add_argument("--view", type=int, default=0)
a = ap.parse_args()
assert "diag" in voxproj_host.LIB_PATH, "run with VOXPROJ_LIB=tools/libvoxproj_diag.so"
n_vox, n_views, W, H = {"R2": (200000, 300, 968, 548), "R1": (80000, 100, 484, 274)}[a.workload]
dev = torch.device("cuda", 0)
s = make_scene(n_vox, n_views, W, H, seed=0)
C = 8
feats = torch.zeros(1, 1, H, W, C, device=dev)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
vmi = torch.from_numpy(s.c2w[[a.view]]).reshape(-1).contiguous().to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
out = torch.zeros(n_vox + 1, C, device=dev)
ws = voxproj_host.Workspace()
for rep in range(3):        # warm: tables built, caches in the state of a running job
    voxproj_host.project_features_raw(feats, occ, vmi, intr, [float(v) for v in s.opts()], count, out, [float(v) for v in s.grid_origin],
                                      s.voxel_size, workspace=ws, sync=True, extra_flags=1 << 21)
img = voxproj_host.hit_image(ws, dev).cpu().numpy()[0, 0].astype(np.int64)
hh, ww = (H // 8) * 8, (W // 8) * 8
t = img[:hh, :ww].reshape(hh // 8, 8, ww // 8, 8)
row0 = t[:, 0, :, :]                      # [tiles_y, tiles_x, 8]: the wavefront's first row of lanes
t0, t1 = row0[..., 0] & 0xffffffff, row0[..., 1] & 0xffffffff
ita = row0[..., 2]
itb = np.zeros_like(ita)
base = t0.min()
start, end = (t0 - base) * 0.01, (t1 - base) * 0.01         # microseconds
dur = end - start
print(f"{a.workload} view {a.view}: {dur.size} full wavefronts; launch span (first start -> last end) {end.max():.1f} us")
qs = [50, 90, 99, 99.9, 100]
print("  wavefront start  (us after the first): " + ", ".join(f"{q}%: {np.percentile(start, q):.1f}" for q in qs))
print("  wavefront duration (us):               " + ", ".join(f"{q}%: {np.percentile(dur, q):.1f}" for q in qs))
print("  wavefront end    (us after the first start): " + ", ".join(f"{q}%: {np.percentile(end, q):.1f}" for q in qs))
print("  iterations:                            " + ", ".join(f"{q}%: {np.percentile(ita, q):.0f}" for q in qs))
k = np.argsort(dur.reshape(-1))[-5:]
print("  five longest wavefronts: " + "; ".join(f"dur {dur.reshape(-1)[i]:.1f} us, it {ita.reshape(-1)[i]}, start {start.reshape(-1)[i]:.1f}" for i in k))
print(f"  us per iteration (duration / iterations): {np.median(dur / np.maximum(ita, 1)):.2f} median")
