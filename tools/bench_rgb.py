"""BASELINE config 5 (R4): the RGB path -- 500 000 voxels x 1000 views, uint8 [1168,1752,3] images, voxel-driven
nearest pixel, no occlusion (debug_project_colors.py:54-81 semantics).  Low arithmetic intensity: 3 bytes gathered
per voxel-view, so no HBM roofline claim -- reports voxel-views/s and the gathered GB/s.  A pool of distinct images
is cycled (1000 x 6.1 MB = 6.1 GB would fit, 64 are used to keep set-up short)."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "3d-semantic-segmentation_amd"))
import torch, voxproj_host
from synthetic_scene import make_scene
dev = torch.device("cuda:0")
N, V, W, H = 500000, 1000, 1752, 1168
s = make_scene(N, V, W, H, seed=0)
occ = torch.from_numpy(s.occ).to(dev)
pool = 64
imgs = torch.randint(0, 256, (pool, H, W, 3), dtype=torch.uint8, device=dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr)[None].repeat(V, 1).to(dev).contiguous()
csum = torch.zeros(N + 1, 3, device=dev); hits = torch.zeros(N + 1, dtype=torch.int32, device=dev)
first = torch.full((N + 1,), 2 ** 30, dtype=torch.int32, device=dev)
origin = [float(v) for v in s.grid_origin]
def run():
    csum.zero_(); hits.zero_()
    for a in range(0, V, pool):
        b = min(V, a + pool)
        voxproj_host.project_colors_raw(occ, c2w[a:b].contiguous(), intr[a:b].contiguous(), origin, s.voxel_size,
                                        imgs[: b - a], csum, hits, first_view=first, view_base=a)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
seen = int(hits.sum().item())
print(json.dumps({"workload": f"R4: {N} voxels x {V} views x {W}x{H}x3 uint8, DPC semantics", "ms_per_pass": round(dt * 1e3, 2),
                  "Mvoxel_views_per_s": round(N * V / dt / 1e6, 1), "voxel_view_hits": seen,
                  "gathered_GBps": round(seen * 3 / dt / 1e9, 2)}))
