"""Latency of the drop-in call used the way the reference uses it (debug_project_features.py:141-208): ONE view per
blocking call at the R2 resolution, (a) the same occupancy tensor every call, (b) a fresh `.long()` copy per call as
DPF:143 makes -- the occupancy-derived tables are then rebuilt every time.  python tools/bench_dropin.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import project_features_cuda as m  # noqa: E402  (the compiled extension)
import project_features_front as front  # noqa: E402
from synthetic_scene import make_features_torch, make_scene  # noqa: E402

dev = torch.device("cuda", 0)
n_vox, W, H, C, NV = 200000, 968, 548, 512, 16
s = make_scene(n_vox, 300, W, H, seed=0)
feats = make_features_torch(NV, H, W, C, dev, seed=0)
occ32 = torch.from_numpy(s.occ).to(dev)
occ = occ32.unsqueeze(0).long().contiguous()
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = torch.from_numpy(s.opts())
origin = torch.from_numpy(s.grid_origin)
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
out = torch.zeros(n_vox + 1, C, device=dev)
pm = torch.tensor([False])
for fresh in (False, True):
    for fn, name in ((m.project_features_cuda, "compiled"), (front.project_features_cuda_py, "python")):
        ts = []
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for v in range(NV):
                o = occ32.unsqueeze(0).long().contiguous() if fresh else occ
                fn(feats[v:v + 1].unsqueeze(0), o, c2w[v].reshape(-1).contiguous(), intr, opts, count, out, pm, origin, s.voxel_size)
            ts.append((time.perf_counter() - t0) / NV)
        t = min(ts)
        print(f"{'fresh occupancy tensor per call' if fresh else 'same occupancy tensor':32s} {name:9s} {t * 1e3:.3f} ms/call  "
              f"{n_vox / t / 1e6:.0f} Mvoxel-views/s  ({W * H * C * 4 / t / 1e9:.0f} GB/s of feature rows)")
