"""Gather time per view at V = 32 / 16 / 8 views per call on the SAME feature-pool allocation, for several
re-allocations of the pool: does the slow placement hurt less when fewer views (pages) are live at once?"""
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

for rnd in range(7):
    feats = None
    torch.cuda.empty_cache()
    hold = [torch.empty(int(np.random.default_rng(rnd).integers(0, 3000)) << 20, dtype=torch.uint8, device=dev)] if rnd % 2 else []
    feats = torch.empty((1, POOL, H, W, C), dtype=torch.float32, device=dev)
    make_features_torch(POOL, H, W, C, dev, seed=0, out=feats[0])
    line = []
    for V in (32, 16, 8):
        ncall = 64 // V
        vm = [c2w[i * V:(i + 1) * V].reshape(-1).contiguous() for i in range(ncall)]
        for rep in range(2):
            voxproj_host.profile_enable(rep > 0)
            for ci in range(ncall):
                sl = (ci * V) % POOL
                voxproj_host.project_features_raw(feats[:, sl:sl + V], occ, vm[ci], intr, opts, count, out, origin,
                                                  s.voxel_size, workspace=ws, sync=False, reuse_accel=(rep + ci > 0 or None))
            torch.cuda.synchronize()
        p = voxproj_host.profile_read()
        voxproj_host.profile_enable(False)
        line.append(f"V={V}: {p['gather_ms'] / 64 * 1e3:.1f} us/view")
    gbs = voxproj_host.stream_read_gbs(feats)
    print(f"alloc {rnd} feats@{feats.data_ptr():#x}  " + "  ".join(line) + f"  stream {gbs:.0f} GB/s", flush=True)
    del hold
