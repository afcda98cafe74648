"""Is the gather's two-speed behaviour tied to an allocation (physical placement) or does it flip between repeats on
the same buffers?  R rounds of re-allocation x N repeats of the same 8 calls on the same buffers."""
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
V, NCALL = 16, 8
s = make_scene(n_vox, n_views, W, H, seed=0)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
vmis = [c2w[i * V:(i + 1) * V].reshape(-1).contiguous() for i in range(NCALL)]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
pipeline = "--pipeline" in sys.argv

feats = out = count = ws = None
for rnd in range(6):
    if which in ("all", "feats") or feats is None:
        feats = None
        torch.cuda.empty_cache()
        feats = torch.empty((1, V, H, W, C), dtype=torch.float32, device=dev)
        make_features_torch(V, H, W, C, dev, seed=0, out=feats[0])
    if which in ("all", "out") or out is None:
        out = count = None
        torch.cuda.empty_cache()
        count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
        out = torch.zeros(n_vox + 1, C, dtype=torch.float32, device=dev)
    if which in ("all", "ws") or ws is None:
        if ws is not None:
            ws.release()
        ws = None
        torch.cuda.empty_cache()
        ws = voxproj_host.Workspace()
    res = []
    for rep in range(7):
        voxproj_host.profile_enable(rep > 0)
        for ci in range(NCALL):
            voxproj_host.project_features_raw(feats, occ, vmis[ci], intr, opts, count, out, origin, s.voxel_size,
                                              workspace=ws, sync=False, reuse_accel=(ci > 0 or None), pipeline=pipeline)
        if pipeline:
            voxproj_host.workspace_status(ws, dev)
        torch.cuda.synchronize()
        if rep > 0:
            p = voxproj_host.profile_read()
            res.append(p["gather_ms"] / max(p["gather_launches"], 1))
        voxproj_host.profile_enable(False)
    print(f"realloc {which} round {rnd}: feats@{feats.data_ptr():#x} out@{out.data_ptr():#x} count@{count.data_ptr():#x} "
          f"ws@{ws.ptr():#x}  gather ms/launch per repeat:", " ".join(f"{r:.3f}" for r in res), flush=True)
