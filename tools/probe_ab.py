"""A/B of two builds of libvoxproj.so on the SAME feature-pool allocation (its placement moves the gather's speed by
several per cent, so builds cannot be compared across processes): python tools/probe_ab.py libA.so libB.so [...]
Prints mean k_gather / k_first_hit time per launch, serial phases (--pipeline: pipelined), R2 scene with 16 views per
call (--r1: the R1 scene, 25 views per call; --f16: fp16 feature maps; --v=N: N views per call)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_features_torch, make_scene  # noqa: E402

libs = [os.path.abspath(a) for a in sys.argv[1:] if a.endswith(".so")]
pipeline = "--pipeline" in sys.argv
half = "--f16" in sys.argv
# --heavy-ts=a,b,c: per library also sweep the heavy-voxel threshold (VP_OPT_HEAVY_THRESHOLD of the workspace; 0 = default)
heavy_ts = next(([int(v) for v in a.split("=")[1].split(",")] for a in sys.argv if a.startswith("--heavy-ts=")), [None])
dev = torch.device("cuda", 0)
if "--r1" in sys.argv:      # BASELINE config 2
    n_vox, n_views, W, H, C = 80000, 100, 484, 274, 512
    V, NCALL = 25, 4
else:
    n_vox, n_views, W, H, C = 200000, 300, 968, 548, 512
    V, NCALL = 16, 8
# --v=N: views per call (the pool holds N maps; calls cycle through the scene's poses)
for _a in sys.argv:
    if _a.startswith("--v="):
        V = int(_a.split("=")[1])
        NCALL = max(2, min(NCALL, n_views // V))
s = make_scene(n_vox, n_views, W, H, seed=0)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
vmis = [c2w[i * V:(i + 1) * V].reshape(-1).contiguous() for i in range(NCALL)]
feats = torch.empty((1, V, H, W, C), dtype=torch.float32, device=dev)
make_features_torch(V, H, W, C, dev, seed=0, out=feats[0])
if half:
    feats = feats.half()
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
out = torch.zeros(n_vox + 1, C, dtype=torch.float32, device=dev)
ref = None
for rnd in range(3):
    for path, ht in [(p_, h_) for p_ in libs for h_ in heavy_ts]:
        voxproj_host._lib = None
        voxproj_host.LIB_PATH = path
        ws = voxproj_host.Workspace()
        if ht is not None:
            ws.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, ht or None)
        out.zero_(); count.zero_()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(3):
            voxproj_host.profile_enable(rep > 0)
            if rep == 1:
                ev0.record()
            for ci in range(NCALL):
                voxproj_host.project_features_raw(feats, occ, vmis[ci], intr, opts, count, out, origin, s.voxel_size,
                                                  workspace=ws, sync=False, reuse_accel=(ci + rep > 0 or None), pipeline=pipeline)
            if pipeline:
                voxproj_host.workspace_status(ws, dev)
            torch.cuda.synchronize()
        ev1.record(); torch.cuda.synchronize()
        p = voxproj_host.profile_read()
        voxproj_host.profile_enable(False)
        chk = (int(count.sum().item()), float(out.double().sum().item()))
        if ref is None:
            ref = chk
        print(f"round {rnd} {os.path.basename(path) + ('' if ht is None else f' T={ht}'):24s} gather {p['gather_ms'] / max(p['gather_launches'], 1):.3f} ms/launch  "
              f"march {p['first_hit_ms'] / max(p['first_hit_launches'], 1):.3f}  wall {ev0.elapsed_time(ev1) / (2 * NCALL):.3f} ms/call  "
              f"same result: {chk == ref}", flush=True)
        ws.release()
        del ws
