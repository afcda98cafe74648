"""Does work on a high-priority stream make progress UNDER a gather that fills the chip?  (What the split collective of
bench.py's multi-rank step relies on: RCCL's kernels on a high-priority stream beside the second row range's gather.)
One GPU cannot run a real collective, so the stand-in is a device-to-device copy of the half-scene rows (205 MB, read +
written = 410 MB of HBM traffic, ~1 % of the gather's) issued on a stream of priority -1 / 0 right after the gather is queued.
Prints: the copy alone, the copy beside the gather (from its own events), the gather alone and beside the copy.
python tools/probe_concurrent_copy.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_features_torch, make_scene  # noqa: E402

dev = torch.device("cuda", 0)
n_vox, n_views, W, H, C, V = 200000, 300, 968, 548, 512, 38
s = make_scene(n_vox, n_views, W, H, seed=0)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
vmi = c2w[:V].reshape(-1).contiguous()
feats = torch.empty((1, V, H, W, C), dtype=torch.float32, device=dev)
make_features_torch(V, H, W, C, dev, seed=0, out=feats[0])
n_rows = n_vox + 1
count = torch.zeros(n_rows, dtype=torch.int32, device=dev)
out = torch.zeros(n_rows, C, dtype=torch.float32, device=dev)
h = (n_rows // 2 + 63) & ~63
src = torch.randn(h, C, device=dev)
dst = torch.empty_like(src)
ws = voxproj_host.Workspace()


def split_pass(side, copies):
    """zero, gather rows [0,h), [copies of 205 MB on `side`], gather rows [h,n) -- as bench.py's multi-rank step does"""
    e0, e1, c0, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    out.zero_(); count.zero_()
    torch.cuda.synchronize()
    e0.record()
    ws.set_row_range(0, h)
    voxproj_host.project_features_raw(feats, occ, vmi, intr, opts, count, out, origin, s.voxel_size, workspace=ws, sync=False, pipeline=True)
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            c0.record()
            for _ in range(copies):
                dst.copy_(src)
            c1.record()
    ws.set_row_range(h, n_rows)
    voxproj_host.project_features_raw(feats, occ, vmi, intr, opts, count, out, origin, s.voxel_size, workspace=ws, sync=False, pipeline=True,
                                      gather_only=True)
    ws.set_row_range()
    e1.record()
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), (c0.elapsed_time(c1) if side is not None else 0.0)


for prio in (-1, 0):
    side = torch.cuda.Stream(dev, priority=prio)
    for copies in (1, 4, 8):
        split_pass(side, copies)
        # the copies alone
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            a0.record()
            for _ in range(copies):
                dst.copy_(src)
            a1.record()
        torch.cuda.synchronize()
        alone = a0.elapsed_time(a1)
        base = min(split_pass(None, 0)[0] for _ in range(3))
        both = [split_pass(side, copies) for _ in range(3)]
        p, c = min(b[0] for b in both), min(b[1] for b in both)
        print(f"side stream priority {prio:2d}, {copies} x 205 MB copy: alone {alone:.3f} ms; beside the second gather {c:.3f} ms; "
              f"projection (march + two gathers) {base:.3f} ms alone, {p:.3f} ms with the copies beside it", flush=True)
