"""would it pay to queue the march of a pipelined call with a lower occupancy cap (VP_OPT_MARCH_LDS_KB 24 instead
of 41 KiB) when the call whose gather it runs beside is mostly misses?  The host enqueues a whole pass ahead of the device, so at
enqueue time it cannot KNOW the previous call's hit fraction without waiting for it; this probe measures the UPPER BOUND of the idea:
the arm `by_hits` is told every call's hit fraction in advance (from an untimed pre-pass) and caps the march of call j at --low KiB
when call j-1 hit less than --below of its pixels.  Arms alternate on ONE allocation, whole pipelined passes as bench.py issues them:
  python tools/probe_march_cap.py [--workload R2T|A1|R2] [--passes 6] [--low 24] [--below 0.5] [--f16]
Prints ms per pass per arm (mean, min, all)."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_features_torch  # noqa: E402


def arg(name, default):
    for i, a in enumerate(sys.argv):
        if a == name:
            return sys.argv[i + 1]
    return default


argv = sys.argv
sys.argv = ["bench.py"]
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bm)
sys.argv = argv
name = arg("--workload", "R2T")
passes = int(arg("--passes", "6"))
low = int(arg("--low", "24"))
below = float(arg("--below", "0.5"))
half = "--f16" in sys.argv
n_vox, n_views, W, H, C = bm.WORKLOADS[name]
esize = 2 if half else 4
V, n_calls, _ = bm.plan_calls(n_views, H, W, C, esize)
dev = torch.device("cuda", 0)
s = bm.workload_scene(name)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
feats = torch.empty((1, V, H, W, C), dtype=torch.float16 if half else torch.float32, device=dev)
if half:
    tmp = torch.empty((1, H, W, C), dtype=torch.float32, device=dev)
    for v in range(V):
        make_features_torch(1, H, W, C, dev, seed=v, out=tmp)
        feats[0, v] = tmp[0].half()
    del tmp
else:
    make_features_torch(V, H, W, C, dev, seed=0, out=feats[0])
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
out = torch.zeros(n_vox + 1, C, dtype=torch.float32, device=dev)
calls = [list(range(ci * V, min(n_views, (ci + 1) * V))) for ci in range(n_calls)]
vmis = [c2w[v].reshape(-1).contiguous() for v in calls]
ws = voxproj_host.Workspace()
# pre-pass: every call's hit fraction
frac = []
for ci, views in enumerate(calls):
    count.zero_()
    voxproj_host.project_features_raw(feats[:, :len(views)], occ, vmis[ci], intr, opts, count, out, origin, s.voxel_size, workspace=ws, sync=True)
    frac.append(float(count.sum().item()) / (len(views) * H * W))
torch.cuda.synchronize(dev)


def one_pass(arm):
    out.zero_(); count.zero_()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for ci, views in enumerate(calls):
        cap = -1
        if arm == "all_low" or (arm == "by_hits" and ci > 0 and frac[ci - 1] < below):
            cap = low
        ws.set_option(voxproj_host.VP_OPT_MARCH_LDS_KB, cap)
        voxproj_host.project_features_raw(feats[:, :len(views)], occ, vmis[ci], intr, opts, count, out, origin, s.voxel_size,
                                          workspace=ws, sync=False, pipeline=True, reuse_accel=True)
    voxproj_host.workspace_status(ws, dev)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e3


arms = ("default", "by_hits", "all_low")
res = {a: [] for a in arms}
for a in arms:
    one_pass(a)
for _ in range(passes):
    for a in arms:
        res[a].append(one_pass(a))
print(f"# {name}{' fp16' if half else ''}: {n_calls} pipelined calls of {V} views per pass; hit fraction per call {[round(f, 2) for f in frac]}; "
      f"march cap of call j lowered to {low} KiB when call j-1 hit < {below} of its pixels (by_hits, told in advance), or always (all_low); "
      f"{passes} passes per arm, alternating, one allocation")
for a in arms:
    t = np.array(res[a])
    print(f"{a:8s} {t.mean():8.3f} ms per pass (min {t.min():8.3f})  {' '.join(f'{x:.2f}' for x in t)}")
