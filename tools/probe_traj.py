"""A/B of builds and split-voxel options on the calls of a trajectory leg, each call ALONE on the device, all arms on the SAME
feature-pool allocation (its placement moves the gather by several per cent, so arms cannot be compared across processes):
  python tools/probe_traj.py [--workload R2T|A1|R2] [--calls 0,1,2] [--rounds 3] [--f16] ARM [ARM ...]
  ARM = path/to/lib.so[:heavy=N][:part=N]     (heavy / part: VP_OPT_HEAVY_THRESHOLD / VP_OPT_PART_PIXELS of the workspace)
Prints, per call and arm, the mean k_gather time per launch (HIP events of the library), the fraction of 8 TB/s on the call's
algorithmic bytes, and the number of parts.  --pipeline: also the WHOLE pass of the leg per arm, its calls pipelined
(VP_FLAG_PIPELINE) as bench.py issues them, wall time per pass."""
import importlib.util
import os
import sys

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
rounds = int(arg("--rounds", "3"))
arms = [a for a in sys.argv[1:] if ".so" in a]
n_vox, n_views, W, H, C = bm.WORKLOADS[name]
half = "--f16" in sys.argv
esize = 2 if half else 4
V, n_calls, _ = bm.plan_calls(n_views, H, W, C, esize)
call_ids = [int(v) for v in arg("--calls", ",".join(str(i) for i in range(n_calls))).split(",")]
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
res = {}
info = {}
for rnd in range(rounds):
    for arm in arms:
        parts = arm.split(":")
        path = os.path.abspath(parts[0])
        kv = dict(p.split("=") for p in parts[1:])
        voxproj_host._lib = None
        voxproj_host.LIB_PATH = path
        ws = voxproj_host.Workspace()
        if "heavy" in kv:
            ws.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, int(kv["heavy"]))
        if "part" in kv:
            ws.set_option(voxproj_host.VP_OPT_PART_PIXELS, int(kv["part"]))
        for ci in call_ids:
            views = list(range(ci * V, min(n_views, (ci + 1) * V)))
            vmi = c2w[views].reshape(-1).contiguous()
            for rep in range(3):
                out.zero_(); count.zero_()
                voxproj_host.profile_enable(rep > 0)
                voxproj_host.project_features_raw(feats[:, :len(views)], occ, vmi, intr, opts, count, out, origin, s.voxel_size,
                                                  workspace=ws, sync=True)
                if rep > 0:
                    p = voxproj_host.profile_read()
                    res.setdefault((ci, arm), []).append((p["gather_ms"], p["first_hit_ms"], p["heavy_ms"]))
            ph, nt = int(count.sum().item()), int((count > 0).sum().item())
            ctr = voxproj_host.counters(ws, dev)
            info[(ci, arm)] = (ph * C * esize + nt * C * 4 * 2 + len(views) * H * W * 4 + (n_vox + 1) * 8, ctr["n_parts"], ctr["n_heavy"], ctr["heavy_t"],
                               float(out.double().sum().item()))
        voxproj_host.profile_enable(False)
        if "--pipeline" in sys.argv:
            import time
            all_calls = [list(range(ci * V, min(n_views, (ci + 1) * V))) for ci in range(n_calls)]
            vmis = [c2w[v].reshape(-1).contiguous() for v in all_calls]
            torch.cuda.synchronize(dev)
            for rep in range(4):
                out.zero_(); count.zero_()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                if rep == 3:
                    voxproj_host.profile_enable(True)
                for ci, views in enumerate(all_calls):
                    voxproj_host.project_features_raw(feats[:, :len(views)], occ, vmis[ci], intr, opts, count, out, origin, s.voxel_size,
                                                      workspace=ws, sync=False, pipeline=True, reuse_accel=True)
                voxproj_host.workspace_status(ws, dev)
                torch.cuda.synchronize(dev)
                if rep > 0:
                    res.setdefault(("pass", arm), []).append((time.perf_counter() - t0) * 1e3)
            p = voxproj_host.profile_read()
            voxproj_host.profile_enable(False)
            info[("pass", arm)] = (p["gather_ms"], p["first_hit_ms"], p["heavy_ms"])
        ws.release()
if "--pipeline" in sys.argv:
    for arm in arms:
        t = np.array(res[("pass", arm)])
        g = info[("pass", arm)]
        print(f"pass    {os.path.basename(arm):44s} {t.mean():8.3f} ms per pipelined pass (min {t.min():8.3f}, {len(t)} passes)  "
              f"gather {g[0]:7.3f}  march {g[1]:7.3f}  combine {g[2]:6.3f}  -> {n_vox * n_views / (t.mean() * 1e-3) / 1e6:8.1f} Mvoxel-views/s")
print(f"# {name}{' fp16 feature maps' if half else ''}: {V} views per call, calls {call_ids}, {rounds} rounds x 2 timed launches per arm, every call alone on the device")
for ci in call_ids:
    for arm in arms:
        t = np.array(res[(ci, arm)])
        b, npart, nheavy, ht, chk = info[(ci, arm)]
        g = t[:, 0].mean()
        print(f"call {ci}  {os.path.basename(arm):44s} gather {g:7.3f} ms (min {t[:, 0].min():7.3f})  frac {b / (g * 1e-3) / 8e12:.4f}  "
              f"march {t[:, 1].mean():6.3f}  combine {t[:, 2].mean():6.3f}  parts {npart:6d} heavy {nheavy:5d} t {ht:6d}  checksum {chk:.6e}")
