"""A/B of builds and one-view options on single frames of a workload, ONE view per blocking call (what the unchanged reference
pipeline issues, debug_project_features.py:201-208), all arms on the SAME feature map and output allocation:
  python tools/probe_one_view.py [--workload R2T|A1|R2|R1] [--view-ids 0,30,59,100,150,200] [--reps 5] [--f16] ARM [ARM ...]
  ARM = path/to/lib.so[:heavy=N][:split=N][:part=N][:grid=N][:serial=1]
        (VP_OPT_HEAVY_THRESHOLD / VP_OPT_ONE_VIEW_SPLIT / VP_OPT_PART_PIXELS / VP_OPT_ONE_VIEW_GATHER of the workspace)
Per frame and arm: wall time of the blocking call (min and median over --reps), the library's HIP-event times per kernel group,
the fraction of 8 TB/s on the call's algorithmic bytes, voxels above the heavy threshold / split voxels / parts, and
the largest difference of the output rows from the FIRST arm's, relative to each row's largest element (counts must be equal)."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_features_torch, make_scene  # noqa: E402


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
reps = int(arg("--reps", "5"))
half = "--f16" in sys.argv
arms = [a for a in sys.argv[1:] if ".so" in a]
n_vox, n_views, W, H, C = bm.WORKLOADS[name]
s = bm.workload_scene(name) if name in ("A1", "R2T") else make_scene(n_vox, n_views, W, H, seed=0)
view_ids = [int(v) for v in arg("--view-ids", "0,30,59,100,150,200").split(",")]
dev = torch.device("cuda", 0)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
feats = torch.empty((1, 1, H, W, C), dtype=torch.float32, device=dev)
make_features_torch(1, H, W, C, dev, seed=0, out=feats[0])
if half:
    feats = feats.half()
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
out = torch.zeros(n_vox + 1, C, dtype=torch.float32, device=dev)
OPT = {"heavy": voxproj_host.VP_OPT_HEAVY_THRESHOLD, "split": voxproj_host.VP_OPT_ONE_VIEW_SPLIT, "part": voxproj_host.VP_OPT_PART_PIXELS,
       "grid": voxproj_host.VP_OPT_ONE_VIEW_GATHER}
rows = []
ref = {}
for arm in arms:
    parts = arm.split(":")
    voxproj_host._lib = None
    voxproj_host.LIB_PATH = os.path.abspath(parts[0])
    import ctypes
    abi = ctypes.CDLL(voxproj_host.LIB_PATH).vp_abi_version()
    voxproj_host.VP_ABI_VERSION = abi      # (an older build as the baseline arm: it has no VP_OPT_ONE_VIEW_SPLIT)
    voxproj_host.VP_OPT_ONE_VIEW_SPLIT = 7 if abi >= 4 else voxproj_host.VP_OPT_PART_PIXELS
    ws = voxproj_host.Workspace()
    serial = False
    for kv in parts[1:]:
        k, v = kv.split("=")
        if k == "serial":      # VP_FLAG_SERIAL_SUMS: every voxel by one wavefront, the oracle's order (the parity aggregator's calls)
            serial = bool(int(v))
            continue
        if k == "split" and abi < 4:
            continue
        ws.set_option(OPT[k], int(v))
    for vi in view_ids:
        vmi = c2w[vi].reshape(-1).contiguous()
        call = lambda: voxproj_host.project_features_raw(feats, occ, vmi, intr, opts, count, out, origin, s.voxel_size, workspace=ws, sync=True, serial_sums=serial)
        out.zero_(); count.zero_()
        call()
        got_c, got_o = count.clone(), out.clone()
        ctr = voxproj_host.counters(ws, dev)
        ph, nt = int(got_c.sum().item()), int((got_c > 0).sum().item())
        algo = ph * C * (2 if half else 4) + nt * C * 4 * 2 + H * W * 4 * 2 + (n_vox + 1) * 4 * 2
        if vi not in ref:
            ref[vi] = (got_c, got_o)
            diff = 0.0
        else:
            assert torch.equal(got_c, ref[vi][0]), f"{arm}: pixel counts differ from the first arm's on frame {vi}"
            scale = ref[vi][1].abs().amax(dim=1, keepdim=True) + 1e-30
            diff = float(((got_o - ref[vi][1]).abs() / scale).max().item())
        call()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t0)
        voxproj_host.profile_enable(True)
        for _ in range(reps):
            call()
        p = voxproj_host.profile_read()
        voxproj_host.profile_enable(False)
        t = min(ts)
        rows.append((vi, arm, t, float(np.median(ts)), p, algo, ctr, diff))
    ws.release()
print(f"# {name}{' fp16' if half else ''}: one view per blocking call ({W}x{H}x{C}), {reps} timed calls + {reps} with HIP events per frame and arm; ctypes front")
for vi in view_ids:
    for r in rows:
        if r[0] != vi:
            continue
        _, arm, t, tm, p, algo, ctr, diff = r
        g = (p["gather_ms"] + p["heavy_ms"]) / reps
        print(f"frame {vi:3d}  {os.path.basename(arm):40s} {t * 1e3:.4f} ms/call (median {tm * 1e3:.4f})  = {algo / t / 8e12:.3f} of peak | march+list {p['first_hit_ms'] / reps * 1e3:6.1f} us "
              f"gather {p['gather_ms'] / reps * 1e3:6.1f} us combine {p['heavy_ms'] / reps * 1e3:5.1f} us -> gather+combine {algo / (g * 1e-3) / 1e12:.2f} TB/s = {algo / (g * 1e-3) / 8e12:.3f}"
              f" | heavy {ctr['n_heavy']:4d} split {ctr['n_split']:4d} parts {ctr['n_parts']:5d} box_miss {ctr['box_miss']} | max rel diff vs first arm {diff:.2e}", flush=True)
