"""Does the gather's speed level belong to the ALLOCATION of the feature pool or to the MOMENT it is measured?

    python3 tools/probe_levels_rr.py [--allocs 5] [--rounds 6] [--soak 0] [libA.so libB.so ...]

Holds --allocs copies of the R2 feature pool (32 maps, 34.8 GB each), then measures k_gather (32 views per call, serial
phases, two calls x two repetitions per cell) going ROUND-ROBIN over the copies, --rounds times, with every library given
(default: the built libvoxproj.so): a level that follows the column is the allocation's, one that follows the row is the
moment's (temperature, refresh, clocks), one that follows the library is the kernel's.  --soak S: afterwards S seconds of
back-to-back launches on copy 0, one line per second (per-launch mean), with rocm-smi's temperatures / clocks before and after.
Writes gpurun_out/levels/rr_<tag>.json and prints the matrix.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_features_torch, make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--allocs", type=int, default=5)
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--soak", type=float, default=0.0)
ap.add_argument("--workspaces", type=int, default=4)
ap.add_argument("--tag", default="rr")
ap.add_argument("--f16", action="store_true")
ap.add_argument("--pool-flags", default="", help="comma list, one per pool copy: hipExtMallocWithFlags flags (1 fine-grained, "
                "3 uncached, 4 contiguous; 0 = torch's allocator) -- does the memory TYPE of the pool reproduce a level?")
ap.add_argument("libs", nargs="*")
a = ap.parse_args()
libs = [os.path.abspath(p) for p in a.libs] or [voxproj_host.LIB_PATH]

dev = torch.device("cuda", 0)
n_vox, n_views, W, H, C = 200000, 300, 968, 548, 512
V = 32
s = make_scene(n_vox, n_views, W, H, seed=0)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
vmis = [c2w[i * V:(i + 1) * V].reshape(-1).contiguous() for i in range(2)]


def smi():
    try:
        return subprocess.run(["rocm-smi", "-t", "-c", "-P"], capture_output=True, text=True, timeout=20).stdout[-1500:]
    except Exception as e:      # diagnostics only
        return repr(e)


first = torch.empty((1, V, H, W, C), dtype=torch.float32, device=dev)
make_features_torch(V, H, W, C, dev, seed=0, out=first[0])
if a.f16:
    first = first.half()
pools = [first]
pflags = [int(v) for v in a.pool_flags.split(",") if v]
for k in range(1, a.allocs):
    fl = pflags[k] if k < len(pflags) else 0
    if fl:
        p, _ = voxproj_host.resident_empty(tuple(first.shape), first.dtype, dev, fallback=False, flags=fl)
    else:
        p = torch.empty_like(first)
    p.copy_(first)
    pools.append(p)
outs = [(torch.zeros(n_vox + 1, C, device=dev), torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)) for _ in pools]
torch.cuda.synchronize()
smi0 = smi()


def measure(lib_path, pool, o, c, wsd, w=0):
    """mean k_gather / k_first_hit ms per launch over 2 calls x 2 repetitions (one warm-up repetition first)"""
    if voxproj_host.LIB_PATH != lib_path or voxproj_host._lib is None:
        voxproj_host._lib = None
        voxproj_host.LIB_PATH = lib_path
    ws = wsd.get((lib_path, w))
    fresh = ws is None
    if fresh:
        ws = wsd[(lib_path, w)] = voxproj_host.Workspace()
    for rep in range(3):
        voxproj_host.profile_enable(rep > 0)
        for ci in range(2):
            voxproj_host.project_features_raw(pool, occ, vmis[ci], intr, opts, c, o, origin, s.voxel_size, workspace=ws, sync=False,
                                              reuse_accel=(not (fresh and rep == 0 and ci == 0)))
        torch.cuda.synchronize()
    p = voxproj_host.profile_read()
    voxproj_host.profile_enable(False)
    return p["gather_ms"] / max(p["gather_launches"], 1), p["first_hit_ms"] / max(p["first_hit_launches"], 1)


wsd, cells = {}, []
t_begin = time.perf_counter()
for r in range(a.rounds):
    for k, pool in enumerate(pools):
        for lp in libs:
            g, m = measure(lp, pool, outs[k][0], outs[k][1], wsd)
            cells.append({"round": r, "alloc": k, "lib": os.path.basename(lp), "t": round(time.perf_counter() - t_begin, 2),
                          "gather_ms": round(g, 4), "march_ms": round(m, 4)})
for lp in libs:
    print(f"== {os.path.basename(lp)}: k_gather ms per 32-view launch; rows = rounds (time), columns = allocations")
    for r in range(a.rounds):
        print("   round %d  " % r + "  ".join(f"{c['gather_ms']:.3f}" for c in cells if c["round"] == r and c["lib"] == os.path.basename(lp)), flush=True)

# which buffer carries the level: the feature pool, the output rows or the workspace (ID image, tables)?
cross = []
for lp in libs:
    print(f"== {os.path.basename(lp)}: pool k (rows) x output rows j (columns), workspace 0")
    for k, pool in enumerate(pools):
        row = []
        for j in range(len(outs)):
            g, _ = measure(lp, pool, outs[j][0], outs[j][1], wsd)
            row.append(g)
            cross.append({"lib": os.path.basename(lp), "pool": k, "out": j, "ws": 0, "gather_ms": round(g, 4)})
        print("   pool %d   " % k + "  ".join(f"{g:.3f}" for g in row), flush=True)
print(f"== {os.path.basename(libs[0])}: pool 0 / output ROWS 0 with the hit-count tensor of set j (columns), twice -- rows or counts?")
for rep in range(2):
    row = []
    for j in range(len(outs)):
        g, _ = measure(libs[0], pools[0], outs[0][0], outs[j][1], wsd)
        row.append(g)
        cross.append({"pool": 0, "out": 0, "count": j, "ws": 0, "gather_ms": round(g, 4)})
    print("   rep %d    " % rep + "  ".join(f"{g:.3f}" for g in row), flush=True)
print(f"== {os.path.basename(libs[0])}: pool 0 / output rows 0 on workspaces 0..{a.workspaces - 1} (each its own allocation), three times")
for rep in range(3):
    row = []
    for w in range(a.workspaces):
        g, _ = measure(libs[0], pools[0], outs[0][0], outs[0][1], wsd, w)
        row.append(g)
        cross.append({"pool": 0, "out": 0, "ws": w, "gather_ms": round(g, 4)})
    print("   rep %d    " % rep + "  ".join(f"{g:.3f}" for g in row), flush=True)

soak = []
if a.soak > 0:
    lp = libs[0]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < a.soak:
        t1 = time.perf_counter()
        gs = []
        while time.perf_counter() - t1 < 1.0:
            gs.append(measure(lp, pools[0], outs[0][0], outs[0][1], wsd)[0])
        soak.append({"t": round(time.perf_counter() - t0, 1), "gather_ms_mean": round(sum(gs) / len(gs), 4), "min": round(min(gs), 4),
                     "max": round(max(gs), 4)})
        print("   soak", json.dumps(soak[-1]), flush=True)
smi1 = smi()
os.makedirs(os.path.join(ROOT, "gpurun_out", "levels"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "levels", f"rr_{a.tag}.json"), "w") as f:
    json.dump({"argv": sys.argv[1:], "cells": cells, "cross": cross, "soak": soak, "smi_before": smi0, "smi_after": smi1,
               "pool_ptrs": [hex(p.data_ptr()) for p in pools]}, f, indent=1)
print("-- rocm-smi before\n" + smi0 + "\n-- rocm-smi after\n" + smi1)
for ws in wsd.values():
    ws.release()
