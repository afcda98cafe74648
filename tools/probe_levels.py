"""Which speed level does an allocation of the resident feature pool land on, and what separates the levels?

    python3 tools/probe_levels.py [--allocs 5] [--views 64] [--chunks 32,8] [--windows 1,4,16,0] [--tag NAME]

Holds --allocs copies of the R2 feature pool (32 maps, 34.8 GB each) at the same time -- so every copy sits somewhere else in
HBM -- and measures on each copy, serial phases, blocking between measurements:
  * plain streaming read (vp_stream_read),
  * k_gather per launch for --chunks views per call (HIP events of the library, one read-back per call),
  * the random whole-row gather of tools/probe_rows.hip restricted to windows of --windows GB (0 = the whole pool).
Every k_gather launch and every probe launch is appended, in launch order, to the "schedule" of the JSON written to
gpurun_out/levels/<tag>.json: run under `rocprofv3 --pmc ...` the counter CSV's k_gather / k_probe_rows dispatches pair with
it one to one (tools/levels_table.py), which turns counters into a table per allocation.
"""
import argparse
import ctypes
import json
import os
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
ap.add_argument("--views", type=int, default=64)
ap.add_argument("--chunks", default="32,8")
ap.add_argument("--windows", default="1,4,16,0")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--tag", default="levels")
ap.add_argument("--f16", action="store_true")
ap.add_argument("--out-dir", default="levels", help="sub-directory of gpurun_out/ for the JSON")
a = ap.parse_args()

dev = torch.device("cuda", 0)
n_vox, n_views, W, H, C = 200000, 300, 968, 548, 512
POOL = 32
s = make_scene(n_vox, n_views, W, H, seed=0)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
chunks = sorted((int(v) for v in a.chunks.split(",") if v), reverse=True)   # largest first: the workspace never grows
windows = [float(v) for v in a.windows.split(",") if v]
esize = 2 if a.f16 else 4
dtype = torch.float16 if a.f16 else torch.float32

probe = None
so = os.path.join(ROOT, "tools", "libprobe_rows.so")
if windows and os.path.exists(so):
    probe = ctypes.CDLL(so)
    probe.probe_rows.restype = ctypes.c_int
    probe.probe_rows.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]

pools = []
first = torch.empty((1, POOL, H, W, C), dtype=dtype, device=dev)
if a.f16:
    for v in range(POOL):
        first[0, v] = make_features_torch(1, H, W, C, dev, seed=v)[0].half()
else:
    make_features_torch(POOL, H, W, C, dev, seed=0, out=first[0])
pools.append(first)
for k in range(1, a.allocs):
    p = torch.empty_like(first)
    p.copy_(first)
    pools.append(p)
torch.cuda.synchronize()

schedule, rows = [], []
ws = voxproj_host.Workspace()
sink = torch.zeros(4, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream

# algorithmic bytes per call of each chunk size (counts from a blocking pre-pass on pool 0)
bytes_of = {}
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
out = torch.zeros(n_vox + 1, C, device=dev)
built = False
for ch in chunks:
    bytes_of[ch] = []
    for i in range(0, a.views, ch):
        count.zero_()
        voxproj_host.project_features_raw(pools[0][:, i % POOL:i % POOL + ch], occ, c2w[i:i + ch].reshape(-1).contiguous(), intr, opts, count,
                                          out, origin, s.voxel_size, workspace=ws, sync=True, reuse_accel=(built or None))
        built = True
        schedule.append({"kind": "k_gather", "alloc": -1, "chunk": ch, "call": i // ch})
        ph, nt = int(count.sum().item()), int((count > 0).sum().item())
        bytes_of[ch].append(ph * C * esize + nt * C * 4 * 2 + ch * H * W * 4 + (n_vox + 1) * 4 * 2)

for k, pool in enumerate(pools):
    o_k = torch.zeros(n_vox + 1, C, device=dev)          # the output rows move with the allocation, as in bench.py
    c_k = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
    rec = {"alloc": k, "pool_ptr": hex(pool.data_ptr()), "out_ptr": hex(o_k.data_ptr()),
           "stream_read_gbs": round(voxproj_host.stream_read_gbs(pool), 1)}
    for ch in chunks:
        vm = [c2w[i:i + ch].reshape(-1).contiguous() for i in range(0, a.views, ch)]
        ms, nl = 0.0, 0
        for rep in range(a.reps + 1):
            for ci, i in enumerate(range(0, a.views, ch)):
                voxproj_host.profile_enable(rep > 0)
                voxproj_host.project_features_raw(pool[:, i % POOL:i % POOL + ch], occ, vm[ci], intr, opts, c_k, o_k, origin, s.voxel_size,
                                                  workspace=ws, sync=False, reuse_accel=True)
                torch.cuda.synchronize()
                schedule.append({"kind": "k_gather", "alloc": k, "chunk": ch, "call": ci, "rep": rep})
                if rep > 0:
                    p = voxproj_host.profile_read()
                    ms += p["gather_ms"]; nl += p["gather_launches"]
                    schedule[-1]["gather_ms"] = round(p["gather_ms"], 4)
        voxproj_host.profile_enable(False)
        gb = sum(bytes_of[ch]) * a.reps / 1e9
        rec[f"gather_chunk{ch}"] = {"ms_per_view": round(ms / (a.views * a.reps), 5), "tbs": round(gb / ms, 4), "launches": nl}
    if probe is not None:
        row_bytes = 1024 if a.f16 else 2048
        total_rows = pool.numel() * esize // row_bytes
        for wgb in windows:
            wr = total_rows if wgb == 0 else min(total_rows, int(wgb * 1e9) // row_bytes)
            waves, iters = 65536, 64
            best = None
            for rep in range(a.reps + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = probe.probe_rows(pool.data_ptr(), 0, wr, row_bytes, waves, iters, 1234 + rep, sink.data_ptr(), stream)
                assert rc == 0, rc
                e1.record(); e1.synchronize()
                schedule.append({"kind": "k_probe_rows", "alloc": k, "window_gb": wgb, "rep": rep})
                t = e0.elapsed_time(e1)
                schedule[-1]["ms"] = round(t, 4)
                if rep > 0:
                    best = t if best is None else min(best, t)
            rec[f"rows_window{wgb:g}gb"] = {"tbs": round(waves * iters * row_bytes / best / 1e9, 4), "ms": round(best, 4)}
    rows.append(rec)
    print(json.dumps(rec), flush=True)

os.makedirs(os.path.join(ROOT, "gpurun_out", a.out_dir), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", a.out_dir, a.tag + ".json"), "w") as f:
    json.dump({"argv": sys.argv[1:], "bytes_per_call": bytes_of, "allocs": rows, "schedule": schedule,
               "when": time.strftime("%Y-%m-%d %H:%M:%S")}, f)
ws.release()
