"""Host-CPU probe for the cpu_baseline leg: visible cores, affinity, cgroup quota, and the oracle / torch-loop
rates at a few thread counts (run on the GPU box: python tools/probe_cpu.py)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, "-", e.__class__.__name__)
print("loadavg", open("/proc/loadavg").read().strip())

from synthetic_scene import make_features_np, make_scene  # noqa: E402
from oracle import oracle  # noqa: E402

s = make_scene(200000, 300, 968, 548, seed=0)
C, nv = 512, 2
feats = make_features_np(nv, s.height, s.width, C, seed=0)[None]
occ = s.occ[None].astype(np.int64)
for nt in (8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1):
        break
    count = np.zeros(s.n_vox + 1, np.int32)
    out = np.zeros((s.n_vox + 1, C), np.float32)
    t0 = time.perf_counter()
    oracle.project_features(feats, occ, s.c2w[:nv].reshape(-1), s.intr[None], s.opts(), s.grid_origin, s.voxel_size,
                            count, out, want_hits=False, nthreads=nt)
    dt = time.perf_counter() - t0
    print(f"oracle threads {nt:4d}: {dt:6.2f} s  {s.n_vox * nv / dt / 1e6:.3f} Mvv/s", flush=True)

from debug_project_features import voxel_centre_diagnostics  # noqa: E402
occ3 = torch.from_numpy(s.occ)
c2w = torch.from_numpy(s.c2w)
intr = torch.from_numpy(s.intr)
origin = torch.from_numpy(np.asarray(s.grid_origin, dtype=np.float32))
for nt in (1, 8, 16, 64):
    torch.set_num_threads(nt)
    voxel_centre_diagnostics(occ3, c2w[0], intr, origin, s.voxel_size, s.width, s.height)
    t0 = time.perf_counter()
    for v in range(4):
        voxel_centre_diagnostics(occ3, c2w[v], intr, origin, s.voxel_size, s.width, s.height)
    dt = time.perf_counter() - t0
    print(f"torch loop threads {nt:4d}: {dt:6.2f} s  {s.n_vox * 4 / dt / 1e6:.3f} Mvv/s", flush=True)
