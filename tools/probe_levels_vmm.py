"""Is the gather's speed level carried by the VIRTUAL or by the PHYSICAL placement of the output rows?

    python3 tools/probe_levels_vmm.py [--pools 3] [--handles 3] [--vas 3]

HIP's virtual-memory management (tools/probe_vmm.hip) maps ONE physical allocation of the output rows (410 MB) at several
reserved virtual addresses, and several physical allocations at one virtual address.  For every feature pool (ordinary
allocations, held at once) the gather of a 32-view call is timed on every (physical handle, virtual address) combination:
a level that follows the row of the printed matrix is the physical memory's, one that follows the column is the address's.
Plus one ordinary torch allocation of the output rows as the reference column.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_features_torch, make_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pools", type=int, default=3)
ap.add_argument("--handles", type=int, default=3)
ap.add_argument("--vas", type=int, default=3)
a = ap.parse_args()

dev = torch.device("cuda", 0)
V = ctypes.CDLL(os.path.join(ROOT, "tools", "libprobe_vmm.so"))
V.vmm_granularity.restype = ctypes.c_longlong
u64 = ctypes.c_ulonglong
n_vox, n_views, W, H, C = 200000, 300, 968, 548, 512
NV = 32
s = make_scene(n_vox, n_views, W, H, seed=0)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
vmis = [c2w[i * NV:(i + 1) * NV].reshape(-1).contiguous() for i in range(2)]
first = torch.empty((1, NV, H, W, C), dtype=torch.float32, device=dev)
make_features_torch(NV, H, W, C, dev, seed=0, out=first[0])
pools = [first]
for _ in range(1, a.pools):
    p = torch.empty_like(first); p.copy_(first); pools.append(p)
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
ref_out = torch.zeros(n_vox + 1, C, device=dev)
torch.cuda.synchronize()

gran = V.vmm_granularity(0)
assert gran > 0, gran
nbytes = ((n_vox + 1) * C * 4 + gran - 1) // gran * gran
handles, vas = [], []
for _ in range(a.handles):
    h = u64()
    rc = V.vmm_create(0, ctypes.c_longlong(nbytes), ctypes.byref(h)); assert rc == 0, rc
    handles.append(h.value)
for k in range(a.vas):
    p = u64()
    rc = V.vmm_reserve(ctypes.c_longlong(nbytes), ctypes.c_longlong(1 << 30 if k % 2 == 0 else 0), ctypes.byref(p)); assert rc == 0, rc
    vas.append(p.value)
print(f"granularity {gran} B, {nbytes} B per output buffer; virtual addresses " + " ".join(hex(v) for v in vas), flush=True)


class Mapped:
    """The mapped range as a float32 [n_rows, C] tensor through the CUDA array interface."""
    def __init__(self, ptr):
        self.__cuda_array_interface__ = {"shape": (n_vox + 1, C), "typestr": "<f4", "data": (ptr, False), "version": 2}


ws = voxproj_host.Workspace()
state = {"built": False}


def measure(pool, out):
    for rep in range(3):
        voxproj_host.profile_enable(rep > 0)
        for ci in range(2):
            voxproj_host.project_features_raw(pool, occ, vmis[ci], intr, opts, count, out, origin, s.voxel_size, workspace=ws, sync=False,
                                              reuse_accel=state["built"])
            state["built"] = True
        torch.cuda.synchronize()
    p = voxproj_host.profile_read()
    voxproj_host.profile_enable(False)
    return p["gather_ms"] / max(p["gather_launches"], 1)


for k, pool in enumerate(pools):
    print(f"== pool {k}: rows = physical allocation of the output rows, columns = virtual address it is mapped at; "
          f"reference (ordinary allocation): {measure(pool, ref_out):.3f} ms")
    for hi, h in enumerate(handles):
        row = []
        for va in vas:
            rc = V.vmm_map(0, u64(va), ctypes.c_longlong(nbytes), u64(h)); assert rc == 0, rc
            holder = Mapped(va)
            out = torch.as_tensor(holder, device=dev)
            out.zero_()
            row.append(measure(pool, out))
            torch.cuda.synchronize()
            del out
            rc = V.vmm_unmap(u64(va), ctypes.c_longlong(nbytes)); assert rc == 0, rc
        print(f"   handle {hi}   " + "  ".join(f"{t:.3f}" for t in row), flush=True)
for va in vas:
    V.vmm_free(u64(va), ctypes.c_longlong(nbytes))
for h in handles:
    V.vmm_release(u64(h))
ws.release()
