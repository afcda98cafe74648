"""Does physical contiguity of the caller's feature pool decide the gather's speed level?  Alternates the pool between a
plain allocation (torch / hipMalloc) and hipExtMallocWithFlags(hipDeviceMallocContiguous) in ONE process and times a
pipelined R2 pass (16 views per call, 8 calls) on each.  python tools/probe_contig.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_features_torch, make_scene  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipFree.argtypes = [ctypes.c_void_p]


class Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f4", "data": (ptr, False), "version": 2}


def alloc(shape, contiguous):
    n = int(np.prod(shape)) * 4
    if not contiguous:
        return torch.empty(shape, dtype=torch.float32, device="cuda"), None
    p = ctypes.c_void_p()
    rc = hip.hipExtMallocWithFlags(ctypes.byref(p), n, 0x4)          # hipDeviceMallocContiguous
    if rc != 0:
        return None, rc
    return torch.as_tensor(Raw(p.value, shape), device="cuda"), p


dev = torch.device("cuda", 0)
n_vox, n_views, W, H, C, V, NCALL = 200000, 300, 968, 548, 512, int(os.environ.get('PROBE_V', '16')), int(os.environ.get('PROBE_NCALL', '8'))
s = make_scene(n_vox, n_views, W, H, seed=0)
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr[None]).to(dev)
opts = [float(v) for v in s.opts()]
origin = [float(v) for v in s.grid_origin]
vmis = [c2w[i * V:(i + 1) * V].reshape(-1).contiguous() for i in range(NCALL)]
count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
out = torch.zeros(n_vox + 1, C, dtype=torch.float32, device=dev)
ws = voxproj_host.Workspace()
for rnd in range(int(os.environ.get('PROBE_ROUNDS', '4'))):
    for contiguous in (False, True):
        feats, h = alloc((1, V, H, W, C), contiguous)
        if feats is None:
            print(f"round {rnd} contiguous allocation failed: hip error {h}")
            continue
        make_features_torch(V, H, W, C, dev, seed=0, out=feats[0])
        gbs = voxproj_host.stream_read_gbs(feats)
        for rep in range(3):
            voxproj_host.profile_enable(rep > 0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for ci in range(NCALL):
                voxproj_host.project_features_raw(feats, occ, vmis[ci], intr, opts, count, out, origin, s.voxel_size,
                                                  workspace=ws, sync=False, reuse_accel=(ci + rep + rnd > 0 or None), pipeline=True)
            voxproj_host.workspace_status(ws, dev)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        p = voxproj_host.profile_read()
        voxproj_host.profile_enable(False)
        print(f"round {rnd} {'contiguous' if contiguous else 'default   '} gather {p['gather_ms'] / max(p['gather_launches'], 1):.3f} ms/launch  "
              f"wall {dt / NCALL * 1e3:.3f} ms/call  stream read {gbs:.0f} GB/s", flush=True)
        del feats
        if h is not None:
            torch.cuda.synchronize()
            hip.hipFree(h)
        else:
            torch.cuda.empty_cache()
