"""What slows k_gather when another kernel shares the GPU?  Runs plain projector calls (phase 1 + gather) on one
stream with a synthetic co-runner on another and prints the gather's HIP-event time."""
import ctypes, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "3d-semantic-segmentation_amd"))
import numpy as np, torch, voxproj_host
from synthetic_scene import make_scene, make_features_torch
dev = torch.device("cuda:0")
V = 16
s = make_scene(200000, V, 968, 548, seed=0)
feats = make_features_torch(V, 548, 968, 512, dev, seed=0)[None]
occ = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
vmi = torch.from_numpy(s.c2w).reshape(-1).to(dev); intr = torch.from_numpy(s.intr[None]).to(dev)
n_rows = s.n_vox + 1
count = torch.zeros(n_rows, dtype=torch.int32, device=dev); out = torch.zeros(n_rows, 512, device=dev)
opts = [float(v) for v in s.opts()]; origin = [float(v) for v in s.grid_origin]
ws = voxproj_host.Workspace()
L = voxproj_host.lib()
L.vp_debug_spin.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
table = torch.randint(0, 1 << 20, (1 << 20,), dtype=torch.int32, device=dev)
sink = torch.zeros(8192, dtype=torch.int32, device=dev)
side = torch.cuda.Stream()
def call():
    voxproj_host.project_features_raw(feats, occ, vmi, intr, opts, count, out, origin, s.voxel_size, workspace=ws, sync=False)
call(); torch.cuda.synchronize()
for mode, blocks, iters, name in [(-1, 0, 0, "alone"), (0, 512, 60000, "valu 512wg"), (0, 2048, 60000, "valu 2048wg"),
                                  (1, 512, 20000, "l2 loads 512wg"), (1, 2048, 20000, "l2 loads 2048wg"),
                                  (2, 512, 8000, "atomics 512wg"), (2, 2048, 8000, "atomics 2048wg")]:
    voxproj_host.profile_enable(True)
    torch.cuda.synchronize()
    if mode >= 0:
        with torch.cuda.stream(side):
            L.vp_debug_spin(mode, blocks, iters, table.data_ptr(), table.numel(), sink.data_ptr(), side.cuda_stream)
        t0 = time.perf_counter()
    for _ in range(3):
        call()
    torch.cuda.current_stream().synchronize()
    p = voxproj_host.profile_read()
    e0 = torch.cuda.Event(enable_timing=True); torch.cuda.synchronize()
    print(f"{name:18s} gather {p['gather_ms']/3:7.3f} ms  first_hit {p['first_hit_ms']/3:7.3f} ms")
