"""The rows in front of the projector at the reference pipeline's own shapes, one JSON line:
  * vp_upsample_features: LSeg map fp16 [512,360,540] (script/extract_lseg_features.py:64-97 on a 1752x1168 image) ->
    876x584x512 channels-last (AGG:209 works at half resolution), float32 and float16 destinations; HIP events per call,
    achieved GB/s against the algorithmic bytes (read the map once + write the transposed copy + read it + write the result)
  * the same step as the torch expression round 1 used (interpolate + cast + cast + permute().contiguous())
  * host->device copy of one .npy-sized map (what feeds the step in the entry point)
  * vp_voxel_coords + vp_scatter_occupancy on an 87 319-voxel grid (the reference scene's size, AGG:28)
python tools/bench_prep.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_scene  # noqa: E402

dev = torch.device("cuda", 0)
C, h, w, H, W = 512, 360, 540, 584, 876
rng = np.random.default_rng(0)
arr = rng.standard_normal((C, h, w)).astype(np.float16)
src = torch.from_numpy(arr).to(dev)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
for name, keep in (("f32_dst", False), ("f16_dst", True)):
    out = torch.empty((H, W, C), dtype=torch.float16 if keep else torch.float32, device=dev)
    ms = timed(lambda: voxproj_host.upsample_features(src, H, W, keep_dtype=keep, out=out))
    algo = C * h * w * 2 * 3 + H * W * C * (2 if keep else 4)      # read src, write + read its transpose, write dst
    res["upsample_" + name] = {"ms": round(ms, 4), "algorithmic_GB": round(algo / 1e9, 3), "GBps": round(algo / ms / 1e6, 1)}


def torch_way():
    up = torch.nn.functional.interpolate(src.float()[None], size=(H, W), mode="bilinear", align_corners=False)[0]
    return up.to(torch.float16).float().permute(1, 2, 0).contiguous()


res["upsample_torch_expression_ms"] = round(timed(torch_way, reps=5), 4)
pinned = torch.from_numpy(arr).pin_memory()
res["h2d_one_map_ms"] = round(timed(lambda: src.copy_(pinned, non_blocking=True), reps=5), 3)
res["h2d_GBps"] = round(arr.nbytes / res["h2d_one_map_ms"] / 1e6, 1)
s = make_scene(87319, 2, 64, 48, seed=0)
pts = torch.from_numpy(s.points).to(dev)
t0 = time.perf_counter()
for _ in range(10):
    occ, _ = voxproj_host.build_occupancy_device(pts, s.grid_origin, s.voxel_size)
torch.cuda.synchronize()
res["occupancy_builder_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
res["occupancy_grid"] = list(occ.shape)
print(json.dumps(res))
