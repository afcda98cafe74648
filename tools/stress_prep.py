"""One-off stress of the rows around the projector against their oracles: the up-sampler (random shapes, odd channel counts,
up- and down-sampling, fp16 / fp32, kept dtype), the occupancy builder (random clouds, duplicates, negative coordinates,
half-way points) and the RGB kernel (random grids, poses, images).  All bit for bit.  python tools/stress_prep.py [N] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import build_sparse_occupancy as bso  # noqa: E402
import voxproj_host  # noqa: E402
from debug_project_colors import project_colors_view  # noqa: E402
from oracle import oracle, resize_oracle as ro  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
DEV = "cuda:0"
bad = 0
for case in range(N):
    C = int(rng.choice([1, 3, 8, 16, 24, 64, 100, 512, 520, 1024, 2056]))
    h, w = int(rng.integers(1, 20)), int(rng.integers(1, 24))
    H, W = int(rng.integers(1, 40)), int(rng.integers(1, 44))
    dt = np.float16 if rng.integers(0, 2) else np.float32
    arr = (rng.standard_normal((C, h, w)) * float(rng.choice([1.0, 30.0]))).astype(dt)
    keep = bool(rng.integers(0, 2)) and dt == np.float16
    exp = ro.upsample_features(arr, H, W, keep_dtype=keep)
    got = voxproj_host.upsample_features(torch.from_numpy(arr).to(DEV), H, W, keep_dtype=keep).cpu().numpy()
    if got.tobytes() != exp.tobytes():
        bad += 1; print("upsample case", case, (C, h, w, H, W, dt.__name__, keep), "differs at", int((got != exp).sum()))
for case in range(N):
    n = int(rng.integers(1, 3000))
    vs = float(rng.choice([0.05, 0.25, 0.5, 1.0]))
    pts = (rng.integers(-20, 20, (n, 3)) * vs * float(rng.choice([1.0, 0.5])) + rng.choice([0.0, 1e-3]) * rng.standard_normal((n, 3))).astype(np.float32)
    origin = [float(v) for v in rng.uniform(-3, 3, 3).astype(np.float32)] if rng.integers(0, 2) else [float(v) for v in pts.min(0)]
    exp = oracle.build_occupancy(pts, origin, vs)
    got = bso.build_occupancy(pts, origin, vs, device=DEV).cpu().numpy()
    if exp.shape != got.shape or not np.array_equal(exp, got):
        bad += 1; print("occupancy case", case, n, vs, exp.shape, got.shape)
for case in range(N):
    dims = rng.integers(2, 20, 3)
    occ = np.zeros(dims, np.int32)
    n = max(1, int(occ.size * float(rng.uniform(0.01, 0.5))))
    occ.reshape(-1)[rng.choice(occ.size, n, replace=False)] = rng.permutation(n) + 1
    vs = float(rng.uniform(0.02, 0.4)); origin = rng.uniform(-2, 2, 3).astype(np.float32)
    q = rng.standard_normal(4); q /= np.linalg.norm(q); ww, x, y, z = q
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)], [2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)],
                   [2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)]]
    c2w[:3, 3] = origin + rng.uniform(-0.8, 1.8, 3) * dims[::-1] * vs
    iw, ih = int(rng.integers(2, 80)), int(rng.integers(2, 60))
    f = float(rng.uniform(0.3, 2.0)) * iw
    intr = np.array([f, f * rng.uniform(0.9, 1.1), rng.choice([iw / 2, iw / 2 + 0.5]), rng.choice([ih / 2, ih / 2 + 0.5])], np.float32)
    img = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
    col, zyx, uv = oracle.rgb_project(occ, c2w, intr, origin, vs, img)
    out = project_colors_view(torch.from_numpy(occ), torch.from_numpy(c2w), torch.from_numpy(intr), torch.from_numpy(origin), vs, img, device=DEV)
    if not (np.array_equal(out["projected_indices"].numpy(), zyx) and np.array_equal(out["pixel_indices"].numpy(), uv)
            and out["projected_colors"].numpy().tobytes() == col.tobytes()):
        bad += 1; print("rgb case", case, dims.tolist())
print(f"{3 * N} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
