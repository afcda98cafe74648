"""One-off stress of the projector against the CPU oracle: N random configurations (grid, poses inside / outside / next to
voxels, intrinsics, ray range and increment, channels, views per call, fp32 / fp16 maps, plain / pipelined call sequences,
heavy thresholds from 3 pixels to none, split voxels in parts of 1 pixel upwards,
calls cut into voxel-ID ranges, one-view calls through the one-view kernel with its
default grid, with grids of one and three workgroups, and through the general kernel; one-view calls with the device's own part
sizes, with fixed split thresholds and with round 5's workgroup role).  IDs, counts and view counts must be exact, sums within 1e-4 of the oracle's
float64 accumulation (bit-identical where no heavy path can be involved).  python tools/stress_differential.py [N] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from oracle import oracle  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
dev = torch.device("cuda", 0)


def rot(rng):
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


bad = 0
for case in range(N):
    B = int(rng.integers(1, 3))
    dims = rng.integers(3, 36, 3)
    n_ids = int(rng.integers(3, 600))
    occ = np.zeros((B, *dims), np.int32)
    for b in range(B):
        n = min(n_ids, int(occ[b].size * float(np.exp(rng.uniform(np.log(0.003), np.log(0.4))))) + 1)
        idx = rng.choice(occ[b].size, n, replace=False)
        occ[b].reshape(-1)[idx] = rng.choice(n_ids, n, replace=False) + 1
    vs = float(np.float32(np.exp(rng.uniform(np.log(0.03), np.log(0.4)))))
    origin = rng.uniform(-2, 2, 3).astype(np.float32)
    ext = dims[::-1] * vs
    W, H = int(rng.integers(1, 48)), int(rng.integers(1, 36))
    C = int(rng.choice([1, 4, 8, 12, 16, 64]))
    f16 = bool(rng.integers(0, 2)) and C % 8 == 0
    pipeline = bool(rng.integers(0, 2))
    ht = int(rng.choice([3, 10, 50, 0]))
    f = float(rng.uniform(0.4, 2.2)) * W
    intr = np.stack([np.array([f, f * rng.uniform(0.8, 1.25), W * rng.uniform(0.2, 0.8), H * rng.uniform(0.2, 0.8)], np.float32) for _ in range(B)])
    dmin = float(rng.choice([0.0, 0.01, 0.25]))
    opts = np.array([W, H, dmin, float(rng.uniform(0.6, 2.5) * np.linalg.norm(ext)), float(np.float32(vs * np.exp(rng.uniform(np.log(0.2), np.log(1.8)))))], np.float32)
    n_rows = n_ids + 1
    count = np.zeros(n_rows, np.int32); out = np.zeros((n_rows, C), np.float32); out64 = np.zeros((n_rows, C)); views = np.zeros(n_rows, np.int64)
    count_t = torch.zeros(n_rows, dtype=torch.int32, device=dev); out_t = torch.zeros(n_rows, C, device=dev)
    views_t = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    occ_t = torch.from_numpy(occ.astype(np.int64)).to(dev); intr_t = torch.from_numpy(intr).to(dev)
    ws = voxproj_host.Workspace(); keep = []
    ws.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, ht or None)
    ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_GATHER, [None, None, 1001, 1003, 0][int(rng.integers(0, 5))])
    ws.set_option(voxproj_host.VP_OPT_PART_PIXELS, [None, None, 1, 2, 5, 40][int(rng.integers(0, 6))])       # parts of split voxels
    # one-view calls (round 6): the device's own sizes, round 5's workgroup role (0), fixed split thresholds
    split_opt = [None, None, None, 0, 4, 30][int(rng.integers(0, 6))]
    ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_SPLIT, split_opt)
    abs64 = np.zeros((n_rows, C))
    for call in range(int(rng.integers(1, 4))):
        V = int(rng.choice([1, 1, 1, 2, 3, 7, 8, 9, 20, 66]))
        c2w = np.zeros((B, V, 4, 4), np.float32)
        for b in range(B):
            for v in range(V):
                c2w[b, v, :3, :3] = rot(rng)
                c2w[b, v, :3, 3] = origin + rng.uniform(-0.5, 1.5, 3) * ext
                c2w[b, v, 3, 3] = 1
        feats = rng.standard_normal((B, V, H, W, C)).astype(np.float32)
        if f16:
            feats = feats.astype(np.float16).astype(np.float32)
        r = oracle.project_features(feats, occ.astype(np.int64), c2w.reshape(-1), intr, opts, origin, vs, count, out, want_f64=True)
        out64 += r["out64"]
        for b in range(B):
            for v in range(V):
                ids = np.unique(r["hits"][b, v]); views[ids[ids > 0]] += 1
                np.add.at(abs64, r["hits"][b, v].reshape(-1), np.abs(feats[b, v].reshape(-1, C)).astype(np.float64))
        ft = torch.from_numpy(feats).to(dev); ft = ft.half() if f16 else ft
        vm = torch.from_numpy(c2w).reshape(-1).to(dev); keep.append((ft, vm))
        # one call, or the same call cut into 2-3 voxel-ID ranges (VP_OPT_ROW_BEGIN/_END + VP_FLAG_GATHER_ONLY)
        cuts = [0, n_rows]
        if rng.integers(0, 3) == 0 and n_rows > 4:
            cuts = sorted({0, n_rows, *[int(c) for c in rng.integers(1, n_rows, int(rng.integers(1, 3)))]})
        for k in range(len(cuts) - 1):
            if len(cuts) > 2:
                ws.set_row_range(cuts[k], cuts[k + 1])
            voxproj_host.project_features_raw(ft, occ_t, vm, intr_t, [float(v) for v in opts], count_t, out_t, [float(v) for v in origin], vs,
                                              workspace=ws, sync=not pipeline, reuse_accel=None if k == 0 else True, pipeline=pipeline,
                                              views_hit=views_t, gather_only=k > 0)
        if len(cuts) > 2:
            ws.set_row_range()
        if not pipeline:
            got_hits = voxproj_host.hit_image(ws, dev).cpu().numpy()
            if not np.array_equal(got_hits, r["hits"]):
                bad += 1; print("case", case, "first-hit IDs differ", (got_hits != r["hits"]).sum())
    voxproj_host.workspace_status(ws, dev)
    ok = np.array_equal(count_t.cpu().numpy(), count) and np.array_equal(views_t.cpu().numpy().astype(np.int64), views)
    # sums: the forward-error bound of a float32 sum on every element (tests/sum_criteria.py: 4 sqrt(n) 2^-24 sum|addend| -- rows of 1-64
    # channels are often cancellation residues altogether, so there is no row magnitude to be relative to) and 1e-4 of the output's scale
    err = np.abs(out_t.cpu().numpy().astype(np.float64) - out64)
    abs64[0] = 0
    ok = ok and (err[1:] <= 4.0 * np.sqrt(count[1:, None].astype(np.float64)) * 2.0 ** -24 * abs64[1:]).all()
    ok = ok and err.max() <= 1e-4 * (np.abs(out64).max() + 1e-30)
    if not ht and split_opt in (None, 0):      # library thresholds: nothing of <= 256 pixels leaves the one-wavefront path
        ok = ok and (out_t.cpu().numpy().tobytes() == out.tobytes() or count.max() > 256)
    if not ok:
        bad += 1; print("case", case, "MISMATCH", dict(B=B, dims=dims.tolist(), W=W, H=H, C=C, f16=f16, pipeline=pipeline, ht=ht))
    ws.release()
print(f"{N} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
