"""The named entry point end to end ON FILES, at the reference pipeline's own sizes (AGG:28,106,209: 87 319 voxels, DSLR
images 1752x1168 worked at 0.5x = 876x584, LSeg maps fp16 [512,360,540] = 199 MB per view): writes N synthetic .npy maps, a
voxel-grid PLY and a camera JSON into a scratch directory and times aggregate_voxel_features_onthefly.main() per view, with
and without the feature feeder (--prefetch), in both modes.  One JSON line.
python tools/bench_entry_files.py [N] [prefetch depths, e.g. 0,3] [trajectory]"""
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import aggregate_voxel_features_onthefly as agg  # noqa: E402
from synthetic_scene import make_scene  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
PREFETCH = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 3]     # feeder depths to time
TRAJECTORY = len(sys.argv) > 3 and sys.argv[3] == "trajectory"     # the A1 leg's hand-held trajectory (0.04-m cells) instead of the benign room
C, h, w = 512, 360, 540
s = (make_scene(87319, N, 876, 584, seed=0, trajectory=True, room=(4.4, 3.5, 2.5), voxel_size=0.04) if TRAJECTORY
     else make_scene(87319, N, 876, 584, seed=0))
tmp = tempfile.mkdtemp(prefix="vp_entry_")
try:
    ply = os.path.join(tmp, f"scene_{s.n_vox}vox_grid.ply")
    with open(ply, "w") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write(f"comment voxel_size {s.voxel_size!r}\ncomment grid_origin {float(s.grid_origin[0])!r} "
                f"{float(s.grid_origin[1])!r} {float(s.grid_origin[2])!r}\n")
        f.write(f"element vertex {s.n_vox}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        np.savetxt(f, s.points.astype(np.float64), fmt="%.9g")
    lseg = os.path.join(tmp, "features")
    os.makedirs(lseg)
    rng = np.random.default_rng(0)
    base = rng.standard_normal((C, h, w)).astype(np.float16)
    images = {}
    for v in range(N):
        name = f"DSC{v:05d}.JPG"
        np.save(os.path.join(lseg, name + ".npy"), np.roll(base, v, axis=2))
        c2w = s.c2w[v].astype(np.float64)
        R = c2w[:3, :3].T
        images[str(v)] = {"name": name, "camera_id": 1, "R": R.tolist(), "tvec": (-R @ c2w[:3, 3]).tolist()}
    cams = {"1": {"params": [float(x) * 2 for x in s.intr], "width": 1752, "height": 1168}}
    cam_json = os.path.join(tmp, "camera_params.json")
    with open(cam_json, "w") as f:
        json.dump({"images": images, "cameras": cams}, f)
    res = {"views": N, "voxels": s.n_vox, "map": [C, h, w], "working_resolution": [584, 876]}
    common = ["--lseg_dir", lseg, "--cam_params", cam_json, "--voxel_ply", ply]
    agg.main(["--mode", "fast", "--checkpoint_dir", os.path.join(tmp, "warm"), "--max_images", "2"] + common)     # warm-up, page cache
    loops = {}

    def run(mode, pf, n, out):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tm = {}
        agg.main(["--mode", mode, "--checkpoint_dir", out, "--prefetch", str(pf), "--max_images", str(n)] + common, timing=tm)
        torch.cuda.synchronize()
        loops[(mode, pf, n)] = tm
        return time.perf_counter() - t0

    for mode in ("parity", "fast"):
        for pf in PREFETCH:
            out = os.path.join(tmp, f"{mode}_{pf}")
            few = run(mode, pf, max(2, N // 4), out + "_few")       # fixed costs (PLY parse, tables, final files) cancel
            t = run(mode, pf, N, out)
            res[f"{mode}_prefetch{pf}_ms_per_view"] = round((t - few) / (N - max(2, N // 4)) * 1e3, 2)
            res[f"{mode}_prefetch{pf}_total_s"] = round(t, 2)
            # the per-view loop alone (first file requested .. last call drained): no PLY parse, no result files -- the
            # difference figure above also carries the result files, whose size grows with the views
            tm = loops[(mode, pf, N)]
            res[f"{mode}_prefetch{pf}_loop_ms_per_view"] = round(tm["loop_s"] / tm["views"] * 1e3, 2)
        a = torch.load(os.path.join(tmp, f"{mode}_{PREFETCH[0]}", f"ALL_nonzero_voxel_features_{N}_vox{s.n_vox}.pt"))
        b = torch.load(os.path.join(tmp, f"{mode}_{PREFETCH[-1]}", f"ALL_nonzero_voxel_features_{N}_vox{s.n_vox}.pt"))
        res[f"{mode}_same_result"] = bool(torch.equal(a["avg_feats"], b["avg_feats"]) and torch.equal(a["voxel_coords"], b["voxel_coords"]))
    print(json.dumps(res))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
