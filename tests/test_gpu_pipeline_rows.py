"""GPU parity for the rows around the kernel: the single-view driver, the
aggregator entry point, and the RGB path -- against the committed goldens (reference-generated for RGB,
oracle-generated for the aggregator) and the live oracle."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _no_heavy_path_by_default(heavy_threshold):
    heavy_threshold(100000000)


def test_hip_path_matches_committed_s1_golden():
    import voxproj_host
    from make_oracle_goldens import S1, s1_inputs
    g = np.load(os.path.join(HERE, "golden", "s1_oracle_golden.npz"))
    s, feats = s1_inputs()
    n_rows, C = s.n_vox + 1, S1["channels"]
    dev = torch.device(DEV)
    count = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    sums = torch.zeros(n_rows, C, device=dev)
    ws = voxproj_host.project_features_raw(
        torch.from_numpy(feats[None]).to(dev), torch.from_numpy(s.occ[None].astype(np.int64)).to(dev),
        torch.from_numpy(s.c2w).reshape(-1).to(dev), torch.from_numpy(s.intr[None]).to(dev),
        [float(v) for v in s.opts()], count, sums, [float(v) for v in s.grid_origin], s.voxel_size, sync=True)
    assert np.array_equal(voxproj_host.hit_image(ws, dev).cpu().numpy()[0], g["hits"])
    assert np.array_equal(count.cpu().numpy(), g["count"])
    assert sums.cpu().numpy().tobytes() == g["sums"].tobytes()


def test_single_view_driver_matches_oracle(oracle_mod):
    from debug_project_features import project_view
    from make_oracle_goldens import S1, s1_inputs
    s, feats = s1_inputs()
    n_rows, C = s.n_vox + 1, S1["channels"]
    for v in (0, 5):
        out = project_view(torch.from_numpy(feats[None, v:v + 1]), torch.from_numpy(s.occ), torch.from_numpy(s.intr)[None, None],
                           torch.from_numpy(s.c2w[v])[None, None], torch.from_numpy(s.grid_origin), s.voxel_size, device=DEV)
        c1 = np.zeros(n_rows, np.int32)
        s1 = np.zeros((n_rows, C), np.float32)
        oracle_mod.project_features(feats[None, v:v + 1], s.occ[None].astype(np.int64), s.c2w[v].reshape(-1), s.intr[None],
                                    s.opts(), s.grid_origin, s.voxel_size, c1, s1)
        ef, ei = oracle_mod.dpf_select_outputs(s.occ, c1, s1)
        assert out["projected_feats"].dtype == torch.float16 and out["projected_indices"].dtype == torch.int32
        assert np.array_equal(out["projected_indices"].numpy(), ei)
        assert out["projected_feats"].numpy().tobytes() == ef.tobytes()


def test_aggregator_parity_mode_matches_golden():
    # reference semantics: per-view fp16 sums, fp16 running sum, hit_count = views, insertion order (AGG:307-313,381-451)
    from aggregate_voxel_features_onthefly import VoxelFeatureAggregator
    from make_oracle_goldens import S1, s1_inputs
    g = np.load(os.path.join(HERE, "golden", "s1_oracle_golden.npz"))
    s, feats = s1_inputs()
    agg = VoxelFeatureAggregator(torch.from_numpy(s.occ), s.grid_origin.astype(np.float64), s.voxel_size, S1["channels"], "parity", DEV)
    f = torch.from_numpy(feats).to(DEV)
    agg.add_views(f[:3], torch.from_numpy(s.c2w[:3]), torch.from_numpy(s.intr))
    agg.add_views(f[3:], torch.from_numpy(s.c2w[3:]), torch.from_numpy(s.intr))
    r = agg.result()
    assert np.array_equal(r["voxel_coords"].numpy(), g["agg_coords"])
    assert np.array_equal(r["hit_count"].numpy(), g["agg_hits"])
    assert r["avg_feats"].dtype == torch.float16 and r["avg_feats"].numpy().tobytes() == g["agg_avg"].tobytes()
    assert r["xyz"].dtype == torch.float32 and r["xyz"].numpy().tobytes() == g["agg_xyz"].tobytes()


def test_aggregator_parity_mode_keeps_host_poses_alive_one_view_per_call():
    """the first add_views of a parity aggregator (and every one that flushes) used to enter its freshly staged
    poses in the keep-alive list BEFORE the flush that empties it; the one-view calls are pipelined, so the library's side
    stream -- which torch's allocator does not know -- could still read a pose block that the next call's staging copy had
    already been handed.  Every view through its own add_views with a HOST pose, back to back: the poses of the call in
    progress are held after it returns, and the file is the golden's bit for bit."""
    from aggregate_voxel_features_onthefly import VoxelFeatureAggregator
    from make_oracle_goldens import S1, s1_inputs
    g = np.load(os.path.join(HERE, "golden", "s1_oracle_golden.npz"))
    s, feats = s1_inputs()
    for rep in range(3):
        agg = VoxelFeatureAggregator(torch.from_numpy(s.occ), s.grid_origin.astype(np.float64), s.voxel_size, S1["channels"], "parity", DEV)
        f = torch.from_numpy(feats).to(DEV)
        for v in range(feats.shape[0]):
            agg.add_views(f[v:v + 1], torch.from_numpy(s.c2w[v:v + 1].copy()), torch.from_numpy(s.intr))
            held = [k[0].data_ptr() for k in agg._keep]
            assert len(held) == v + 1 and len(set(held)) == v + 1          # this call's poses included, every block distinct
        r = agg.result()
        assert np.array_equal(r["voxel_coords"].numpy(), g["agg_coords"]) and np.array_equal(r["hit_count"].numpy(), g["agg_hits"])
        assert r["avg_feats"].numpy().tobytes() == g["agg_avg"].tobytes()


def test_aggregator_fast_mode_is_the_exact_fp32_version(oracle_mod):
    from aggregate_voxel_features_onthefly import VoxelFeatureAggregator
    from make_oracle_goldens import S1, s1_inputs
    s, feats = s1_inputs()
    n_rows, C = s.n_vox + 1, S1["channels"]
    agg = VoxelFeatureAggregator(torch.from_numpy(s.occ), s.grid_origin.astype(np.float64), s.voxel_size, C, "fast", DEV)
    f = torch.from_numpy(feats).to(DEV)
    for a, b in ((0, 3), (3, 4), (4, 8)):
        agg.add_views(f[a:b], torch.from_numpy(s.c2w[a:b]), torch.from_numpy(s.intr))
    r = agg.result()
    count = np.zeros(n_rows, np.int32)
    sums = np.zeros((n_rows, C), np.float32)
    views = np.zeros(n_rows, np.int32)
    for a, b in ((0, 3), (3, 4), (4, 8)):
        oracle_mod.project_features(feats[None, a:b], s.occ[None].astype(np.int64), s.c2w[a:b].reshape(-1), s.intr[None],
                                    s.opts(), s.grid_origin, s.voxel_size, count, sums)
    for v in range(8):
        c1 = np.zeros(n_rows, np.int32)
        oracle_mod.project_features(feats[None, v:v + 1], s.occ[None].astype(np.int64), s.c2w[v].reshape(-1), s.intr[None],
                                    s.opts(), s.grid_origin, s.voxel_size, c1, np.zeros((n_rows, C), np.float32))
        views += c1 > 0
    ids = r["voxel_ids"].numpy()
    assert np.array_equal(ids, np.nonzero(views > 0)[0])
    assert np.array_equal(r["count"].numpy(), count[ids]) and np.array_equal(r["hit_count"].numpy(), views[ids])
    assert r["sum"].numpy().tobytes() == sums[ids].tobytes()
    exp_avg = (sums[ids] / views[ids].astype(np.float32)[:, None]).astype(np.float16)
    assert r["avg_feats"].numpy().tobytes() == exp_avg.tobytes()


def test_rgb_kernel_matches_reference_debug_project_colors():
    from debug_project_colors import project_colors_view
    g = np.load(os.path.join(HERE, "golden", "reference_python_goldens.npz"))
    for v in range(3):
        out = project_colors_view(torch.from_numpy(g["occ"]), torch.from_numpy(g["c2w"][v]), torch.from_numpy(g["intr"]),
                                  torch.from_numpy(g["grid_origin"]), float(g["voxel_size"]), g[f"img{v}"], device=DEV)
        assert np.array_equal(out["projected_indices"].numpy(), g[f"zyx{v}"])
        assert np.array_equal(out["pixel_indices"].numpy(), g[f"uv{v}"])
        assert out["projected_colors"].numpy().tobytes() == g[f"colors{v}"].tobytes()


def test_rgb_aggregator_matches_reference_dict_semantics():
    # aggregate_voxel_colors_onthefly.py:134-140,182-218 replayed on the reference-generated per-view outputs
    from aggregate_voxel_colors_onthefly import VoxelColorAggregator
    g = np.load(os.path.join(HERE, "golden", "reference_python_goldens.npz"))
    sums, counts = {}, {}
    for v in range(3):
        for idx, col in zip(g[f"zyx{v}"], g[f"colors{v}"]):
            k = tuple(int(t) for t in idx)
            sums[k] = col.copy() if k not in sums else (sums[k] + col).astype(np.float32)
            counts[k] = counts.get(k, 0) + 1
    keys = list(sums)
    exp_avg = np.stack([sums[k] / np.float32(counts[k]) for k in keys]).astype(np.float32)
    agg = VoxelColorAggregator(torch.from_numpy(g["occ"]), g["grid_origin"].astype(np.float64), float(g["voxel_size"]), DEV)
    imgs = torch.from_numpy(np.stack([g[f"img{v}"] for v in range(3)]))
    intr = torch.from_numpy(g["intr"])[None].repeat(3, 1)
    agg.add_views(imgs[:2], torch.from_numpy(g["c2w"][:2]), intr[:2])
    agg.add_views(imgs[2:], torch.from_numpy(g["c2w"][2:]), intr[2:])
    r = agg.result()
    assert np.array_equal(r["voxel_coords"].numpy(), np.array(keys, np.int32))
    assert np.array_equal(r["hit_count"].numpy(), np.array([counts[k] for k in keys]))
    assert r["avg_color"].numpy().tobytes() == exp_avg.tobytes()


def _write_scene_files(tmp_path, n_views=5, C=16):
    """Files laid out like the reference's inputs: voxel-grid PLY with header comments (AGG:65-92), camera JSON
    (PTD:56-72), fp16 .npy features [C,h,w] (script/extract_lseg_features.py:97).  Returns (scene, ply, lseg_dir, cam_json)."""
    import json

    from synthetic_scene import make_scene
    s = make_scene(1500, n_views, 48, 32, seed=91, room=(5.0, 4.0, 2.4))
    ply = tmp_path / f"scene_{s.n_vox}vox_grid.ply"
    with open(ply, "w") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write(f"comment voxel_size {s.voxel_size!r}\ncomment grid_origin {float(s.grid_origin[0])!r} "
                f"{float(s.grid_origin[1])!r} {float(s.grid_origin[2])!r}\n")
        f.write(f"element vertex {s.n_vox}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        for q in s.points:
            f.write(f"{float(q[0])!r} {float(q[1])!r} {float(q[2])!r}\n")
    lseg = tmp_path / "features"
    lseg.mkdir()
    rng = np.random.default_rng(91)
    images = {}
    for v in range(n_views):
        name = f"DSC{v:05d}.JPG"
        np.save(lseg / f"{name}.npy", rng.standard_normal((C, 16, 24)).astype(np.float16))      # half-res map, upsampled x2
        c2w = s.c2w[v].astype(np.float64)
        R = c2w[:3, :3].T                                   # PTD:165-172 inverts [R|t]
        t = -R @ c2w[:3, 3]
        images[str(v)] = {"name": name, "camera_id": 1, "R": R.tolist(), "tvec": t.tolist()}
    # the entry point scales intrinsics by 0.5 (AGG:209, PTD:132-143): give it the double-resolution camera
    cams = {"1": {"params": [float(x) * 2 for x in s.intr], "width": 96, "height": 64}}
    cam_json = tmp_path / "camera_params.json"
    cam_json.write_text(json.dumps({"images": images, "cameras": cams}))
    return s, ply, lseg, cam_json


def test_entry_point_end_to_end_on_synthetic_files(tmp_path):
    # aggregate_voxel_features_onthefly.main() on files: parity / fast / fast+fp16 modes must agree where they should.
    import aggregate_voxel_features_onthefly as agg
    s, ply, lseg, cam_json = _write_scene_files(tmp_path)
    outs = {}
    for mode, extra in (("parity", []), ("fast", []), ("fast16", ["--half_features"])):
        out_dir = tmp_path / mode
        agg.main(["--mode", mode.replace("16", ""), "--lseg_dir", str(lseg), "--cam_params", str(cam_json), "--voxel_ply", str(ply),
                  "--checkpoint_dir", str(out_dir), "--views_per_call", "2"] + extra)
        f = out_dir / f"ALL_nonzero_voxel_features_5_vox{s.n_vox}.pt"
        outs[mode] = torch.load(f)
        assert set(outs[mode]) == {"xyz", "avg_feats", "voxel_coords"}
        assert outs[mode]["avg_feats"].dtype == torch.float16 and outs[mode]["voxel_coords"].dtype == torch.int32
    assert outs["parity"]["xyz"].shape[0] > 300
    # fast and fast+fp16 are the same numbers; parity differs only by the reference's fp16 round trips (and row order)
    assert torch.equal(outs["fast"]["avg_feats"], outs["fast16"]["avg_feats"])
    assert torch.equal(outs["fast"]["voxel_coords"], outs["fast16"]["voxel_coords"])
    key = lambda t: (t[:, 0].long() * 10**6 + t[:, 1].long() * 10**3 + t[:, 2].long())
    pa, fa = torch.argsort(key(outs["parity"]["voxel_coords"])), torch.argsort(key(outs["fast"]["voxel_coords"]))
    assert torch.equal(outs["parity"]["voxel_coords"][pa], outs["fast"]["voxel_coords"][fa])
    a, b = outs["parity"]["avg_feats"][pa].float(), outs["fast"]["avg_feats"][fa].float()
    assert (a - b).abs().max() <= 2e-2 * b.abs().max()


def test_entry_point_two_ranks_match_one(tmp_path):
    # torchrun with two ranks (views r::2, one all-reduce of {sum, count, views}); gloo on one GPU rehearses the RCCL path.
    # Integer outputs equal the single-process run's, fp32 sums differ only by the order of the two partial sums.
    import subprocess

    import aggregate_voxel_features_onthefly as agg
    s, ply, lseg, cam_json = _write_scene_files(tmp_path, n_views=7)
    common = ["--mode", "fast", "--lseg_dir", str(lseg), "--cam_params", str(cam_json), "--voxel_ply", str(ply), "--views_per_call", "2"]
    agg.main(common + ["--checkpoint_dir", str(tmp_path / "one")])
    env = dict(os.environ, VOXPROJ_SINGLE_DEVICE="1", VOXPROJ_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    script = os.path.join(ROOT, "3d-semantic-segmentation_amd", "aggregate_voxel_features_onthefly.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", script] + common + ["--checkpoint_dir", str(tmp_path / "two")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    name = f"ALL_nonzero_voxel_features_7_vox{s.n_vox}_fp32.pt"
    one, two = torch.load(tmp_path / "one" / name), torch.load(tmp_path / "two" / name)
    for k in ("voxel_ids", "voxel_coords", "count", "hit_count"):
        assert torch.equal(one[k], two[k]), k
    assert (one["sum"] - two["sum"]).abs().max() <= 1e-5 * one["sum"].abs().max()
    assert (one["avg_feats"].float() - two["avg_feats"].float()).abs().max() <= 2e-3 * one["avg_feats"].float().abs().max()


def test_nearest_voxel_map_matches_reference_kdtree():
    # golden = the reference's own map_gaussians_to_voxels (sklearn KDTree, k=1) run in the build container
    from make_stage5_golden import inputs
    from voxel_to_gaussian_map import map_gaussians_to_voxels
    g = np.load(os.path.join(HERE, "golden", "nearest_voxel_golden.npz"))
    vox, mu = inputs()
    assert vox.astype(np.float64).sum() == float(g["vox_checksum"]) and mu.astype(np.float64).sum() == float(g["mu_checksum"])
    got = map_gaussians_to_voxels(torch.from_numpy(vox), torch.from_numpy(mu), device=DEV)
    assert got.dtype == torch.int64 and got.device.type == "cpu" and got.shape == (len(mu),)
    got = got.numpy()
    ref = g["idx"]
    diff = np.nonzero(got != ref)[0]
    # where the index differs the two voxels must be exactly equidistant (sklearn's tie choice is unspecified)
    d_got = ((mu[diff].astype(np.float64) - vox[got[diff]].astype(np.float64)) ** 2).sum(1)
    d_ref = ((mu[diff].astype(np.float64) - vox[ref[diff]].astype(np.float64)) ** 2).sum(1)
    assert np.array_equal(d_got, d_ref)
    assert len(diff) < 0.01 * len(ref)
    assert np.array_equal(got[-300:], np.arange(300))            # centres sitting exactly on voxels map to them


def test_color_entry_point_end_to_end(tmp_path, oracle_mod):
    # aggregate_voxel_colors_onthefly.main() on files: PLY + camera JSON + PNG images (+ empty .npy names, AGGC:62)
    import json

    from PIL import Image

    import aggregate_voxel_colors_onthefly as aggc
    from synthetic_scene import make_scene
    s = make_scene(1500, 3, 60, 40, seed=92, room=(5.0, 4.0, 2.4))
    ply = tmp_path / f"scene_{s.n_vox}vox_grid.ply"
    with open(ply, "w") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write(f"comment voxel_size {s.voxel_size!r}\ncomment grid_origin {float(s.grid_origin[0])!r} "
                f"{float(s.grid_origin[1])!r} {float(s.grid_origin[2])!r}\n")
        f.write(f"element vertex {s.n_vox}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        for q in s.points:
            f.write(f"{float(q[0])!r} {float(q[1])!r} {float(q[2])!r}\n")
    (tmp_path / "features").mkdir()
    (tmp_path / "images").mkdir()
    rng = np.random.default_rng(92)
    images, imgs = {}, []
    for v in range(3):
        name = f"IMG{v:03d}"
        np.save(tmp_path / "features" / f"{name}.npy", np.zeros((1, 1, 1), np.float16))
        img = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
        Image.fromarray(img).save(tmp_path / "images" / f"{name}.png")
        imgs.append(img)
        c2w = s.c2w[v].astype(np.float64)
        R = c2w[:3, :3].T
        images[str(v)] = {"name": name, "camera_id": 1, "R": R.tolist(), "tvec": (-R @ c2w[:3, 3]).tolist()}
    cam_json = tmp_path / "cams.json"
    cam_json.write_text(json.dumps({"images": images, "cameras": {"1": {"params": [float(x) for x in s.intr]}}}))
    out_dir = tmp_path / "out"
    aggc.main(["--lseg_dir", str(tmp_path / "features"), "--images_dir", str(tmp_path / "images"), "--cam_params", str(cam_json),
               "--voxel_ply", str(ply), "--checkpoint_dir", str(out_dir), "--views_per_call", "2"])
    r = torch.load(out_dir / f"ALL_nonzero_voxel_colors_3_vox{s.n_vox}.pt")
    assert set(r) == {"xyz", "avg_color", "hit_count", "voxel_coords"}
    # the same aggregate from the oracle's per-view RGB projection (dict semantics of AGGC:134-140,182-186)
    import build_sparse_occupancy as bso
    occ = bso.build_occupancy(bso.read_voxel_ply(str(ply)), [float(x) for x in s.grid_origin], s.voxel_size).numpy()
    sums, counts = {}, {}
    for v in range(3):
        R = np.array(images[str(v)]["R"], np.float32)
        t = np.array(images[str(v)]["tvec"], np.float32)
        c2w = np.eye(4, dtype=np.float32)
        c2w[:3, :3] = R.T
        c2w[:3, 3] = -R.T @ t                                  # prepare_tensor_data.py:165-172
        col, zyx, _ = oracle_mod.rgb_project(occ, c2w, s.intr, s.grid_origin, float(s.voxel_size), imgs[v])
        for k, c in zip(map(tuple, zyx.tolist()), col):
            sums[k] = c.copy() if k not in sums else (sums[k] + c).astype(np.float32)
            counts[k] = counts.get(k, 0) + 1
    keys = list(sums)
    assert np.array_equal(r["voxel_coords"].numpy(), np.array(keys, np.int32))
    assert np.array_equal(r["hit_count"].numpy(), np.array([counts[k] for k in keys]))
    exp = np.stack([sums[k] / np.float32(counts[k]) for k in keys]).astype(np.float32)
    assert r["avg_color"].numpy().tobytes() == exp.tobytes()


def test_rgb_kernel_randomized_against_the_oracle(oracle_mod):
    # debug_project_colors.py:54-81 semantics on random grids, poses (cameras inside, outside and behind voxels),
    # intrinsics and image sizes: voxel order, rounded pixel coordinates and colours bit for bit (float64 math,
    # half-to-even rounding, image-bounds rejection, cam.z <= 0 skipped)
    from debug_project_colors import project_colors_view
    rng = np.random.default_rng(4242)
    seen = 0
    for case in range(40):
        dims = rng.integers(2, 24, 3)
        occ = np.zeros(dims, np.int32)
        n = max(1, int(occ.size * float(rng.uniform(0.01, 0.4))))
        idx = rng.choice(occ.size, n, replace=False)
        occ.reshape(-1)[idx] = rng.permutation(n) + 1
        vs = float(rng.uniform(0.02, 0.4))
        origin = rng.uniform(-2, 2, 3).astype(np.float32)
        ext = dims[::-1] * vs
        q = rng.standard_normal(4); q /= np.linalg.norm(q)
        w, x, y, z = q
        c2w = np.eye(4, dtype=np.float32)
        c2w[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
        c2w[:3, 3] = origin + rng.uniform(-0.8, 1.8, 3) * ext
        iw, ih = int(rng.integers(2, 90)), int(rng.integers(2, 70))
        f = float(rng.uniform(0.3, 2.0)) * iw
        # half-integer principal points put many projections exactly on .5 (banker's rounding matters)
        intr = np.array([f, f * rng.uniform(0.9, 1.1), rng.choice([iw / 2, iw / 2 + 0.5]), rng.choice([ih / 2, ih / 2 + 0.5])], np.float32)
        img = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
        colors, zyx, uv = oracle_mod.rgb_project(occ, c2w, intr, origin, vs, img)
        out = project_colors_view(torch.from_numpy(occ), torch.from_numpy(c2w), torch.from_numpy(intr), torch.from_numpy(origin),
                                  vs, img, device=DEV)
        assert np.array_equal(out["projected_indices"].numpy(), zyx), case
        assert np.array_equal(out["pixel_indices"].numpy(), uv), case
        assert out["projected_colors"].numpy().tobytes() == colors.tobytes(), case
        seen += len(zyx) > 0
    assert seen > 15


def test_nearest_voxel_map_randomized_against_brute_force():
    # exhaustive float64 nearest neighbour (what sklearn's KDTree with k=1 returns, voxeltoGaussian_logits.py:87-96) on
    # random clouds: clustered and sparse voxel sets, queries far outside the voxel bounding box, single-voxel sets
    from voxel_to_gaussian_map import map_gaussians_to_voxels
    rng = np.random.default_rng(99)
    for case in range(12):
        n = int(rng.choice([1, 2, 17, 500, 4000]))
        m = int(rng.choice([1, 33, 3000]))
        scale = float(rng.choice([0.05, 1.0, 40.0]))
        if case % 3 == 0:       # voxel-like: integer lattice points times a voxel size
            vox = (rng.integers(-30, 30, (n, 3)) * 0.032 * scale).astype(np.float32)
            vox = np.unique(vox, axis=0)
        else:
            vox = (rng.standard_normal((n, 3)) * scale).astype(np.float32)
        mu = (rng.standard_normal((m, 3)) * scale * float(rng.choice([0.5, 1.0, 6.0]))).astype(np.float32)
        got = map_gaussians_to_voxels(torch.from_numpy(vox), torch.from_numpy(mu), device=DEV).numpy()
        d = ((mu[:, None, :].astype(np.float64) - vox[None, :, :].astype(np.float64)) ** 2).sum(2)
        best = d.min(1)
        assert got.shape == (m,) and got.min() >= 0 and got.max() < len(vox)
        assert np.array_equal(d[np.arange(m), got], best), case       # a true nearest neighbour (ties may differ in index)


def test_integration_md_binding_stub_runs(oracle_mod):
    # the ctypes stub printed in INTEGRATION.md section 2 is executed as it stands (library path resolved) and must give
    # the oracle's counts and sums
    import re

    from synthetic_scene import make_features_np, make_scene
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", txt, flags=re.S)
    stub = next(b for b in blocks if "ctypes.CDLL" in b)
    stub = stub.replace('ctypes.CDLL("libvoxproj.so")',
                        f'ctypes.CDLL({os.path.join(ROOT, "3d-semantic-segmentation_amd", "libvoxproj.so")!r})')
    ns = {}
    exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    s = make_scene(2000, 3, 40, 24, seed=71, room=(5.0, 4.0, 2.4))
    C = 8
    feats = make_features_np(3, 24, 40, C, seed=71)[None]
    n_rows = s.n_vox + 1
    count = np.zeros(n_rows, np.int32)
    out = np.zeros((n_rows, C), np.float32)
    oracle_mod.project_features(feats, s.occ[None].astype(np.int64), s.c2w.reshape(-1), s.intr[None], s.opts(), s.grid_origin,
                                s.voxel_size, count, out)
    count_t = torch.zeros(n_rows, dtype=torch.int32, device=DEV)
    out_t = torch.zeros(n_rows, C, device=DEV)
    ns["project_features_cuda"](torch.from_numpy(feats).to(DEV), torch.from_numpy(s.occ[None].astype(np.int64)).to(DEV),
                                torch.from_numpy(s.c2w).reshape(-1).to(DEV), torch.from_numpy(s.intr[None]).to(DEV),
                                torch.from_numpy(s.opts()), count_t, out_t, torch.tensor([False]), torch.from_numpy(s.grid_origin),
                                float(s.voxel_size))
    assert np.array_equal(count_t.cpu().numpy(), count) and count.sum() > 0
    assert out_t.cpu().numpy().tobytes() == out.tobytes()


def test_c_abi_from_a_plain_hip_program(tmp_path, oracle_mod):
    # examples/abi_example.cpp: libvoxproj.so driven from C++ with hipMalloc'd buffers -- no torch in the process.
    # Its printed counts and sums (two accumulating calls on the K1 wall scene) must equal the oracle's.
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    pkg = os.path.join(ROOT, "3d-semantic-segmentation_amd")
    exe = tmp_path / "abi_example"
    subprocess.run([hipcc, "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "abi_example.cpp"),
                    "-L" + pkg, "-lvoxproj", "-Wl,-rpath," + pkg, "-o", str(exe)], check=True, timeout=600)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "abi 4 rows 290"
    Z, Y, X, H, W, C = 8, 17, 17, 16, 16, 4
    occ = np.zeros((1, Z, Y, X), np.int64)
    occ[0, 5] = 1 + np.arange(Y * X).reshape(Y, X)
    feats = ((np.arange(H * W, dtype=np.float32).reshape(H, W, 1)) + np.float32(0.25) * np.arange(C, dtype=np.float32)).astype(np.float32)
    count = np.zeros(290, np.int32)
    out = np.zeros((290, C), np.float32)
    for _ in range(2):
        oracle_mod.project_features(feats[None, None], occ, np.eye(4, dtype=np.float32).reshape(-1), np.array([[8, 8, 8, 8]], np.float32),
                                    np.array([W, H, 0.01, 10.0, 0.5], np.float32), np.array([-8, -8, 0], np.float32), 1.0, count, out)
    got = {}
    for ln in lines[1:]:
        t = ln.split()
        got[int(t[1])] = (int(t[3]), [float(v) for v in t[5:9]])
    ids = np.nonzero(count)[0]
    assert sorted(got) == ids.tolist() and len(ids) > 50
    for i in ids:
        assert got[i][0] == count[i]
        assert np.array_equal(np.array(got[i][1], np.float32), out[i]), i      # %.9g round-trips float32


def test_entry_point_writes_checkpoints_and_the_feature_ply(tmp_path):
    # AGG:318-352 (a consolidated checkpoint every 20 views) and AGG:425-440 (the final ASCII .ply with the first three
    # channels as uchar rgb), in both modes; the parity mode's checkpoint equals a 20-view run's final result
    import aggregate_voxel_features_onthefly as agg
    s, ply, lseg, cam_json = _write_scene_files(tmp_path, n_views=22, C=8)
    common = ["--lseg_dir", str(lseg), "--cam_params", str(cam_json), "--voxel_ply", str(ply), "--views_per_call", "3"]
    for mode in ("parity", "fast"):
        out_dir = tmp_path / mode
        agg.main(["--mode", mode, "--checkpoint_dir", str(out_dir)] + common)
        ck = torch.load(out_dir / "checkpoint_features_20.pt")
        assert set(ck) == {"xyz", "avg_feats", "hit_count", "voxel_coords"}
        assert ck["xyz"].dtype == torch.float64 and ck["avg_feats"].dtype == torch.float16 and ck["hit_count"].dtype == torch.int32
        fin = torch.load(out_dir / f"ALL_nonzero_voxel_features_22_vox{s.n_vox}.pt")
        assert fin["xyz"].shape[0] >= ck["xyz"].shape[0] > 300
        lines = (out_dir / f"ALL_nonzero_voxels_with_features_22_vox{s.n_vox}.ply").read_text().splitlines()
        n = fin["xyz"].shape[0]
        assert lines[:3] == ["ply", "format ascii 1.0", f"element vertex {n}"] and lines[9] == "end_header" and len(lines) == 10 + n
        x, y, z, r, g, b = lines[10].split()
        assert np.allclose([float(x), float(y), float(z)], fin["xyz"][0].numpy(), rtol=1e-6)
        exp_rgb = (np.clip(fin["avg_feats"][0, :3].numpy(), 0, 1) * 255).astype(np.uint8)
        assert [int(r), int(g), int(b)] == exp_rgb.tolist()
    out20 = tmp_path / "parity20"
    agg.main(["--mode", "parity", "--checkpoint_dir", str(out20), "--max_images", "20"] + common)
    ck = torch.load(tmp_path / "parity" / "checkpoint_features_20.pt")
    fin20 = torch.load(out20 / f"ALL_nonzero_voxel_features_20_vox{s.n_vox}.pt")
    assert torch.equal(ck["voxel_coords"], fin20["voxel_coords"]) and torch.equal(ck["avg_feats"], fin20["avg_feats"])


def test_parity_aggregator_is_bit_exact_with_large_voxels_under_production_settings(oracle_mod, heavy_threshold):
    # Cameras a few voxels from a wall: single voxels collect far more than the production heavy threshold (256 + 64
    # pixels for a one-view call).  The parity mode promises the reference's bits (per-view sums rounded to float16,
    # DPF:252), so it asks the projector for serial sums (VP_FLAG_SERIAL_SUMS) -- no threshold override here.
    from aggregate_voxel_features_onthefly import VoxelFeatureAggregator
    from synthetic_scene import make_features_np, make_scene
    heavy_threshold(None)
    s = make_scene(2000, 4, 96, 64, seed=33, room=(5.0, 4.0, 2.4))
    c2w = s.c2w.copy()
    for v in range(4):                                            # move every camera to 3 voxels from the wall it faces
        wall = s.points[np.argmax(s.points @ c2w[v, :3, 2])]
        c2w[v, :3, 3] = wall - c2w[v, :3, 2] * np.float32(3.0 * s.voxel_size)
    C = 16
    feats = make_features_np(4, 64, 96, C, seed=33)
    n_rows = s.n_vox + 1
    per_view, biggest = [], 0
    for v in range(4):
        cnt = np.zeros(n_rows, np.int32)
        sums = np.zeros((n_rows, C), np.float32)
        oracle_mod.project_features(feats[None, v:v + 1], s.occ[None].astype(np.int64), c2w[v].reshape(-1), s.intr[None], s.opts(),
                                    s.grid_origin, s.voxel_size, cnt, sums)
        biggest = max(biggest, int(cnt.max()))
        per_view.append(oracle_mod.dpf_select_outputs(s.occ, cnt, sums))
    assert biggest > 256 + 64, biggest                            # the heavy path WOULD have been taken
    exp = oracle_mod.aggregate_views(per_view, s.grid_origin.astype(np.float64), s.voxel_size)
    agg = VoxelFeatureAggregator(torch.from_numpy(s.occ), s.grid_origin.astype(np.float64), s.voxel_size, C, "parity", DEV)
    agg.add_views(torch.from_numpy(feats).to(DEV), torch.from_numpy(c2w), torch.from_numpy(s.intr))
    r = agg.result()
    assert np.array_equal(r["voxel_coords"].numpy(), exp["voxel_coords"]) and np.array_equal(r["hit_count"].numpy(), exp["hit_count"])
    assert r["avg_feats"].numpy().tobytes() == exp["avg_feats"].tobytes()


def test_reference_three_script_chain_on_files(tmp_path, oracle_mod):
    # What aggregate_voxel_features_onthefly.py:118-127,248-294 runs as sub-processes per view -- build_sparse_occupancy.py ->
    # prepare_tensor_data.py (--max_images 1 --downsample_factor 0.5 --image_size H W) -> debug_project_features.py -- with
    # this package's scripts of the same names, through their command-line mains and the files in between.  The final
    # proj_output.pt must be what the oracle computes from the same inputs (occupancy builder, OpenCV-rule resize,
    # projector, fp16 hit rows), bit for bit.
    import build_sparse_occupancy as bso
    import debug_project_features as dpf
    import prepare_tensor_data as ptd
    from oracle import resize_oracle as ro
    s, ply, lseg, cam_json = _write_scene_files(tmp_path, n_views=2, C=16)
    occ_pt, td_pt, out_pt = tmp_path / "ALL_occupancy.pt", tmp_path / "tensor_data.pt", tmp_path / "proj_output.pt"
    vs, origin, _, _ = bso.extract_voxel_params(str(ply))
    bso.main(["--voxel_ply", str(ply), "--voxel_size", repr(vs), "--grid_origin", *[repr(v) for v in origin], "--out_tensor", str(occ_pt)])
    ptd.main(["--lseg_dir", str(lseg), "--scaled_camera_params", str(cam_json), "--occupancy", str(occ_pt), "--voxel_size", repr(vs),
              "--grid_origin", *[repr(v) for v in origin], "--max_images", "1", "--output", str(td_pt),
              "--image_size", "32", "48", "--downsample_factor", "0.5"])
    dpf.main(["--tensor_data", str(td_pt), "--output", str(out_pt)])
    got = torch.load(out_pt)
    assert set(got) == {"projected_feats", "projected_indices"}
    # the oracle's version of the same chain
    occ = oracle_mod.build_occupancy(bso.read_voxel_ply(str(ply)), origin, vs)
    assert np.array_equal(torch.load(occ_pt).numpy(), occ)
    first = sorted(os.listdir(lseg))[0]
    feats = ro.upsample_features(np.load(lseg / first), 32, 48)[None, None]                  # [1,1,H,W,C] float32
    td = torch.load(td_pt)
    assert td["encoded_2d_features"].numpy().tobytes() == feats.tobytes()
    intr0 = td["intrinsicParams"][:, 0, :].numpy()
    c2w0 = td["viewMatrixInv"][0, 0].numpy()
    n_rows = int(occ.max()) + 1
    cnt = np.zeros(n_rows, np.int32)
    sums = np.zeros((n_rows, 16), np.float32)
    opts = np.array([48, 32, 0.01, 10.0, vs * 0.5], np.float32)                               # DPF:167-169
    oracle_mod.project_features(feats, occ[None].astype(np.int64), c2w0.reshape(-1), intr0, opts, np.array(origin, np.float32), vs,
                                cnt, sums)
    ef, ei = oracle_mod.dpf_select_outputs(occ, cnt, sums)
    assert len(ei) > 100
    assert np.array_equal(got["projected_indices"].numpy(), ei)
    assert got["projected_feats"].dtype == torch.float16 and got["projected_feats"].numpy().tobytes() == ef.tobytes()


def test_reference_colour_script_chain_on_files(tmp_path, oracle_mod):
    # aggregate_voxel_colors_onthefly.py:103-121 per view: prepare_tensor_data_color.py -> debug_project_colors.py, through
    # this package's command-line mains and the tensor_data.pt (with its `image` key, PTDC:144) in between
    import json

    from PIL import Image

    import debug_project_colors as dpc
    import prepare_tensor_data_color as ptdc
    from synthetic_scene import make_scene
    s = make_scene(1500, 1, 60, 40, seed=93, room=(5.0, 4.0, 2.4))
    rng = np.random.default_rng(93)
    lseg, images = tmp_path / "features", tmp_path / "images"
    lseg.mkdir(); images.mkdir()
    np.save(lseg / "IMG000.npy", rng.standard_normal((4, 10, 15)).astype(np.float16))
    img = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    Image.fromarray(img).save(images / "IMG000.png")
    c2w = s.c2w[0].astype(np.float64)
    R = c2w[:3, :3].T
    cams = {"images": {"0": {"name": "IMG000", "camera_id": 1, "R": R.tolist(), "tvec": (-R @ c2w[:3, 3]).tolist()}},
            "cameras": {"1": {"params": [float(x) for x in s.intr]}}}
    (tmp_path / "cams.json").write_text(json.dumps(cams))
    torch.save(torch.from_numpy(s.occ), tmp_path / "occ.pt")
    td, out = tmp_path / "tensor_data.pt", tmp_path / "proj_output.pt"
    ptdc.main(["--lseg_dir", str(lseg), "--scaled_camera_params", str(tmp_path / "cams.json"), "--occupancy", str(tmp_path / "occ.pt"),
               "--voxel_size", repr(s.voxel_size), "--grid_origin", *[repr(float(v)) for v in s.grid_origin], "--max_images", "1",
               "--output", str(td), "--images_dir", str(images)])
    dpc.main(["--tensor_data", str(td), "--output", str(out)])
    got = torch.load(out)
    assert set(got) == {"projected_colors", "projected_indices", "pixel_indices"}
    d = torch.load(td, weights_only=False)
    col, zyx, uv = oracle_mod.rgb_project(s.occ, d["viewMatrixInv"][0, 0].numpy(), d["intrinsicParams"][0, 0].numpy(),
                                          d["grid_origin"].numpy(), d["voxel_size"], d["image"])
    assert len(zyx) > 100 and np.array_equal(d["image"], img)
    assert np.array_equal(got["projected_indices"].numpy(), zyx) and np.array_equal(got["pixel_indices"].numpy(), uv)
    assert got["projected_colors"].numpy().tobytes() == col.tobytes()


def test_feature_feeder_delivers_every_map_in_order(tmp_path):
    # the entry point's read-ahead (worker threads, pinned staging, copy stream) must hand over exactly the files' bytes, in
    # file order, whatever the depth -- including more workers than files and maps of different shapes and dtypes
    from aggregate_voxel_features_onthefly import FeatureFeeder
    rng = np.random.default_rng(17)
    paths, arrays = [], []
    for i in range(7):
        shape = (6, 5 + (i % 3), 9) if i != 4 else (3, 4, 4)
        a = rng.standard_normal(shape).astype(np.float16 if i % 2 == 0 else np.float32)
        p = tmp_path / f"m{i:02d}.npy"
        np.save(p, a)
        paths.append(str(p)); arrays.append(a)
    # a Fortran-ordered file (numpy's general loader, not the piece-wise reader) and a map larger than several pieces
    f = np.asfortranarray(rng.standard_normal((4, 6, 5)).astype(np.float32))
    np.save(tmp_path / "m07.npy", f)
    paths.append(str(tmp_path / "m07.npy")); arrays.append(np.ascontiguousarray(f))
    be = rng.standard_normal((3, 5, 4)).astype(">f4")                    # foreign byte order: converted by the general loader
    np.save(tmp_path / "m07b.npy", be)
    paths.append(str(tmp_path / "m07b.npy")); arrays.append(be.astype("<f4"))
    big = rng.standard_normal((16, 96, 130)).astype(np.float16)
    np.save(tmp_path / "m08.npy", big)
    paths.append(str(tmp_path / "m08.npy")); arrays.append(big)
    for depth, chunk, threads in ((0, None, None), (1, None, 1), (3, 4096, 3), (16, 100000, None), (2, 1 << 20, 2)):
        feeder = FeatureFeeder(paths, DEV, depth, io_threads=threads)
        if chunk:
            feeder.CHUNK = chunk                 # small pieces: every file is cut into many, the last one short
        seen = []
        for i, p, t in feeder:
            assert p == paths[i] and t.is_cuda
            seen.append(t.cpu().numpy())
        assert len(seen) == len(paths)
        for a, b in zip(arrays, seen):
            assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def test_feature_feeder_closes_its_files_on_errors_and_early_exits(tmp_path):
    # a truncated map surfaces as an IOError that NAMES the file; whether the feeder ends by exhaustion, by the
    # consumer breaking out or by an error, no descriptor of a read-ahead file stays open
    from aggregate_voxel_features_onthefly import FeatureFeeder
    rng = np.random.default_rng(19)
    paths = []
    for i in range(8):
        p = tmp_path / f"f{i:02d}.npy"
        np.save(p, rng.standard_normal((8, 40, 50)).astype(np.float32))
        paths.append(str(p))
    short = tmp_path / "f03.npy"
    data = short.read_bytes()
    short.write_bytes(data[:len(data) // 2])                              # the header promises twice the bytes

    def open_fds():
        return {os.readlink(f"/proc/self/fd/{n}") for n in os.listdir("/proc/self/fd") if os.path.exists(f"/proc/self/fd/{n}")}

    before = {f for f in open_fds() if str(tmp_path) in f}
    feeder = FeatureFeeder(paths, DEV, depth=3, io_threads=3)
    feeder.CHUNK = 8192
    got = []
    with pytest.raises(IOError, match="f03.npy"):
        for i, p, t in feeder:
            got.append(i)
    assert got == [0, 1, 2]
    assert {f for f in open_fds() if str(tmp_path) in f} == before
    short.write_bytes(data)
    for i, p, t in FeatureFeeder(paths, DEV, depth=4, io_threads=2):      # the consumer walks away after two maps
        if i == 1:
            break
    import gc
    gc.collect()
    assert {f for f in open_fds() if str(tmp_path) in f} == before


@pytest.mark.parametrize("dims,n_vox", [((1, 1, 1), 1), ((1, 1, 300), 40), ((3, 1, 2), 6), ((2, 130, 5), 300), ((70, 9, 33), 2500),
                                         ((150, 170, 190), 3000), ((64, 64, 64), 64 ** 3), ((5, 4, 1030), 900)])
def test_rgb_voxel_list_along_the_curve_on_awkward_grids(oracle_mod, dims, n_vox):
    """k_color_cells walks the grid's 4x4x4 blocks along a Morton curve and k_project_colors sums the voxels in that order:
    grids of one cell, of one row, with axes that are no multiple of four or run out of bits early, a full grid (every block
    of the curve occupied), more curve positions than walking wavefronts.  Five views in two calls, with and without the
    sampled pixels: sums (view order), hit counts, first views and pixels equal the oracle's bit for bit; IDs that label no
    cell stay untouched."""
    import voxproj_host
    dev = torch.device(DEV)
    rng = np.random.default_rng(sum(dims) + n_vox)
    Z, Y, X = dims
    occ = np.zeros(dims, np.int32)
    n_rows = n_vox + 4                                             # three IDs beyond the labelled ones: rows without a cell
    occ.reshape(-1)[rng.choice(occ.size, n_vox, replace=False)] = rng.permutation(n_vox) + 1
    vs = 0.05
    origin = np.array([-0.5 * X * vs, -0.5 * Y * vs, -0.5 * Z * vs], np.float32)
    V, iw, ih = 5, 96, 64
    c2w = np.zeros((V, 4, 4), np.float32)
    reach = 0.6 * vs * max(dims) + 0.5
    for v in range(V):
        a = 2.0 * np.pi * v / V + 0.3
        pos = np.array([reach * np.cos(a), reach * np.sin(a), 0.3 * reach * np.sin(2 * a)])
        f = -pos / np.linalg.norm(pos)
        right = np.cross(f, np.array([0.0, 0.0, 1.0])); right /= np.linalg.norm(right)
        c2w[v, :3, 0], c2w[v, :3, 1], c2w[v, :3, 2], c2w[v, :3, 3] = right, np.cross(f, right), f, pos
        c2w[v, 3, 3] = 1.0
    intr = np.tile(np.array([70.0, 72.0, iw / 2 + 0.5, ih / 2], np.float32), (V, 1))
    imgs = rng.integers(0, 256, (V, ih, iw, 3), dtype=np.uint8)
    ref_sum = np.zeros((n_rows, 3), np.float32)
    ref_hits = np.zeros(n_rows, np.int32)
    ref_first = np.full(n_rows, 2 ** 30, np.int32)
    ref_uv = np.full((V, n_rows, 2), -1, np.int32)
    for v in range(V):
        colors, zyx, uv = oracle_mod.rgb_project(occ, c2w[v], intr[v], origin, vs, imgs[v])
        ids = occ[zyx[:, 0], zyx[:, 1], zyx[:, 2]]
        ref_sum[ids] += colors
        ref_hits[ids] += 1
        ref_first[ids] = np.minimum(ref_first[ids], 7 + v)
        ref_uv[v, ids] = uv
    assert ref_hits.sum() > 0
    occ_t = torch.from_numpy(occ).to(dev)
    for want_uv in (False, True):
        csum = torch.zeros(n_rows, 3, device=dev)
        hits = torch.zeros(n_rows, dtype=torch.int32, device=dev)
        first = torch.full((n_rows,), 2 ** 30, dtype=torch.int32, device=dev)
        uv_t = torch.full((V, n_rows, 2), 77, dtype=torch.int32, device=dev) if want_uv else None
        for lo, hi in ((0, 3), (3, 5)):
            voxproj_host.project_colors_raw(occ_t, torch.from_numpy(c2w[lo:hi]).to(dev), torch.from_numpy(intr[lo:hi]).to(dev),
                                            [float(v) for v in origin], vs, torch.from_numpy(imgs[lo:hi]).to(dev), csum, hits,
                                            first_view=first, view_base=7 + lo, pixel_uv=uv_t[lo:hi] if want_uv else None)
        assert csum.cpu().numpy().tobytes() == ref_sum.tobytes()
        assert np.array_equal(hits.cpu().numpy(), ref_hits) and np.array_equal(first.cpu().numpy(), ref_first)
        if want_uv:
            assert np.array_equal(uv_t.cpu().numpy(), ref_uv)
