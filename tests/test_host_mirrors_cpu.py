"""Host-side mirrors of the reference's packing script (prepare_tensor_data.py) on the CPU: camera conventions,
JSON schema, feature up-sampling contract, and the tensor_data.pt schema."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rot(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_camera_for_inverts_world_to_camera_and_scales_intrinsics():
    import prepare_tensor_data as ptd
    rng = np.random.default_rng(5)
    R, t = _rot(rng), rng.standard_normal(3)
    entry = {"camera_id": 3, "R": R.tolist(), "tvec": t.tolist()}
    cams = {"3": {"params": [600.0, 610.0, 320.0, 240.0]}, "4": {"params": [500.0, 100.0, 80.0]}}
    intr, c2w = ptd.camera_for(entry, cams, downsample_factor=0.5)
    assert intr.dtype == torch.float32 and intr.tolist() == [300.0, 305.0, 160.0, 120.0]       # PTD:132-143
    w2c = np.eye(4)
    w2c[:3, :3], w2c[:3, 3] = R, t
    assert np.abs(w2c @ c2w.numpy().astype(np.float64) - np.eye(4)).max() < 1e-6              # PTD:165-172
    # float32 arithmetic like the reference: R^T and -R^T t formed in float32
    R32, t32 = R.astype(np.float32), t.astype(np.float32)
    assert c2w.numpy()[:3, 3].tobytes() == (-R32.T @ t32).tobytes()
    intr3, _ = ptd.camera_for(dict(entry, camera_id=4), cams)                                  # simple-pinhole: fy = fx
    assert intr3.tolist() == [500.0, 500.0, 100.0, 80.0]


def test_load_camera_params_accepts_dict_and_list(tmp_path):
    import prepare_tensor_data as ptd
    a = {"images": {"1": {"name": "IMG_1", "camera_id": 1, "R": np.eye(3).tolist(), "tvec": [0, 0, 0]}},
         "cameras": {"1": {"params": [1, 1, 0, 0]}}}
    b = dict(a, images=list(a["images"].values()))
    for i, d in enumerate((a, b)):
        p = tmp_path / f"c{i}.json"
        p.write_text(json.dumps(d))
        by_name, cams = ptd.load_camera_params(str(p))
        assert list(by_name) == ["IMG_1"] and by_name["IMG_1"]["camera_id"] == 1 and "1" in cams


def test_upsample_features_contract():
    import prepare_tensor_data as ptd
    C, h, w = 5, 6, 9
    rng = np.random.default_rng(6)
    arr = rng.standard_normal((C, h, w)).astype(np.float16)
    same = ptd.upsample_features(arr)                                       # no size: only the layout changes
    assert same.dtype == torch.float32 and same.shape == (h, w, C)
    assert np.array_equal(same.numpy(), arr.astype(np.float32).transpose(1, 2, 0))
    kept = ptd.upsample_features(arr, keep_dtype=True)
    assert kept.dtype == torch.float16 and np.array_equal(kept.numpy(), arr.transpose(1, 2, 0))
    # a change of size is the HIP up-sampler's job (tests/test_gpu_prep_rows.py): no CPU fallback
    with pytest.raises(RuntimeError, match="GPU"):
        ptd.upsample_features(arr, size=(12, 18))


def test_prepare_tensor_data_main_writes_the_reference_schema(tmp_path):
    import prepare_tensor_data as ptd
    rng = np.random.default_rng(7)
    lseg = tmp_path / "feats"
    lseg.mkdir()
    names = ["DSC_0002", "DSC_0001", "DSC_0003"]
    for n in names:
        np.save(lseg / f"{n}.npy", rng.standard_normal((4, 5, 7)).astype(np.float16))
    np.save(lseg / "no_camera.npy", np.zeros((4, 5, 7), np.float16))
    cams = {"images": {str(i): {"name": n, "camera_id": 1, "R": _rot(rng).tolist(), "tvec": rng.standard_normal(3).tolist()}
                       for i, n in enumerate(names)},
            "cameras": {"1": {"params": [700.0, 700.0, 350.0, 250.0]}}}
    (tmp_path / "cam.json").write_text(json.dumps(cams))
    occ = torch.zeros(3, 4, 5, dtype=torch.int32)
    occ[1, 2, 3] = 1
    torch.save(occ, tmp_path / "occ.pt")
    out = tmp_path / "tensor_data.pt"
    ptd.main(["--lseg_dir", str(lseg), "--scaled_camera_params", str(tmp_path / "cam.json"), "--occupancy", str(tmp_path / "occ.pt"),
              "--voxel_size", "0.05", "--grid_origin", "1", "2", "3", "--max_images", "10", "--output", str(out),
              "--image_size", "5", "7", "--downsample_factor", "0.5", "--device", "cpu"])     # same size: no resize, no GPU
    d = torch.load(out)
    assert set(d) == {"encoded_2d_features", "occupancy_3D", "intrinsicParams", "viewMatrixInv", "grid_origin", "voxel_size"}
    assert d["encoded_2d_features"].shape == (1, 3, 5, 7, 4) and d["encoded_2d_features"].dtype == torch.float32
    assert d["viewMatrixInv"].shape == (1, 3, 4, 4) and d["viewMatrixInv"].dtype == torch.float32
    # PTD:143 and :162 both append when a downsample factor is given: two intrinsics rows per view,
    # the scaled one first -- index 0 is what debug_project_features.py:146 reads
    assert d["intrinsicParams"].shape == (1, 6, 4)
    assert d["intrinsicParams"][0, 0].tolist() == [350.0, 350.0, 175.0, 125.0]
    assert d["intrinsicParams"][0, 1].tolist() == [700.0, 700.0, 350.0, 250.0]
    assert torch.equal(d["occupancy_3D"], occ) and d["voxel_size"] == 0.05
    assert d["grid_origin"].dtype == torch.float32 and d["grid_origin"].tolist() == [1.0, 2.0, 3.0]
    # without a factor the reference still appends twice (PTD:152 and :162): two identical rows per view
    ptd.main(["--lseg_dir", str(lseg), "--scaled_camera_params", str(tmp_path / "cam.json"), "--occupancy", str(tmp_path / "occ.pt"),
              "--voxel_size", "0.05", "--grid_origin", "1", "2", "3", "--max_images", "2", "--output", str(tmp_path / "t2.pt"),
              "--device", "cpu"])
    d2 = torch.load(tmp_path / "t2.pt")
    assert d2["encoded_2d_features"].shape == (1, 2, 5, 7, 4) and d2["intrinsicParams"].shape == (1, 4, 4)
    assert d2["intrinsicParams"][0, 0].tolist() == d2["intrinsicParams"][0, 1].tolist() == [700.0, 700.0, 350.0, 250.0]
    # views follow the sorted file names (PTD:101-104), the file without a camera entry is skipped
    first = np.load(lseg / "DSC_0001.npy")
    assert torch.equal(d["encoded_2d_features"][0, 0], ptd.upsample_features(first))


def test_prepare_tensor_data_color_writes_the_reference_schema_with_image_key(tmp_path):
    # prepare_tensor_data_color.py:60-160: same dict as prepare_tensor_data.py plus `image` (uint8 [H,W,3], PTDC:144);
    # the feature map is up-sampled to the IMAGE size with torch's bilinear interpolate (PTDC:101-105); intrinsics are
    # the unscaled ones, one row per view; a view without an image file keeps its feature map's size
    from PIL import Image

    import prepare_tensor_data_color as ptdc
    rng = np.random.default_rng(9)
    lseg, images = tmp_path / "feats", tmp_path / "images"
    lseg.mkdir(); images.mkdir()
    arr = rng.standard_normal((4, 5, 7)).astype(np.float16)
    np.save(lseg / "DSC_0001.npy", arr)
    img = rng.integers(0, 256, (10, 14, 3), dtype=np.uint8)
    Image.fromarray(img).save(images / "DSC_0001.png")
    cams = {"images": {"0": {"name": "DSC_0001", "camera_id": 1, "R": _rot(rng).tolist(), "tvec": rng.standard_normal(3).tolist()}},
            "cameras": {"1": {"params": [700.0, 710.0, 350.0, 250.0]}}}
    (tmp_path / "cam.json").write_text(json.dumps(cams))
    occ = torch.zeros(3, 4, 5, dtype=torch.int32)
    occ[1, 2, 3] = 1
    torch.save(occ, tmp_path / "occ.pt")
    out = tmp_path / "tensor_data.pt"
    common = ["--lseg_dir", str(lseg), "--scaled_camera_params", str(tmp_path / "cam.json"), "--occupancy", str(tmp_path / "occ.pt"),
              "--voxel_size", "0.05", "--grid_origin", "1", "2", "3", "--max_images", "1", "--output", str(out)]
    ptdc.main(common + ["--images_dir", str(images)])
    d = torch.load(out, weights_only=False)
    assert set(d) == {"encoded_2d_features", "occupancy_3D", "intrinsicParams", "viewMatrixInv", "grid_origin", "voxel_size", "image"}
    assert isinstance(d["image"], np.ndarray) and d["image"].dtype == np.uint8 and np.array_equal(d["image"], img)
    assert d["encoded_2d_features"].shape == (1, 1, 10, 14, 4) and d["encoded_2d_features"].dtype == torch.float32
    exp = torch.nn.functional.interpolate(torch.from_numpy(arr)[None].float(), size=(10, 14), mode="bilinear", align_corners=False)[0]
    assert torch.equal(d["encoded_2d_features"][0, 0], exp.permute(1, 2, 0))
    assert d["intrinsicParams"].shape == (1, 1, 4) and d["intrinsicParams"][0, 0].tolist() == [700.0, 710.0, 350.0, 250.0]
    assert d["viewMatrixInv"].shape == (1, 1, 4, 4) and d["voxel_size"] == 0.05
    ptdc.main(common + ["--images_dir", str(tmp_path / "nowhere")])
    d2 = torch.load(out, weights_only=False)
    assert d2["encoded_2d_features"].shape == (1, 1, 5, 7, 4) and d2["image"].shape == (5, 7, 3) and not d2["image"].any()


def test_npy_layout_reads_the_header_only(tmp_path):
    # the feeder's piece-wise reader needs (data offset, shape, dtype) of plain C-ordered arrays and must hand anything else
    # (Fortran order, object arrays) to numpy's loader
    import aggregate_voxel_features_onthefly as agg
    rng = np.random.default_rng(3)
    a = rng.standard_normal((5, 7, 9)).astype(np.float16)
    np.save(tmp_path / "a.npy", a)
    off, shape, dtype = agg._npy_layout(str(tmp_path / "a.npy"))
    assert shape == (5, 7, 9) and dtype == np.float16
    raw = open(tmp_path / "a.npy", "rb").read()
    assert raw[off:] == a.tobytes() and len(raw) == off + a.nbytes
    np.save(tmp_path / "f.npy", np.asfortranarray(a))
    assert agg._npy_layout(str(tmp_path / "f.npy")) is None
    np.save(tmp_path / "o.npy", np.array([{"k": 1}], dtype=object), allow_pickle=True)
    assert agg._npy_layout(str(tmp_path / "o.npy")) is None
    np.save(tmp_path / "be.npy", a.astype(">f4"))                       # foreign byte order: numpy's loader converts, the piece reader must not
    assert agg._npy_layout(str(tmp_path / "be.npy")) is None
    np.save(tmp_path / "s.npy", np.zeros(3, dtype=[("x", "f4"), ("y", "i2")]))
    assert agg._npy_layout(str(tmp_path / "s.npy")) is None
    assert agg._granted_cpus() >= 1


def test_bench_colour_loop_restatement_equals_the_oracle(oracle_mod):
    """bench.py's cpu_torch_loop of the R4 leg -- the reference's colour loop (debug_project_colors.py:58-73) as a vectorised
    torch-CPU expression -- against oracle.rgb_project (pinned by the reference's own run): colours, voxels and pixels equal."""
    import importlib.util
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_module_colours", os.path.join(ROOT, "bench.py"))
        bm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bm)
    finally:
        sys.argv = argv
    from synthetic_scene import make_scene
    s = make_scene(3000, 3, 72, 48, seed=9, room=(5.0, 4.0, 2.4))
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (48, 72, 3), dtype=np.uint8)
    for v in range(3):
        colors, zyx, uv = oracle_mod.rgb_project(s.occ, s.c2w[v], s.intr, s.grid_origin, s.voxel_size, img)
        c2, z2, uv2 = bm.colour_projection_torch_cpu(torch.from_numpy(s.occ), torch.from_numpy(s.c2w[v]), torch.from_numpy(s.intr),
                                                     torch.from_numpy(s.grid_origin), s.voxel_size, torch.from_numpy(img))
        assert colors.shape[0] > 500
        assert np.array_equal(z2.numpy(), zyx) and np.array_equal(uv2.numpy(), uv)
        assert c2.numpy().tobytes() == colors.tobytes()


def test_bench_distinct_image_lines_against_a_brute_force_count(oracle_mod):
    """bench.py's algorithmic bytes of the R4 leg: distinct 64-byte image lines that hold a sampled pixel, per view, against a
    set built from the oracle's pixels."""
    import importlib.util
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_module_lines", os.path.join(ROOT, "bench.py"))
        bm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bm)
    finally:
        sys.argv = argv
    from synthetic_scene import make_scene
    W, H = 64, 48                                                      # H*W*3 is a multiple of 64, like config 5's images
    s = make_scene(3000, 3, W, H, seed=9, room=(5.0, 4.0, 2.4))
    img = np.zeros((H, W, 3), np.uint8)
    want = 0
    for v in range(3):
        uv = oracle_mod.rgb_project(s.occ, s.c2w[v], s.intr, s.grid_origin, s.voxel_size, img)[2]
        lines = set()
        for u, w in uv:
            off = (int(w) * W + int(u)) * 3
            lines.update((off // 64, (off + 2) // 64))
        want += len(lines)
    assert want > 100
    assert bm.distinct_image_lines(s, [0, 1, 2], H, W, torch.device("cpu")) == want
