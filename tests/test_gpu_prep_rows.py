"""GPU parity for the data-format rows either side of the projector, through the C-ABI:

  vp_upsample_features   vs oracle/resize_oracle.py (OpenCV's INTER_LINEAR rule in numpy float32): bit for bit, and vs a
                         float64 bilinear evaluation within one fp16 ulp
  vp_voxel_coords + vp_scatter_occupancy   vs oracle.build_occupancy (build_sparse_occupancy.py:30-53): bit for bit
  vp_aggregate_view_f16  vs oracle.aggregate_views (aggregate_voxel_features_onthefly.py:307-313): bit for bit
"""
import numpy as np
import pytest
import torch

from synthetic_scene import make_features_np, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape,size", [((5, 6, 9), (12, 18)), ((512, 36, 54), (58, 87)), ((16, 45, 60), (73, 97)),
                                         ((7, 9, 13), (9, 13)), ((24, 10, 12), (5, 7)), ((3, 1, 1), (4, 5)), ((20, 33, 2), (40, 7))])
def test_upsampler_matches_the_resize_oracle_bit_for_bit(shape, size):
    import voxproj_host
    from oracle import resize_oracle as ro
    rng = np.random.default_rng(sum(shape) * 7 + sum(size))
    H, W = size
    for dt in (np.float16, np.float32):
        arr = rng.standard_normal(shape).astype(dt)
        arr.reshape(-1)[:: 11] *= 40                                   # some large values (fp16 rounding at coarse ulps)
        exp = ro.upsample_features(arr, H, W)
        got = voxproj_host.upsample_features(torch.from_numpy(arr).to(DEV), H, W)
        assert got.dtype == torch.float32 and tuple(got.shape) == (H, W, shape[0]) and got.is_contiguous()
        assert got.cpu().numpy().tobytes() == exp.tobytes()
        if dt == np.float16:
            kept = voxproj_host.upsample_features(torch.from_numpy(arr).to(DEV), H, W, keep_dtype=True)
            assert kept.dtype == torch.float16 and kept.cpu().numpy().tobytes() == ro.upsample_features(arr, H, W, keep_dtype=True).tobytes()
            # float64 bilinear with half-pixel centres: the fp16 results agree within one ulp of the value
            if H >= shape[1] and W >= shape[2]:
                ref = np.transpose(ro.bilinear_f64(arr, H, W), (1, 2, 0))
                r16 = ref.astype(np.float16)
                ulp = np.spacing(np.abs(r16)).astype(np.float64)
                scale = np.abs(arr.astype(np.float64)).max()
                assert (np.abs(kept.cpu().numpy().astype(np.float64) - r16.astype(np.float64)) <= ulp + 4e-6 * scale).all()


def test_upsampler_at_the_pipeline_shape_and_through_the_host_mirror():
    # LSeg map of a 1752x1168 DSLR image (shorter side 360 -> fp16 [512,360,540], script/extract_lseg_features.py:64-97)
    # up-sampled to the aggregator's working resolution 876x584 (AGG:209, PTD:119-127): sampled rows against the oracle
    import prepare_tensor_data as ptd
    from oracle import resize_oracle as ro
    rng = np.random.default_rng(12)
    arr = rng.standard_normal((512, 360, 540)).astype(np.float16)
    H, W = 584, 876
    got = ptd.upsample_features(arr, (H, W), device=DEV)
    assert got.dtype == torch.float32 and tuple(got.shape) == (H, W, 512)
    rows = [0, 1, 291, 292, 582, 583]
    sub = ro.resize_linear_f32(arr.astype(np.float32), H, W)[:, rows, :].astype(np.float16).astype(np.float32)
    assert np.array_equal(got[rows].cpu().numpy(), np.transpose(sub, (1, 2, 0)))
    half = ptd.upsample_features(arr, (H, W), device=DEV, keep_dtype=True)
    assert half.dtype == torch.float16 and torch.equal(half.float(), got)


def test_occupancy_builder_kernels_match_the_oracle(oracle_mod):
    import build_sparse_occupancy as bso
    rng = np.random.default_rng(3)
    cases = []
    pts = (rng.uniform(-1, 1, size=(500, 3)) * np.array([2.0, 1.5, 1.0])).astype(np.float32)
    pts[10] = pts[3]                                                     # duplicates: the last vertex wins (Q12)
    pts[499] = pts[3]
    cases += [(pts, [-2.0, -1.5, -1.0], 0.25), (pts, [0.3, -0.2, 0.1], 0.25)]     # second origin: negative coords, shifted (Q11)
    # points exactly half-way between cells: np.round is half-to-even (Q10)
    g = np.stack(np.meshgrid(np.arange(6), np.arange(5), np.arange(4), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    cases.append((g * 0.5 + 0.25, [0.0, 0.0, 0.0], 0.5))
    s = make_scene(20000, 2, 64, 48, seed=4)
    cases.append((s.points, [float(v) for v in s.grid_origin], s.voxel_size))
    for p, origin, vs in cases:
        exp = oracle_mod.build_occupancy(p, origin, vs)
        got = bso.build_occupancy(p, origin, vs, device=DEV)
        assert got.dtype == torch.int32 and got.is_cuda and tuple(got.shape) == exp.shape
        assert np.array_equal(got.cpu().numpy(), exp)
    assert np.array_equal(bso.build_occupancy(s.points, s.grid_origin, s.voxel_size, device=DEV).cpu().numpy(), s.occ)


def test_per_view_fp16_accumulate_matches_the_reference_dict_loop(oracle_mod):
    # vp_aggregate_view_f16 alone: random per-view sums (some beyond the fp16 range -> inf, flagged) against the
    # dict loop of aggregate_voxel_features_onthefly.py:307-313 restated in oracle.aggregate_views
    import voxproj_host
    rng = np.random.default_rng(8)
    n_rows, C, V = 300, 40, 6
    occ = np.zeros((5, 10, 12), np.int32)
    idx = rng.choice(occ.size, n_rows - 1, replace=False)
    occ.reshape(-1)[idx] = np.arange(1, n_rows)
    dev = torch.device(DEV)
    view_sum = torch.zeros(n_rows, C, device=dev)
    view_cnt = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    run16 = torch.zeros(n_rows, C, dtype=torch.float16, device=dev)
    views = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    first = torch.full((n_rows,), 2 ** 30, dtype=torch.int32, device=dev)
    flags = torch.zeros(V, dtype=torch.int32, device=dev)
    per_view = []
    for v in range(V):
        cnt = (rng.random(n_rows) < 0.4).astype(np.int32) * rng.integers(1, 50, n_rows).astype(np.int32)
        cnt[0] = 0
        sums = (rng.standard_normal((n_rows, C)) * 30).astype(np.float32) * (cnt > 0)[:, None]
        if v == 3:
            sums[np.nonzero(cnt)[0][0], 5] = 1e6                         # overflows float16 -> inf -> reported (AGG:303-304)
        per_view.append(oracle_mod.dpf_select_outputs(occ, cnt, sums))
        view_sum.copy_(torch.from_numpy(sums))
        view_cnt.copy_(torch.from_numpy(cnt))
        voxproj_host.aggregate_view_f16(view_sum, view_cnt, run16, views, first, v, flags, v)
        assert int(view_cnt.abs().sum()) == 0 and float(view_sum.abs().sum()) == 0.0      # scratch is clean again
    assert flags.cpu().tolist() == [0, 0, 0, 1, 0, 0]
    exp = oracle_mod.aggregate_views(per_view, [0.0, 0.0, 0.0], 0.1)
    ids = torch.nonzero(views > 0).reshape(-1)
    order = torch.argsort(first[ids].long() * n_rows + ids)
    ids = ids[order]
    zyx = np.stack(np.unravel_index([int(np.nonzero(occ.reshape(-1) == i)[0][0]) for i in ids.tolist()], occ.shape), 1)
    assert np.array_equal(zyx.astype(np.int32), exp["voxel_coords"])
    assert np.array_equal(views[ids].cpu().numpy(), exp["hit_count"])
    with np.errstate(invalid="ignore"):
        avg = (run16[ids].float() / views[ids].float()[:, None]).to(torch.float16).cpu().numpy()
    assert avg.tobytes() == exp["avg_feats"].tobytes()


def test_parity_aggregator_reports_nonfinite_views(capsys):
    # the aggregator prints the reference's per-view error line (AGG:303-304) for a view whose fp16 rows overflow
    from aggregate_voxel_features_onthefly import VoxelFeatureAggregator
    s = make_scene(2000, 3, 48, 32, seed=21, room=(5.0, 4.0, 2.4))
    C = 8
    feats = make_features_np(3, 32, 48, C, seed=21)
    feats[1] *= 1e5                                                     # per-view pixel sums far beyond 65504
    agg = VoxelFeatureAggregator(torch.from_numpy(s.occ), s.grid_origin.astype(np.float64), s.voxel_size, C, "parity", DEV)
    agg.add_views(torch.from_numpy(feats).to(DEV), torch.from_numpy(s.c2w), torch.from_numpy(s.intr))
    agg.result()
    out = capsys.readouterr().out
    assert "NaN or Inf detected in projected features for view 1" in out and "view 0" not in out and "view 2" not in out


def test_new_entry_points_refuse_bad_arguments():
    # loud failures instead of undefined behaviour: duplicate voxel IDs on the colour path, points that cannot be binned,
    # a float16 destination for a float32 source, a workspace that is too small
    import ctypes

    import voxproj_host
    dev = torch.device(DEV)
    occ = torch.zeros(3, 4, 5, dtype=torch.int32, device=dev)
    occ[0, 0, 0] = 1
    occ[1, 2, 3] = 1                                                   # the same ID in two cells
    c2w = torch.eye(4, device=dev)[None].contiguous()
    intr = torch.tensor([[10.0, 10.0, 4.0, 4.0]], device=dev)
    img = torch.zeros(1, 8, 8, 3, dtype=torch.uint8, device=dev)
    with pytest.raises(voxproj_host.VoxprojError, match="more than one cell"):
        voxproj_host.project_colors_raw(occ, c2w, intr, [0, 0, 0], 0.1, img, torch.zeros(2, 3, device=dev),
                                        torch.zeros(2, dtype=torch.int32, device=dev))
    occ[1, 2, 3] = 7
    with pytest.raises(voxproj_host.VoxprojError, match="outside"):
        voxproj_host.project_colors_raw(occ, c2w, intr, [0, 0, 0], 0.1, img, torch.zeros(2, 3, device=dev),
                                        torch.zeros(2, dtype=torch.int32, device=dev))
    pts = torch.tensor([[0.0, 0.0, 0.0], [float("nan"), 1.0, 2.0]], device=dev)
    with pytest.raises(voxproj_host.VoxprojError, match="not finite"):
        voxproj_host.build_occupancy_device(pts, [0, 0, 0], 0.1)
    L = voxproj_host.lib()
    src = torch.zeros(4, 3, 3, device=dev)
    dst = torch.zeros(6, 6, 4, dtype=torch.float16, device=dev)
    ws = torch.zeros(4096, dtype=torch.uint8, device=dev)
    p = ctypes.c_void_p
    assert L.vp_upsample_features(p(src.data_ptr()), 0, 4, 3, 3, p(dst.data_ptr()), 1, 6, 6, p(ws.data_ptr()), 4096, None) == -1
    assert b"float16 destination" in L.vp_last_error()
    assert L.vp_upsample_features(p(src.data_ptr()), 0, 4, 3, 3, p(dst.data_ptr()), 0, 6, 6, p(ws.data_ptr()), 16, None) == -2
    assert b"workspace" in L.vp_last_error()
