"""The gather of ONE-VIEW calls (k_gather_one, csrc/vp_gather.h): what the drop-in module issues once per image
(debug_project_features.py:201-208) and the parity aggregator once per view.  A fixed grid of wavefronts is dealt the
size-ordered work list, lane j of a wavefront prepares its j-th voxel (ID, pixel count, pixel box), the next voxel's ID tile
and output row are fetched under the current voxel's rows.  Every case against the oracle bit for bit, and against the general
gather kernel (VP_OPT_ONE_VIEW_GATHER = 0, round 3's path for these calls)."""
import numpy as np
import pytest
import torch

from synthetic_scene import make_features_np, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tensors(s, feats, v, dev, occ=None):
    occ = s.occ if occ is None else occ
    return dict(feats=torch.from_numpy(np.ascontiguousarray(feats[:, v:v + 1])).to(dev), occ=torch.from_numpy(occ[None].astype(np.int64)).to(dev),
                vmi=torch.from_numpy(s.c2w[v]).reshape(-1).contiguous().to(dev), intr=torch.from_numpy(s.intr[None]).to(dev),
                opts=[float(x) for x in s.opts()], origin=[float(x) for x in s.grid_origin])


def _call(t, s, ws, count, out, **kw):
    import voxproj_host
    return voxproj_host.project_features_raw(t["feats"], t["occ"], t["vmi"], t["intr"], t["opts"], count, out, t["origin"], s.voxel_size,
                                             workspace=ws, **kw)


def _oracle_views(oracle_mod, s, feats, views, occ=None, n_rows=None):
    occ = s.occ if occ is None else occ
    n_rows = s.n_vox + 1 if n_rows is None else n_rows
    C = feats.shape[-1]
    c, o = np.zeros(n_rows, np.int32), np.zeros((n_rows, C), np.float32)
    nviews = np.zeros(n_rows, np.int32)
    for v in views:
        c1 = np.zeros(n_rows, np.int32)
        oracle_mod.project_features(np.ascontiguousarray(feats[:, v:v + 1]).astype(np.float32), occ[None].astype(np.int64), s.c2w[v].reshape(-1),
                                    s.intr[None], s.opts(), s.grid_origin, s.voxel_size, c1, o)
        c += c1
        nviews += c1 > 0
    return c, o, nviews


@pytest.mark.parametrize("C,half", [(512, False), (512, True), (64, False), (7, False), (1000, False), (260, False), (1, False), (24, True)])
@pytest.mark.parametrize("grid", [None, 1003, 1001])
def test_one_view_calls_match_the_oracle_and_the_general_kernel(oracle_mod, C, half, grid):
    """Eight one-view calls accumulate into the same outputs (K.cu:77,88: +=): every row width class of the kernel (two 1-KiB
    loads per row, one, scalar; fp16 rows), the default grid and grids of three and of ONE workgroup -- the latter walk the
    batches of 64 entries per wavefront (every wavefront takes hundreds of voxels) -- bit for bit the oracle's serial sums;
    the general kernel (option 0) leaves the same bits."""
    import voxproj_host
    if grid is not None and C not in (512, 7):
        pytest.skip("the small grids are exercised on two row widths")
    dev = torch.device(DEV)
    V = 8
    s = make_scene(3000, V, 64, 48, seed=301 + C, room=(5.0, 4.0, 2.4))
    feats = make_features_np(V, 48, 64, C, seed=301 + C)[None]
    if half:
        feats = feats.astype(np.float16)
    n_rows = s.n_vox + 1
    ref_c, ref_o, ref_v = _oracle_views(oracle_mod, s, feats, range(V))
    res = []
    for opt in (grid, 0):
        ws = voxproj_host.Workspace()
        ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_GATHER, opt)
        ws.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, 10 ** 8)
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, C, device=dev)
        views = torch.zeros(n_rows, dtype=torch.int32, device=dev)
        for v in range(V):
            _call(_tensors(s, feats, v, dev), s, ws, count, out, sync=True, views_hit=views)
            assert voxproj_host.counters(ws, dev)["box_miss"] == 0
        res.append((count.cpu().numpy(), out.cpu().numpy(), views.cpu().numpy()))
        ws.release()
    for got_c, got_o, got_v in res:
        assert np.array_equal(got_c, ref_c) and np.array_equal(got_v, ref_v)
        assert got_o.tobytes() == ref_o.tobytes()
    assert ref_c.sum() > 0.9 * V * 64 * 48


def test_one_view_heavy_voxels_take_the_workgroup_role_of_the_same_launch(oracle_mod):
    """Threshold 6 pixels: most voxels of the view are "heavy" and go to the first workgroups of k_gather_one (four wavefronts
    per voxel), which join the deal afterwards; counts exact, sums within 1e-4 of the oracle.  (The general path sums such a
    voxel with 16 wavefronts, this one with 4: two fixed trees that differ in the last bits, so each is checked against the
    oracle, not against the other.)"""
    import voxproj_host
    dev = torch.device(DEV)
    s = make_scene(2000, 3, 48, 32, seed=331, room=(5.0, 4.0, 2.4))
    feats = make_features_np(3, 32, 48, 32, seed=331)[None]
    n_rows = s.n_vox + 1
    ref_c, ref_o, _ = _oracle_views(oracle_mod, s, feats, range(3))
    scale = np.abs(ref_o).max(axis=1, keepdims=True) + 1e-30
    for opt in (None, 1002, 0):
        ws = voxproj_host.Workspace()
        ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_GATHER, opt)
        ws.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, 6)
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 32, device=dev)
        heavy = 0
        for v in range(3):
            _call(_tensors(s, feats, v, dev), s, ws, count, out, sync=True)
            ctr = voxproj_host.counters(ws, dev)
            heavy += ctr["n_heavy"]
            assert ctr["box_miss"] == 0
        assert heavy > 100
        assert np.array_equal(count.cpu().numpy(), ref_c)
        assert (np.abs(out.cpu().numpy() - ref_o) / scale).max() <= 1e-4
        ws.release()


def test_one_view_box_misses_fall_back_to_the_whole_image(oracle_mod):
    """An ID that labels several cells: the pixel box of the one cell the table remembers misses pixels, the wavefront notices
    (phase 1 counted more) and rescans the whole image from the row as it still is in memory.  Results exact."""
    import voxproj_host
    dev = torch.device(DEV)
    s = make_scene(2000, 2, 40, 24, seed=341, room=(5.0, 4.0, 2.4))
    occ = np.where(s.occ > 0, (s.occ % 7) + 1, 0).astype(np.int32)
    feats = make_features_np(2, 24, 40, 8, seed=341)[None]
    n_rows = 9
    ref_c, ref_o, _ = _oracle_views(oracle_mod, s, feats, range(2), occ=occ, n_rows=n_rows)
    for opt in (None, 1001):
        ws = voxproj_host.Workspace()
        ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_GATHER, opt)
        ws.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, 10 ** 8)
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 8, device=dev)
        miss = 0
        for v in range(2):
            _call(_tensors(s, feats, v, dev, occ=occ), s, ws, count, out, sync=True)
            miss += voxproj_host.counters(ws, dev)["box_miss"]
        assert miss > 0
        assert np.array_equal(count.cpu().numpy(), ref_c) and out.cpu().numpy().tobytes() == ref_o.tobytes()
        ws.release()


def test_one_view_calls_in_job_mode_and_cut_into_row_ranges(oracle_mod):
    """One-view calls pipelined (the parity aggregator's pattern: march of view k+1 under the gather of view k), every other
    one cut into two row ranges with a gather-only second half."""
    import voxproj_host
    dev = torch.device(DEV)
    V = 6
    s = make_scene(2500, V, 56, 40, seed=351, room=(5.0, 4.0, 2.4))
    feats = make_features_np(V, 40, 56, 64, seed=351)[None]
    n_rows = s.n_vox + 1
    ref_c, ref_o, _ = _oracle_views(oracle_mod, s, feats, range(V))
    ws = voxproj_host.Workspace()
    count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 64, device=dev)
    ts = [_tensors(s, feats, v, dev) for v in range(V)]
    torch.cuda.synchronize()
    h = 1100
    for v in range(V):
        if v % 2:
            ws.set_row_range(0, h)
            _call(ts[v], s, ws, count, out, sync=False, pipeline=True, serial_sums=True)
            ws.set_row_range(h, n_rows)
            _call(ts[v], s, ws, count, out, sync=False, pipeline=True, serial_sums=True, gather_only=True)
            ws.set_row_range()
        else:
            _call(ts[v], s, ws, count, out, sync=False, pipeline=True, serial_sums=True)
    voxproj_host.workspace_status(ws, dev)
    assert np.array_equal(count.cpu().numpy(), ref_c) and out.cpu().numpy().tobytes() == ref_o.tobytes()
    ws.release()


def test_one_view_empty_and_tiny_outputs(oracle_mod):
    """Nothing hit (empty grid), and outputs with a single row (only the dummy row 0): no work list entry, no launch beyond
    the grid's minimum; outputs untouched."""
    import voxproj_host
    dev = torch.device(DEV)
    s = make_scene(2000, 1, 40, 24, seed=361, room=(5.0, 4.0, 2.4))
    feats = make_features_np(1, 24, 40, 8, seed=361)[None]
    t = _tensors(s, feats, 0, dev, occ=np.zeros_like(s.occ))
    ws = voxproj_host.Workspace()
    for n_rows in (s.n_vox + 1, 1):
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.full((n_rows, 8), 3.0, device=dev)
        _call(t, s, ws, count, out, sync=True)
        assert int(count.sum().item()) == 0 and float((out - 3.0).abs().max().item()) == 0.0
    ws.release()
