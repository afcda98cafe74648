"""The gather of ONE-VIEW calls (k_gather_one, csrc/vp_gather.h): what the drop-in module issues once per image
(debug_project_features.py:201-208) and the parity aggregator once per view.  A fixed grid of wavefronts is dealt the
size-ordered work list, lane j of a wavefront prepares its j-th voxel (ID, pixel count, pixel box), the next voxel's ID tile
and output row are fetched under the current voxel's rows.  Every case against the oracle bit for bit, and against the general
gather kernel (VP_OPT_ONE_VIEW_GATHER = 0, round 3's path for these calls)."""
import numpy as np
import pytest
import torch

from sum_criteria import assert_sums
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


def _oracle_views(oracle_mod, s, feats, views, occ=None, n_rows=None, sums64=None):
    """One oracle call per view, accumulating like the kernel does (K.cu:77,88: +=).  ``sums64``: optional dict that receives the
    float64 accumulation 'ref64' and the float64 sum of |addend| 'abs64' of the same pixels (tests/sum_criteria.py)."""
    occ = s.occ if occ is None else occ
    n_rows = s.n_vox + 1 if n_rows is None else n_rows
    C = feats.shape[-1]
    c, o = np.zeros(n_rows, np.int32), np.zeros((n_rows, C), np.float32)
    nviews = np.zeros(n_rows, np.int32)
    if sums64 is not None:
        sums64["ref64"], sums64["abs64"] = np.zeros((n_rows, C)), np.zeros((n_rows, C))
    for v in views:
        c1 = np.zeros(n_rows, np.int32)
        f = np.ascontiguousarray(feats[:, v:v + 1]).astype(np.float32)
        r = oracle_mod.project_features(f, occ[None].astype(np.int64), s.c2w[v].reshape(-1), s.intr[None], s.opts(), s.grid_origin,
                                        s.voxel_size, c1, o, want_f64=sums64 is not None)
        c += c1
        nviews += c1 > 0
        if sums64 is not None:
            sums64["ref64"] += r["out64"]
            np.add.at(sums64["abs64"], r["hits"][0, 0].reshape(-1), np.abs(f[0, 0].reshape(-1, C)).astype(np.float64))
            sums64["abs64"][0] = 0
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
    """Round 5's path for the large voxels of a one-view call, kept as the A/B arm (VP_OPT_ONE_VIEW_SPLIT = 0).  Threshold 6 pixels:
    most voxels of the view are "heavy" and go to the workgroups of k_gather_one (four wavefronts
    per voxel), which join the deal afterwards; counts exact, sums within 1e-4 of the oracle.  (The general path sums such a
    voxel in parts, this one with 4 wavefronts: two fixed trees that differ in the last bits, so each is checked against the
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
        ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_SPLIT, 0)
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


@pytest.mark.parametrize("C,half,heavy_t,split_t,part_px", [
    (32, False, None, 12, 5), (512, False, 6, None, 3), (512, True, None, 20, 7), (64, False, 10 ** 8, 8, 1), (7, False, 6, 30, 40),
    (1000, False, None, 9, None), (32, False, None, None, 6)])
def test_one_view_split_voxels_match_the_oracle(oracle_mod, C, half, heavy_t, split_t, part_px):
    """Round 6: a one-view call cuts the voxels above a threshold into parts (one wavefront of k_gather_one per part,
    k_combine_parts behind it); the others stay with the one-wavefront deal.  The threshold is VP_OPT_ONE_VIEW_SPLIT, else
    VP_OPT_HEAVY_THRESHOLD, else twice the part size (at least 256 pixels on a view this small); the part size
    VP_OPT_PART_PIXELS, else half the threshold (rounded up).
    Counts and views-hit exact, the plan as the options say (P = ceil(c / part_px) per voxel above max(threshold, part)), rows of
    one-wavefront voxels the oracle's bits, the others within 1e-4 of each row's largest element, two runs bit-identical."""
    import voxproj_host
    dev = torch.device(DEV)
    V = 3
    s = make_scene(2000, V, 48, 32, seed=371 + C, room=(5.0, 4.0, 2.4))
    feats = make_features_np(V, 32, 48, C, seed=371 + C)[None]
    if half:
        feats = feats.astype(np.float16)
    n_rows = s.n_vox + 1
    s64 = {}
    ref_c, ref_o, ref_v = _oracle_views(oracle_mod, s, feats, range(V), sums64=s64)
    T = split_t if split_t is not None else heavy_t
    px = part_px if part_px is not None else (T + 1) // 2
    part_t = max(T, px) if T is not None else max(2 * px, 256)      # no threshold given: twice the part, 256 at least on a small view
    runs = []
    for grid in (None, 1002, None):
        ws = voxproj_host.Workspace()
        ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_GATHER, grid)
        ws.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, heavy_t)
        ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_SPLIT, split_t)
        ws.set_option(voxproj_host.VP_OPT_PART_PIXELS, part_px)
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, C, device=dev)
        views = torch.zeros(n_rows, dtype=torch.int32, device=dev)
        n_split = 0
        light = np.ones(n_rows, bool)       # voxels that never left the one-wavefront deal
        for v in range(V):
            c1, _, _ = _oracle_views(oracle_mod, s, feats, [v])
            _call(_tensors(s, feats, v, dev), s, ws, count, out, sync=True, views_hit=views)
            ctr = voxproj_host.counters(ws, dev)
            assert ctr["box_miss"] == 0
            assert (ctr["part_t"], ctr["part_px"], ctr["heavy_t"]) == (part_t, px, part_t)
            big = c1[c1 > part_t]
            assert ctr["n_split"] == len(big) and ctr["n_parts"] == int(np.sum((big + px - 1) // px))
            assert ctr["n_heavy"] == len(big)
            n_split += len(big)
            light &= c1 <= part_t
        assert n_split >= (5 if T is not None else 1)
        got_c, got_o, got_v = count.cpu().numpy(), out.cpu().numpy(), views.cpu().numpy()
        assert np.array_equal(got_c, ref_c) and np.array_equal(got_v, ref_v)
        assert got_o[light].tobytes() == ref_o[light].tobytes()
        # the split rows by the one criterion (tests/sum_criteria.py): forward-error bound on every element, 1e-4 of the row, and no
        # worse than twice the serial float32 order's own distance from the float64 sum
        assert_sums(got_o, s64["ref64"], s64["abs64"], ref_c, split=~light, oracle32=ref_o, dev=dev)
        runs.append(got_o)
        ws.release()
    assert runs[0].tobytes() == runs[2].tobytes()      # same grid: the same fixed summation tree


def test_one_view_parts_are_sized_from_the_views_hit_total(oracle_mod):
    """No option set: the march counts the pixels whose ray hit a voxel, k_worklist sizes the parts from it -- part_px = max(32,
    ceil(2 * hits / slots)) with 8192 slots for a one-view call of this size, threshold twice that but at least 256 pixels for a
    view of up to 262144 pixels.  A 512 x 384 view of a small room (voxels of a thousand pixels and more), and the same room
    with most of it removed (few hits: the smallest parts); blocking calls (the combine is launched only if the gather's note
    says a voxel was split) and job-mode calls (always launched) leave the same bits."""
    import voxproj_host
    dev = torch.device(DEV)
    s = make_scene(2000, 1, 512, 384, seed=401, room=(5.0, 4.0, 2.4))
    feats = make_features_np(1, 384, 512, 16, seed=401)[None]
    n_rows = s.n_vox + 1
    sparse = np.where(s.occ % 5 == 0, s.occ, 0).astype(s.occ.dtype)
    for occ in (s.occ, sparse):
        ref_c, ref_o, _ = _oracle_views(oracle_mod, s, feats, [0], occ=occ)
        hits = int(ref_c.sum())
        px = max(32, -(-2 * hits // 8192))
        T = max(2 * px, 256)
        ws = voxproj_host.Workspace()
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 16, device=dev)
        t = _tensors(s, feats, 0, dev, occ=occ)
        _call(t, s, ws, count, out, sync=True)
        ctr = voxproj_host.counters(ws, dev)
        assert ctr["n_hit"] == hits and ctr["part_px"] == px and ctr["part_t"] == T and ctr["heavy_t"] == T
        big = ref_c[ref_c > T]
        assert len(big) > 20 and ctr["n_split"] == len(big) and ctr["n_parts"] == int(np.sum((big + px - 1) // px)) <= 8192
        assert np.array_equal(count.cpu().numpy(), ref_c)
        scale = np.abs(ref_o).max(axis=1, keepdims=True) + 1e-30
        got = out.cpu().numpy()
        assert (np.abs(got - ref_o) / scale).max() <= 1e-4
        count2, out2 = torch.zeros_like(count), torch.zeros_like(out)
        _call(t, s, ws, count2, out2, sync=False, pipeline=True)
        voxproj_host.workspace_status(ws, dev)
        assert torch.equal(count2, count) and out2.cpu().numpy().tobytes() == got.tobytes()
        ws.release()
    assert hits < 196608 // 2      # the sparse room really is mostly misses (its parts are the smallest)


def test_one_view_split_voxels_fp16_equals_fp32_bit_for_bit(oracle_mod):
    """The fp16 feature-map mode widens exactly and sums in the same order, parts included."""
    import voxproj_host
    dev = torch.device(DEV)
    s = make_scene(2000, 2, 48, 32, seed=381, room=(5.0, 4.0, 2.4))
    f16 = make_features_np(2, 32, 48, 64, seed=381)[None].astype(np.float16)
    n_rows = s.n_vox + 1
    res = []
    for feats in (f16, f16.astype(np.float32)):
        ws = voxproj_host.Workspace()
        ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_SPLIT, 10)
        ws.set_option(voxproj_host.VP_OPT_PART_PIXELS, 4)
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 64, device=dev)
        for v in range(2):
            _call(_tensors(s, feats, v, dev), s, ws, count, out, sync=True)
            assert voxproj_host.counters(ws, dev)["n_parts"] > 0
        res.append((count.cpu().numpy(), out.cpu().numpy()))
        ws.release()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1].tobytes() == res[1][1].tobytes()


def test_one_view_split_voxels_box_misses_are_redone_by_the_combine(oracle_mod):
    """An ID that labels several cells and is large enough to be split: its parts scan the box of the one cell the table remembers,
    find fewer pixels than phase 1 counted, and k_combine_parts redoes the voxel over the whole image.  Counts exact, sums within
    1e-4 (the redo sums with four wavefronts)."""
    import voxproj_host
    dev = torch.device(DEV)
    s = make_scene(2000, 2, 40, 24, seed=341, room=(5.0, 4.0, 2.4))
    occ = np.where(s.occ > 0, (s.occ % 7) + 1, 0).astype(np.int32)
    feats = make_features_np(2, 24, 40, 8, seed=341)[None]
    n_rows = 9
    ref_c, ref_o, _ = _oracle_views(oracle_mod, s, feats, range(2), occ=occ, n_rows=n_rows)
    ws = voxproj_host.Workspace()
    ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_SPLIT, 16)
    ws.set_option(voxproj_host.VP_OPT_PART_PIXELS, 8)
    count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 8, device=dev)
    miss = split = 0
    for v in range(2):
        _call(_tensors(s, feats, v, dev, occ=occ), s, ws, count, out, sync=True)
        ctr = voxproj_host.counters(ws, dev)
        miss += ctr["box_miss"]
        split += ctr["n_split"]
    assert miss > 0 and split > 0
    assert np.array_equal(count.cpu().numpy(), ref_c)
    scale = np.abs(ref_o).max(axis=1, keepdims=True) + 1e-30
    assert (np.abs(out.cpu().numpy() - ref_o) / scale).max() <= 1e-4
    ws.release()


def test_one_view_split_voxels_in_job_mode_and_row_ranges(oracle_mod):
    """One-view calls with split voxels pipelined (march of view k+1 under the gather of view k), every other one cut into two
    row ranges with a gather-only second half: the same bits as the blocking, uncut calls."""
    import voxproj_host
    dev = torch.device(DEV)
    V = 6
    s = make_scene(2500, V, 56, 40, seed=391, room=(5.0, 4.0, 2.4))
    feats = make_features_np(V, 40, 56, 64, seed=391)[None]
    n_rows = s.n_vox + 1
    ref_c, _, ref_v = _oracle_views(oracle_mod, s, feats, range(V))
    ts = [_tensors(s, feats, v, dev) for v in range(V)]
    res = []
    for mode in ("blocking", "job"):
        ws = voxproj_host.Workspace()
        ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_SPLIT, 14)
        ws.set_option(voxproj_host.VP_OPT_PART_PIXELS, 6)
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 64, device=dev)
        views = torch.zeros(n_rows, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        h = 1100
        for v in range(V):
            if mode == "blocking":
                _call(ts[v], s, ws, count, out, sync=True, views_hit=views)
                assert voxproj_host.counters(ws, dev)["n_parts"] > 0
            elif v % 2:
                ws.set_row_range(0, h)
                _call(ts[v], s, ws, count, out, sync=False, pipeline=True, views_hit=views)
                ws.set_row_range(h, n_rows)
                _call(ts[v], s, ws, count, out, sync=False, pipeline=True, gather_only=True, views_hit=views)
                ws.set_row_range()
            else:
                _call(ts[v], s, ws, count, out, sync=False, pipeline=True, views_hit=views)
        voxproj_host.workspace_status(ws, dev)
        res.append((count.cpu().numpy(), out.cpu().numpy(), views.cpu().numpy()))
        ws.release()
    assert np.array_equal(res[0][0], ref_c) and np.array_equal(res[1][0], ref_c)
    assert np.array_equal(res[0][2], ref_v) and np.array_equal(res[1][2], ref_v)
    assert res[0][1].tobytes() == res[1][1].tobytes()


def test_blocking_one_view_call_behind_a_busy_stream_still_combines_its_parts(oracle_mod):
    """A blocking one-view call launches k_combine_parts only when the gather's first wavefront reported split voxels through the
    record's pinned page -- and launches it anyway when that note has not arrived within 2 ms.  Here the caller's stream is kept
    busy for tens of milliseconds in front of the call (the gather cannot start, the note cannot arrive in time), then an idle
    stream, then a view WITHOUT split voxels between two views with them on the same workspace (the note of one call must never
    be taken for another's): every variant leaves the bits of the job-mode call, which always launches the combine."""
    import voxproj_host
    dev = torch.device(DEV)
    s = make_scene(2000, 3, 48, 32, seed=403, room=(5.0, 4.0, 2.4))
    feats = make_features_np(3, 32, 48, 64, seed=403)[None]
    n_rows = s.n_vox + 1
    empty = np.zeros_like(s.occ)
    ref_c, ref_o, _ = _oracle_views(oracle_mod, s, feats, [0, 2])

    def run(mode):
        ws = voxproj_host.Workspace()
        ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_SPLIT, 10)
        ws.set_option(voxproj_host.VP_OPT_PART_PIXELS, 4)
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 64, device=dev)
        busy = torch.randn(4096, 4096, device=dev)
        for v, occ in ((0, s.occ), (1, empty), (2, s.occ)):
            t = _tensors(s, feats, v, dev, occ=occ)
            if mode == "busy":
                for _ in range(40):
                    busy = busy @ busy * 1e-4          # ~20+ ms of work queued in front of the call on the same stream
            if mode == "job":
                _call(t, s, ws, count, out, sync=False, pipeline=True)
                voxproj_host.workspace_status(ws, dev)
            else:
                _call(t, s, ws, count, out, sync=True)
            ctr = voxproj_host.counters(ws, dev)
            assert (ctr["n_split"] == 0 and ctr["n_parts"] == 0) if v == 1 else (ctr["n_split"] >= 1 and ctr["n_parts"] > 20), ctr
        res = (count.cpu().numpy(), out.cpu().numpy())
        ws.release()
        return res

    job, idle, busy = run("job"), run("idle"), run("busy")
    assert np.array_equal(job[0], ref_c) and np.array_equal(idle[0], ref_c) and np.array_equal(busy[0], ref_c)
    assert job[1].tobytes() == idle[1].tobytes() == busy[1].tobytes()
    scale = np.abs(ref_o).max(axis=1, keepdims=True) + 1e-30
    assert (np.abs(job[1] - ref_o) / scale).max() <= 1e-4


def test_two_threads_issue_blocking_one_view_calls_on_their_own_workspaces(oracle_mod):
    """Two host threads, each with its own workspace, stream and outputs, issue blocking one-view calls with split voxels at the same
    time (ctypes releases the GIL around the foreign call; the compiled front does too): every workspace record has its own pinned
    page, so each thread polls its own gather's note.  Both threads must leave exactly what a single thread leaves."""
    import threading
    import voxproj_host
    dev = torch.device(DEV)
    V = 6
    s = make_scene(2000, V, 48, 32, seed=405, room=(5.0, 4.0, 2.4))
    feats = make_features_np(V, 32, 48, 32, seed=405)[None]
    n_rows = s.n_vox + 1
    ts = [_tensors(s, feats, v, dev) for v in range(V)]

    def work(res, key, stream):
        ws = voxproj_host.Workspace()
        ws.set_option(voxproj_host.VP_OPT_ONE_VIEW_SPLIT, 8)
        ws.set_option(voxproj_host.VP_OPT_PART_PIXELS, 3)
        with torch.cuda.stream(stream):
            count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 32, device=dev)
            stream.synchronize()
            for rep in range(5):
                for v in range(V):
                    _call(ts[v], s, ws, count, out, sync=True)
            res[key] = (count.cpu().numpy(), out.cpu().numpy())
        ws.release()

    res = {}
    work(res, "alone", torch.cuda.Stream(dev))
    threads = [threading.Thread(target=work, args=(res, k, torch.cuda.Stream(dev))) for k in ("a", "b")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    ref_c, _, _ = _oracle_views(oracle_mod, s, feats, range(V))
    assert np.array_equal(res["alone"][0], 5 * ref_c)
    for k in ("a", "b"):
        assert np.array_equal(res[k][0], res["alone"][0]) and res[k][1].tobytes() == res["alone"][1].tobytes()
