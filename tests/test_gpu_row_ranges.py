"""Row ranges of phase 2 (VP_OPT_ROW_BEGIN / _END + VP_FLAG_GATHER_ONLY): a call cut into [0, h) and [h, n_rows) leaves
exactly what one call leaves, bit for bit -- every voxel is summed by the same kernel role in both forms (one wavefront, or
the split of voxels above the heavy threshold into parts, planned per range) -- and the rows below h are
final as soon as the first gather is over: what a multi-GPU job needs to start their all-reduce under the second gather.
With VP_FLAG_SERIAL_SUMS those bits are the oracle's.  No counterpart in the reference (single GPU, one atomicAdd per channel)."""
import numpy as np
import pytest
import torch

from synthetic_scene import make_features_np, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(V, C, seed, half=False):
    dev = torch.device(DEV)
    s = make_scene(2000, V, 48, 32, seed=seed, room=(5.0, 4.0, 2.4))
    feats = make_features_np(V, 32, 48, C, seed=seed)[None]
    if half:
        feats = feats.astype(np.float16)
    t = dict(feats=torch.from_numpy(feats).to(dev), occ=torch.from_numpy(s.occ[None].astype(np.int64)).to(dev),
             vmi=torch.from_numpy(s.c2w).reshape(-1).to(dev), intr=torch.from_numpy(s.intr[None]).to(dev),
             opts=[float(v) for v in s.opts()], origin=[float(v) for v in s.grid_origin])
    return dev, s, feats, t


def _oracle(oracle_mod, s, feats, times=1):
    n_rows, C = s.n_vox + 1, feats.shape[-1]
    c, o = np.zeros(n_rows, np.int32), np.zeros((n_rows, C), np.float32)
    for _ in range(times):
        oracle_mod.project_features(feats.astype(np.float32), s.occ[None].astype(np.int64), s.c2w.reshape(-1), s.intr[None], s.opts(),
                                    s.grid_origin, s.voxel_size, c, o)
    return c, o


def _call(t, s, ws, count, out, **kw):
    import voxproj_host
    return voxproj_host.project_features_raw(t["feats"], t["occ"], t["vmi"], t["intr"], t["opts"], count, out, t["origin"], s.voxel_size,
                                             workspace=ws, **kw)


@pytest.mark.parametrize("V,half", [(3, False), (9, False), (9, True)])
def test_blocking_call_cut_into_row_ranges(oracle_mod, V, half):
    """Three ranges (one call + two gather-only calls), few views (separate heavy launch left out) and many (merged kernel),
    fp32 and fp16 maps; after the FIRST range its rows already hold the final sums and the others nothing."""
    import voxproj_host
    dev, s, feats, t = _setup(V, 64, seed=151 + V, half=half)
    n_rows = s.n_vox + 1
    ref_c, ref_o = _oracle(oracle_mod, s, feats)
    cuts = [0, 700, 1500, n_rows]
    count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 64, device=dev)
    views = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    ws = voxproj_host.Workspace()
    for k in range(3):
        ws.set_row_range(cuts[k], cuts[k + 1])
        _call(t, s, ws, count, out, sync=True, gather_only=k > 0, views_hit=views, serial_sums=True)
        got_c, got_o = count.cpu().numpy(), out.cpu().numpy()
        done = cuts[k + 1]
        assert np.array_equal(got_c[:done], ref_c[:done]) and got_o[:done].tobytes() == ref_o[:done].tobytes()
        assert not got_c[done:].any() and not got_o[done:].any()
    assert int((views.cpu().numpy() > 0).sum()) == int((ref_c > 0).sum())
    assert voxproj_host.counters(ws, dev)["box_miss"] == 0
    # back to whole calls on the same workspace
    ws.set_row_range()
    _call(t, s, ws, count, out, sync=True)
    assert np.array_equal(count.cpu().numpy(), 2 * ref_c)
    ws.release()


@pytest.mark.parametrize("V", [3, 9])
def test_heavy_voxels_of_a_ranged_call_take_the_workgroup_path_once(oracle_mod, V):
    """With the production threshold lowered to 6 pixels most voxels of the scene are summed in parts, planned per row range from
    the whole call's pixel histogram: every ranged gather plans the voxels of its own range, none is summed twice, and the result
    is the unsplit call's bit for bit (a voxel's parts depend on its pixel count and boxes alone)."""
    import voxproj_host
    dev, s, feats, t = _setup(V, 32, seed=163 + V)
    n_rows = s.n_vox + 1
    ref_c, ref_o = _oracle(oracle_mod, s, feats)
    res = []
    for cuts in ([0, n_rows], [0, n_rows // 3, n_rows // 2, n_rows]):
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 32, device=dev)
        ws = voxproj_host.Workspace()
        ws.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, 6)
        for k in range(len(cuts) - 1):
            if len(cuts) > 2:
                ws.set_row_range(cuts[k], cuts[k + 1])
            _call(t, s, ws, count, out, sync=True, gather_only=k > 0)
            assert voxproj_host.counters(ws, dev)["n_heavy"] > 50
            if len(cuts) > 2:
                done = cuts[k + 1]
                assert np.array_equal(count.cpu().numpy()[:done], ref_c[:done]) and not count.cpu().numpy()[done:].any()
        res.append((count.cpu().numpy(), out.cpu().numpy()))
        ws.release()
    assert np.array_equal(res[0][0], ref_c) and np.array_equal(res[1][0], ref_c)
    assert res[0][1].tobytes() == res[1][1].tobytes()                       # split == whole, bit for bit
    scale = np.abs(ref_o).max(axis=1, keepdims=True) + 1e-30
    assert (np.abs(res[0][1] - ref_o) / scale).max() <= 1e-4                # parts combined in slot order vs the serial order


def test_job_mode_with_every_call_cut_in_two(oracle_mod):
    """Pipelined calls, each followed by its gather-only half, alternating with whole calls: the buffer sets advance once per
    PAIR, the march of the next call runs beside both halves, errors stay readable."""
    import voxproj_host
    dev, s, feats, t = _setup(10, 64, seed=167)
    n_rows = s.n_vox + 1
    ref_c, ref_o = _oracle(oracle_mod, s, feats, times=5)
    count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 64, device=dev)
    ws = voxproj_host.Workspace()
    h = 900
    for k in range(5):
        if k % 2 == 0:
            ws.set_row_range(0, h)
            _call(t, s, ws, count, out, sync=False, pipeline=True)
            ws.set_row_range(h, n_rows)
            _call(t, s, ws, count, out, sync=False, pipeline=True, gather_only=True)
            ws.set_row_range()
        else:
            _call(t, s, ws, count, out, sync=False, pipeline=True)
    voxproj_host.workspace_status(ws, dev)
    torch.cuda.synchronize()
    got_c, got_o = count.cpu().numpy(), out.cpu().numpy()
    assert np.array_equal(got_c, ref_c)
    # the two whole calls of the five may split heavy voxels into parts: their sums differ in the last bits
    scale = np.abs(ref_o).max(axis=1, keepdims=True) + 1e-30
    assert (np.abs(got_o - ref_o) / scale).max() <= 1e-4
    ws.release()


def test_gather_only_is_refused_without_a_matching_predecessor():
    import voxproj_host
    dev, s, feats, t = _setup(3, 16, seed=173)
    n_rows = s.n_vox + 1
    count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, 16, device=dev)
    ws = voxproj_host.Workspace()
    ws.set_row_range(0, 100)
    with pytest.raises(voxproj_host.VoxprojError, match="repeats phase 2 of the previous call"):
        _call(t, s, ws, count, out, sync=True, gather_only=True)          # nothing precedes it
    _call(t, s, ws, count, out, sync=True)
    other = dict(t, feats=t["feats"].clone())
    with pytest.raises(voxproj_host.VoxprojError, match="repeats phase 2 of the previous call"):
        _call(other, s, ws, count, out, sync=True, gather_only=True)      # other feature maps
    ws.set_row_range()
    _call(t, s, ws, count, out, sync=True)
    with pytest.raises(voxproj_host.VoxprojError, match="without a row range"):
        _call(t, s, ws, count, out, sync=True, gather_only=True)
    # a predecessor that had no row range gathered every row already (the refused call above withdrew what a
    # gather-only call could repeat: a successful whole call first)
    _call(t, s, ws, count, out, sync=True)
    ws.set_row_range(0, 100)
    with pytest.raises(voxproj_host.VoxprojError, match="had no row range"):
        _call(t, s, ws, count, out, sync=True, gather_only=True)
    # ... other outputs than the predecessor's
    ws.set_row_range(0, 100)
    _call(t, s, ws, count, out, sync=True)
    ws.set_row_range(100, n_rows)
    with pytest.raises(voxproj_host.VoxprojError, match="repeats phase 2 of the previous call"):
        _call(t, s, ws, count, torch.zeros_like(out), sync=True, gather_only=True)
    # ... and a predecessor that FAILED leaves nothing to repeat (its arguments matched an older call's before)
    ws.set_row_range(0, 100)
    _call(t, s, ws, count, out, sync=True)
    bad = dict(t, opts=t["opts"][:4] + [0.0])                             # rayIncrement 0: refused before any launch
    with pytest.raises(voxproj_host.VoxprojError, match="rayIncrement"):
        _call(bad, s, ws, count, out, sync=True)
    ws.set_row_range(100, n_rows)
    with pytest.raises(voxproj_host.VoxprojError, match="repeats phase 2 of the previous call"):
        _call(t, s, ws, count, out, sync=True, gather_only=True)
    ws.release()
