"""Maximum sizes: output rows beyond the 2^31-element and the 4-GiB marks.  A small scene whose voxel IDs are spread over
4.4 million rows of 512 channels (a 9-GB `out`): every row address (`id * C`, the write-through row descriptor, the
per-voxel tables, the work list of eight size classes x n_rows) is computed for IDs whose element index exceeds 2^31 and
whose byte offset exceeds 2^33.  Checked against the oracle on the touched rows, and through an exact non-zero count over
the WHOLE buffer (a store that went to a truncated address lands on an untouched row)."""
import numpy as np
import pytest
import torch

from synthetic_scene import make_features_np, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SPREAD = 2200                      # ID k of the scene becomes ID k * SPREAD
C = 512


def _case(V, seed):
    dev = torch.device(DEV)
    s = make_scene(2000, V, 48, 32, seed=seed, room=(5.0, 4.0, 2.4))
    occ = s.occ[None].astype(np.int64) * SPREAD
    n_rows = s.n_vox * SPREAD + 1
    assert (n_rows - 1) * C > 2**31 and (n_rows - 1) * C * 4 > 2**33
    return dev, s, occ, n_rows


def _oracle(oracle_mod, s, feats, occ, n_rows, repeats=1, splits=None):
    # np.zeros is lazily committed: only the touched rows of these 9 GB ever become resident
    ref_c, ref_o = np.zeros(n_rows, np.int32), np.zeros((n_rows, C), np.float32)
    V = feats.shape[1]
    for _ in range(repeats):
        for a, b in (splits or [(0, V)]):
            oracle_mod.project_features(np.ascontiguousarray(feats[:, a:b]), occ, s.c2w[a:b].reshape(-1), s.intr[None], s.opts(),
                                        s.grid_origin, s.voxel_size, ref_c, ref_o)
    return ref_c, ref_o


def _check(count_t, out_t, ref_c, ref_o, bitwise=True):
    assert np.array_equal(count_t.cpu().numpy(), ref_c)
    rows = np.nonzero(ref_c)[0]
    assert rows.size > 500 and rows.max() * C > 2**31
    got = out_t[torch.from_numpy(rows).to(out_t.device)].cpu().numpy()
    want = ref_o[rows]
    if bitwise:
        assert got.tobytes() == want.tobytes()
    else:
        scale = np.abs(want).max(axis=1, keepdims=True) + 1e-30
        assert (np.abs(got - want) / scale).max() <= 1e-4
    # nothing was written anywhere else in the 9 GB
    assert int(torch.count_nonzero(out_t).item()) == int(np.count_nonzero(got))


def test_drop_in_call_with_rows_past_4_gib(oracle_mod, heavy_threshold):
    """Few views per call, heavy threshold 6: most voxels are summed in parts whose partial rows k_combine_parts adds up."""
    import voxproj_host
    heavy_threshold(6)
    dev, s, occ, n_rows = _case(3, seed=131)
    feats = make_features_np(3, 32, 48, C, seed=131)[None]
    ref_c, ref_o = _oracle(oracle_mod, s, feats, occ, n_rows)
    count_t = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    out_t = torch.zeros(n_rows, C, device=dev)
    ws = voxproj_host.Workspace()
    voxproj_host.project_features_raw(torch.from_numpy(feats).to(dev), torch.from_numpy(occ).to(dev),
                                      torch.from_numpy(s.c2w).reshape(-1).to(dev), torch.from_numpy(s.intr[None]).to(dev),
                                      [float(v) for v in s.opts()], count_t, out_t, [float(v) for v in s.grid_origin], s.voxel_size,
                                      workspace=ws, sync=True)
    assert voxproj_host.counters(ws, dev)["n_heavy"] > 0
    _check(count_t, out_t, ref_c, ref_o, bitwise=False)      # the heavy path sums in a fixed tree of its own
    ws.release()


@pytest.mark.parametrize("half", [False, True])
def test_job_mode_with_rows_past_4_gib(oracle_mod, heavy_threshold, half):
    """Pipelined calls of 8 and 4 views, twice over (accumulation into rows that already hold sums): the merged gather with
    its write-through row stores, fp32 and fp16 feature maps, serial sums -> the oracle's bits."""
    import voxproj_host
    heavy_threshold(100000000)
    dev, s, occ, n_rows = _case(12, seed=137)
    feats = make_features_np(12, 32, 48, C, seed=137)[None]
    if half:
        feats = feats.astype(np.float16)
    splits = [(0, 8), (8, 12)]
    ref_c, ref_o = _oracle(oracle_mod, s, feats.astype(np.float32), occ, n_rows, repeats=2, splits=splits)
    count_t = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    out_t = torch.zeros(n_rows, C, device=dev)
    ws = voxproj_host.Workspace()
    f_t, occ_t = torch.from_numpy(feats).to(dev), torch.from_numpy(occ).to(dev)
    intr_t, c2w_t = torch.from_numpy(s.intr[None]).to(dev), torch.from_numpy(s.c2w).to(dev)
    vm = [c2w_t[a:b].reshape(-1).contiguous() for a, b in splits]
    for rep in range(2):
        for (a, b), v in zip(splits, vm):
            voxproj_host.project_features_raw(f_t[:, a:b], occ_t, v, intr_t, [float(x) for x in s.opts()], count_t, out_t,
                                              [float(x) for x in s.grid_origin], s.voxel_size, workspace=ws, sync=False,
                                              pipeline=True)
    voxproj_host.workspace_status(ws, dev)
    torch.cuda.synchronize()
    _check(count_t, out_t, ref_c, ref_o)
    ws.release()


def test_occupancy_grid_just_under_2_31_cells(oracle_mod):
    """A 1200 x 1300 x 1300 grid (2.03e9 cells, a 16-GB int64 tensor; the library refuses 2^31 and more) with the scene in
    its far corner: every cell index the march, the table builders and the gather's cube projection form is close to
    2^31.  World coordinates are those of the small scene (the grid origin moves instead), so the oracle marches the same
    rays through a lazily committed 16-GB array; hit images, counts and sums must be its bits."""
    import voxproj_host
    dev = torch.device(DEV)
    V, Cs = 3, 64
    s = make_scene(2000, V, 48, 32, seed=139, room=(5.0, 4.0, 2.4))
    nz, ny, nx = s.occ.shape
    Z, Y, X = 1200, 1300, 1300
    assert Z * Y * X < 2**31 < (Z + 80) * Y * X
    oz, oy, ox = Z - nz, Y - ny, X - nx
    origin = (s.grid_origin.astype(np.float64) - np.array([ox, oy, oz]) * s.voxel_size).astype(np.float32)
    occ = np.zeros((1, Z, Y, X), np.int64)                 # calloc: only the corner's pages become resident
    occ[0, oz:, oy:, ox:] = s.occ
    feats = make_features_np(V, 32, 48, Cs, seed=139)[None]
    n_rows = s.n_vox + 1
    ref_c, ref_o = np.zeros(n_rows, np.int32), np.zeros((n_rows, Cs), np.float32)
    r = oracle_mod.project_features(feats, occ, s.c2w.reshape(-1), s.intr[None], s.opts(), origin, s.voxel_size, ref_c, ref_o)
    assert r["rc"] == 0 and (r["hits"] > 0).mean() > 0.9
    occ_t = torch.zeros((1, Z, Y, X), dtype=torch.int64, device=dev)
    occ_t[0, oz:, oy:, ox:] = torch.from_numpy(s.occ.astype(np.int64)).to(dev)
    count_t = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    out_t = torch.zeros(n_rows, Cs, device=dev)
    ws = voxproj_host.Workspace()
    args = (torch.from_numpy(feats).to(dev), occ_t, torch.from_numpy(s.c2w).reshape(-1).to(dev), torch.from_numpy(s.intr[None]).to(dev),
            [float(v) for v in s.opts()], count_t, out_t, [float(v) for v in origin], s.voxel_size)
    voxproj_host.project_features_raw(*args, workspace=ws, sync=True)
    assert np.array_equal(voxproj_host.hit_image(ws, dev).cpu().numpy(), r["hits"])
    assert np.array_equal(count_t.cpu().numpy(), ref_c)
    assert out_t.cpu().numpy().tobytes() == ref_o.tobytes()
    ctr = voxproj_host.counters(ws, dev)
    assert ctr["bad_id"] == 0 and ctr["box_miss"] == 0
    # the literal sample-by-sample march finds the same image in the same grid
    count_t.zero_(); out_t.zero_()
    voxproj_host.project_features_raw(*args, workspace=ws, sync=True, exact_march=True)
    assert np.array_equal(voxproj_host.hit_image(ws, dev).cpu().numpy(), r["hits"])
    assert out_t.cpu().numpy().tobytes() == ref_o.tobytes()
    ws.release()
    del occ_t
    torch.cuda.empty_cache()


def test_grids_of_2_31_cells_and_more_are_refused():
    """The limit is checked on the host before anything is read: VP_EINVAL for 2048 x 1024 x 1024 cells (the arguments point
    at a small grid -- nothing dereferences them)."""
    import ctypes

    import voxproj_host
    dev = torch.device(DEV)
    ws = voxproj_host.Workspace()
    ptr = ws.ensure(1 << 20, dev)
    small = torch.zeros(64, device=dev)
    o, g = (ctypes.c_float * 5)(48, 32, 0.01, 10.0, 0.05), (ctypes.c_float * 3)(0, 0, 0)
    rc = voxproj_host.lib().vp_project_features(small.data_ptr(), small.data_ptr(), small.data_ptr(), small.data_ptr(), o,
                                                small.data_ptr(), small.data_ptr(), None, g, ctypes.c_float(0.1),
                                                1, 1, 32, 48, 8, 2048, 1024, 1024, 8, ptr, ws.capacity(),
                                                torch.cuda.current_stream(dev).cuda_stream, 0)
    assert rc == -1                                       # VP_EINVAL
    with pytest.raises(voxproj_host.VoxprojError, match="2\\^31 cells"):
        voxproj_host.check(rc)
    ws.release()
