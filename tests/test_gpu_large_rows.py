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
    """Few views per call: the gather variant without the heavy role, heavy voxels by `k_gather_heavy` (threshold 6)."""
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
