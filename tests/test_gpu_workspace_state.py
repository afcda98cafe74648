"""The library's per-workspace state is owned by the workspace, not by its address (ABI v3): a header in the workspace memory
carries the generation of the record that initialised it and a key of the tables it holds, every call compares them on the
device, options are per workspace, and sticky errors are reported one at a time without erasing each other."""
import numpy as np
import pytest
import torch

from synthetic_scene import make_features_np, make_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _scene(seed=101, C=8, V=3):
    dev = torch.device(DEV)
    s = make_scene(2000, V, 48, 32, seed=seed, room=(5.0, 4.0, 2.4))
    feats = make_features_np(V, 32, 48, C, seed=seed)[None]
    t = dict(feats=torch.from_numpy(feats).to(dev), occ=torch.from_numpy(s.occ[None].astype(np.int64)).to(dev),
             vmi=torch.from_numpy(s.c2w).reshape(-1).to(dev), intr=torch.from_numpy(s.intr[None]).to(dev),
             opts=[float(v) for v in s.opts()], origin=[float(v) for v in s.grid_origin])
    return s, feats, t


def _call(t, s, ws, count, out, **kw):
    import voxproj_host
    return voxproj_host.project_features_raw(t["feats"], t["occ"], t["vmi"], t["intr"], t["opts"], count, out, t["origin"], s.voxel_size,
                                             workspace=ws, **kw)


@pytest.mark.parametrize("fill", [0, 0xA5])
def test_recycled_workspace_memory_is_recognised_not_trusted(oracle_mod, fill):
    """A caller frees a workspace without vp_workspace_release and gets the address back (here: the same buffer, overwritten
    -- what recycled memory looks like to the library).  The address still has a record with builds > 0, so
    VP_FLAG_REUSE_ACCEL passes the host-side check; the device-side header check must catch it: no work is done, the call
    reports VP_EINVAL, nothing is written to the outputs, and a call without the flag rebuilds and is exact again."""
    import voxproj_host
    dev = torch.device(DEV)
    s, feats, t = _scene()
    n_rows, C = s.n_vox + 1, feats.shape[-1]
    ref_count, ref_out = np.zeros(n_rows, np.int32), np.zeros((n_rows, C), np.float32)
    oracle_mod.project_features(feats, s.occ[None].astype(np.int64), s.c2w.reshape(-1), s.intr[None], s.opts(), s.grid_origin, s.voxel_size,
                                ref_count, ref_out)
    ws = voxproj_host.Workspace()
    count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, C, device=dev)
    _call(t, s, ws, count, out, sync=True)
    assert np.array_equal(count.cpu().numpy(), ref_count) and voxproj_host.table_builds(ws) == 1
    ws.buf.fill_(fill)                                    # the memory went through someone else's hands
    torch.cuda.synchronize()
    count.zero_(); out.zero_()
    with pytest.raises(voxproj_host.VoxprojError, match="no longer holds the tables"):
        _call(t, s, ws, count, out, sync=True, reuse_accel=True)
    assert int(count.sum().item()) == 0 and float(out.abs().sum().item()) == 0.0, "a stale call must not write outputs"
    voxproj_host.workspace_status(ws, dev)               # reported once; garbage in the other sticky words is not an error
    # the record no longer vouches for tables: the flag is refused on the host until a rebuild
    with pytest.raises(voxproj_host.VoxprojError, match="holds no occupancy tables"):
        _call(t, s, ws, count, out, sync=True, reuse_accel=True)
    _call(t, s, ws, count, out, sync=True, reuse_accel=False)
    assert np.array_equal(count.cpu().numpy(), ref_count) and out.cpu().numpy().tobytes() == ref_out.tobytes()
    _call(t, s, ws, count, out, sync=True, reuse_accel=True)          # and trusted again afterwards
    assert np.array_equal(count.cpu().numpy(), 2 * ref_count)
    # pipelined calls on overwritten memory: the error is sticky until the job asks
    ws.buf.fill_(fill)
    torch.cuda.synchronize()
    count.zero_()
    for _ in range(3):
        _call(t, s, ws, count, out, sync=False, pipeline=True, reuse_accel=True)
    with pytest.raises(voxproj_host.VoxprojError, match="no longer holds the tables"):
        voxproj_host.workspace_status(ws, dev)
    assert int(count.sum().item()) == 0
    ws.release()


def test_workspace_create_starts_a_new_generation(oracle_mod):
    """vp_workspace_create on an address the library knows drops its record: tables, streams and options are gone, the
    memory's old header belongs to a generation nobody expects any more."""
    import voxproj_host
    dev = torch.device(DEV)
    s, feats, t = _scene(seed=103)
    n_rows, C = s.n_vox + 1, feats.shape[-1]
    ws = voxproj_host.Workspace()
    count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, C, device=dev)
    _call(t, s, ws, count, out, sync=True)
    assert voxproj_host.table_builds(ws) == 1
    voxproj_host.check(voxproj_host.lib().vp_workspace_create(ws.ptr(), ws.capacity()))
    assert voxproj_host.table_builds(ws) == 0
    with pytest.raises(voxproj_host.VoxprojError, match="holds no occupancy tables"):
        _call(t, s, ws, count, out, sync=True, reuse_accel=True)
    total = int(count.sum().item())
    _call(t, s, ws, count, out, sync=True, reuse_accel=False)        # the old header is overwritten by the new generation
    assert int(count.sum().item()) == 2 * total and voxproj_host.table_builds(ws) == 1
    ws.release()
    assert voxproj_host.table_builds(ws) == 0


def test_sticky_errors_are_reported_one_by_one_and_never_erase_each_other():
    """a stuck-ray error and an out-of-range ID raised in the same interval are BOTH reported, by two successive
    status reads (the header promises that no later call -- and no other report -- erases them)."""
    import voxproj_host
    dev = torch.device(DEV)
    s, feats, t = _scene(seed=105)
    n_rows, C = s.n_vox + 1, feats.shape[-1]
    ws = voxproj_host.Workspace()
    count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, C, device=dev)
    bad = dict(t, occ=t["occ"] + (t["occ"] > 0) * 5000)               # every ID beyond the outputs' rows
    stuck = dict(t, opts=t["opts"][:4] + [1e-8])                      # increment below half an ulp of t
    _call(bad, s, ws, count, out, sync=False, pipeline=True)
    _call(stuck, s, ws, count, out, sync=False, pipeline=True)
    _call(t, s, ws, count, out, sync=False, pipeline=True)
    with pytest.raises(voxproj_host.VoxprojError, match="never terminate"):
        voxproj_host.workspace_status(ws, dev)
    with pytest.raises(voxproj_host.VoxprojError, match="outside"):
        voxproj_host.workspace_status(ws, dev)
    voxproj_host.workspace_status(ws, dev)
    ws.release()


def test_options_belong_to_the_workspace(oracle_mod):
    """VP_OPT_HEAVY_THRESHOLD set on one workspace does not leak into another (ABI v2 read an environment variable on
    every call): with threshold 6 the scene has heavy voxels (n_heavy > 0), the neighbour keeps the default (none)."""
    import voxproj_host
    dev = torch.device(DEV)
    s, feats, t = _scene(seed=107, V=5)
    n_rows, C = s.n_vox + 1, feats.shape[-1]
    ws_a, ws_b = voxproj_host.Workspace(), voxproj_host.Workspace()
    ws_a.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, 6)
    res = []
    for ws in (ws_a, ws_b):
        count, out = torch.zeros(n_rows, dtype=torch.int32, device=dev), torch.zeros(n_rows, C, device=dev)
        _call(t, s, ws, count, out, sync=True)
        res.append((voxproj_host.counters(ws, dev)["n_heavy"], count.cpu().numpy(), out.cpu().numpy()))
    assert res[0][0] > 0 and res[1][0] == 0
    assert np.array_equal(res[0][1], res[1][1])
    scale = np.abs(res[1][2]).max(axis=1, keepdims=True) + 1e-30
    assert (np.abs(res[0][2] - res[1][2]) / scale).max() <= 1e-4
    with pytest.raises(voxproj_host.VoxprojError, match="unknown workspace option"):
        voxproj_host.check(voxproj_host.lib().vp_workspace_set_option(ws_a.ptr(), 99, 1))
    ws_a.release(); ws_b.release()


def test_two_threads_on_two_workspaces(oracle_mod):
    """The library's records are per workspace and its registry is mutex-guarded: two host threads, each with its own
    workspace, stream and outputs, issue pipelined calls at the same time (ctypes releases the GIL during the foreign call)
    and both get the oracle's counts and bit-identical sums."""
    import threading

    import voxproj_host
    dev = torch.device(DEV)
    jobs = []
    for seed in (111, 113):
        s, feats, t = _scene(seed=seed, V=6)
        n_rows, C = s.n_vox + 1, feats.shape[-1]
        ref_c, ref_o = np.zeros(n_rows, np.int32), np.zeros((n_rows, C), np.float32)
        for _ in range(4):
            for a, b in ((0, 2), (2, 6)):
                oracle_mod.project_features(feats[:, a:b], s.occ[None].astype(np.int64), s.c2w[a:b].reshape(-1), s.intr[None], s.opts(),
                                            s.grid_origin, s.voxel_size, ref_c, ref_o)
        jobs.append(dict(s=s, t=t, ref_c=ref_c, ref_o=ref_o, n_rows=n_rows, C=C, err=None,
                         count=torch.zeros(n_rows, dtype=torch.int32, device=dev), out=torch.zeros(n_rows, C, device=dev),
                         vm=[t["vmi"].reshape(-1, 16)[a:b].reshape(-1).contiguous() for a, b in ((0, 2), (2, 6))]))
    torch.cuda.synchronize()
    start = threading.Barrier(2)

    def work(j):
        try:
            torch.cuda.set_device(dev)
            ws = voxproj_host.Workspace()
            stream = torch.cuda.Stream(dev)
            start.wait()
            with torch.cuda.stream(stream):
                for _ in range(4):
                    for k, (a, b) in enumerate(((0, 2), (2, 6))):
                        voxproj_host.project_features_raw(j["t"]["feats"][:, a:b], j["t"]["occ"], j["vm"][k], j["t"]["intr"], j["t"]["opts"],
                                                          j["count"], j["out"], j["t"]["origin"], j["s"].voxel_size, workspace=ws,
                                                          sync=False, pipeline=True)
                voxproj_host.workspace_status(ws, dev)
            stream.synchronize()
            ws.release()
        except Exception as e:      # surfaced by the main thread
            j["err"] = e

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    for j in jobs:
        assert j["err"] is None, j["err"]
        assert np.array_equal(j["count"].cpu().numpy(), j["ref_c"])
        assert j["out"].cpu().numpy().tobytes() == j["ref_o"].tobytes()
