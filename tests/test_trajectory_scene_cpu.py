"""The second generator mode of synthetic_scene (round 5): a hand-held trajectory through a cluttered room with a missing wall
segment -- the properties the bench legs A1 / R2T and tests/test_gpu_trajectory.py rely on, checked on CPU with the oracle's
march at a quarter of the resolution."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_module_traj", os.path.join(ROOT, "bench.py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


def _steps(c2w):
    P, F = c2w[:, :3, 3].astype(np.float64), c2w[:, :3, 2].astype(np.float64)
    d = np.linalg.norm(np.diff(P, axis=0), axis=1)
    ang = np.degrees(np.arccos(np.clip((F[1:] * F[:-1]).sum(1), -1.0, 1.0)))
    return d, ang


def test_trajectory_views_are_a_smooth_path_and_the_scene_is_exact(oracle_mod):
    bm = _bench_module()
    for name, cell in (("A1", 0.04), ("R2T", None)):
        n_vox, n_views, W, H, C = bm.WORKLOADS[name]
        s = bm.workload_scene(name, W=W // 4, H=H // 4)
        assert s.n_vox == n_vox == int((s.occ > 0).sum()) and s.n_views == n_views
        assert np.array_equal(np.sort(s.occ[s.occ > 0]), np.arange(1, n_vox + 1))
        if cell is not None:
            assert abs(s.voxel_size - cell) < 1e-6                    # AGG:28 (voxel_size 0.04)
        d, ang = _steps(s.c2w)
        assert d.max() <= 0.05 and ang.max() <= 3.0, (d.max(), ang.max())
        assert d[:55].max() < 0.012 and ang[:55].max() < 1.0          # the first shot is a dwell
        # rotation blocks are orthonormal, right-handed
        R = s.c2w[:, :3, :3].astype(np.float64)
        assert np.abs(R @ R.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5 and (np.linalg.det(R) > 0.999).all()
        s2 = bm.workload_scene(name, W=W // 4, H=H // 4)
        assert np.array_equal(s.occ, s2.occ) and np.array_equal(s.c2w, s2.c2w)
        # what the rays see, at a quarter of the resolution
        hits = oracle_mod.first_hit(s.occ[None].astype(np.int64), s.c2w.reshape(-1), s.intr[None], s.opts(), s.grid_origin,
                                    s.voxel_size, 1, n_views)
        miss = float((hits == 0).mean())
        assert 0.2 <= miss <= 0.45, (name, miss)                     # 20-40 % of the rays miss
        chunk, n_calls, _ = bm.plan_calls(n_views, H, W, C, 4)
        first = np.bincount(hits[0, :chunk].reshape(-1), minlength=n_vox + 1)[1:]
        assert first.max() * 16 >= 100000, first.max() * 16           # a voxel with >= 10^5 full-resolution pixels in the first call
        heavy_t = min(256 + 64 * chunk, 2048) / 16.0
        assert first[first > heavy_t].sum() > 0.9 * first.sum()       # the close-up call: nearly every pixel in a heavy voxel
        # clutter: about half of the occupied cells are not on the room's shell
        nz, ny, nx = s.occ.shape
        shell = np.zeros_like(s.occ, bool)
        shell[0], shell[-1], shell[:, 0], shell[:, -1], shell[:, :, 0], shell[:, :, -1] = True, True, True, True, True, True
        frac = float(((s.occ > 0) & ~shell).sum()) / n_vox
        assert 0.4 < frac < 0.6, frac


def test_benign_room_is_what_it_was():
    """The first generator mode is untouched by the second (fixtures and the headline bench depend on it)."""
    from synthetic_scene import make_scene
    s = make_scene(10000, 8, 64, 64, seed=0)
    assert s.occ.shape == (18, 44, 56) and abs(s.voxel_size - 0.17993463575839996) < 1e-12
    assert np.allclose(s.c2w[3, :3, 3], [-2.1131508, 1.3627753, 1.48], atol=1e-6)
    assert int(s.occ.astype(np.int64).sum()) == 10000 * 10001 // 2
