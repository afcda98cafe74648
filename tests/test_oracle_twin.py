"""The C oracle and the independently written numpy twin must agree bit for bit."""
import numpy as np

from oracle import numpy_twin
from synthetic_scene import make_features_np, make_scene


def test_ray_setup_bitwise(oracle_mod):
    s = make_scene(2000, 3, 48, 32, seed=3, room=(5.0, 4.0, 2.4))
    for v in range(s.n_views):
        (cdx, cdy, cdz), pos, (wdx, wdy, wdz) = numpy_twin.rays(s.c2w[v], s.intr, 0.01, 10.0, 48, 32)
        for (x, y) in [(0, 0), (47, 31), (13, 7), (24, 16), (0, 31)]:
            r = oracle_mod.ray(s.c2w[v], s.intr, 0.01, 10.0, x, y)
            tw = np.array([cdx[y, x], cdy[y, x], cdz[y, x], pos[0], pos[1], pos[2],
                           wdx[y, x], wdy[y, x], wdz[y, x]], np.float32)
            assert r.tobytes() == tw.tobytes(), (v, x, y, r, tw)


def test_first_hit_and_sums_bitwise(oracle_mod):
    s = make_scene(2000, 3, 48, 32, seed=4, room=(5.0, 4.0, 2.4))
    C = 16
    feats = make_features_np(s.n_views, 32, 48, C, seed=4)
    N = s.n_vox
    count = np.zeros(N + 1, np.int32)
    out = np.zeros((N + 1, C), np.float32)
    r = oracle_mod.project_features(feats[None], s.occ[None].astype(np.int64), s.c2w.reshape(-1), s.intr[None],
                                    s.opts(), s.grid_origin, s.voxel_size, count, out)
    assert r["rc"] == 0
    tcount = np.zeros_like(count)
    tout = np.zeros_like(out)
    for v in range(s.n_views):
        hit = numpy_twin.first_hit(s.occ, s.c2w[v], s.intr, s.opts(), s.grid_origin, s.voxel_size)
        assert np.array_equal(hit, r["hits"][0, v]), v
        numpy_twin.accumulate(hit, feats[v], tcount, tout)
    assert (r["hits"] > 0).mean() > 0.9
    assert np.array_equal(count, tcount)
    assert out.tobytes() == tout.tobytes()


def test_first_hit_with_holes_and_tilted_grid(oracle_mod):
    # sparse random occupancy (rays leave the grid, long marches, misses) and a non-trivial origin
    rng = np.random.default_rng(5)
    occ = np.zeros((12, 20, 24), np.int32)
    idx = rng.choice(occ.size, 300, replace=False)
    occ.reshape(-1)[idx] = np.arange(1, 301)
    s = make_scene(2000, 2, 40, 24, seed=6, room=(5.0, 4.0, 2.4))
    origin = np.array([-2.1, -1.7, 0.05], np.float32)
    opts = np.array([40, 24, 0.01, 10.0, 0.11], np.float32)
    hits = oracle_mod.first_hit(occ[None].astype(np.int64), s.c2w.reshape(-1), s.intr[None], opts, origin, 0.2, 1, 2)
    for v in range(2):
        tw = numpy_twin.first_hit(occ, s.c2w[v], s.intr, opts, origin, 0.2)
        assert np.array_equal(tw, hits[0, v])
    assert 0.02 < (hits > 0).mean() < 0.98
