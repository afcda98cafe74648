"""CPU suite: the oracle against the committed golden vectors.

  reference_python_goldens.npz  outputs of the REFERENCE's own debug_project_colors.py / the diagnostics its
                                debug_project_features.py prints, produced in the build container by
                                tests/golden/make_reference_goldens.py -> pins the RGB oracle and the DPF
                                diagnostics restatement to the real reference.
  s1_oracle_golden.npz          oracle outputs on fixture S1 (guards the oracle itself against drift; the
                                GPU suite compares the HIP path with the same file).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_rgb_oracle_matches_reference_debug_project_colors(oracle_mod):
    g = np.load(os.path.join(HERE, "golden", "reference_python_goldens.npz"))
    for v in range(3):
        colors, zyx, uv = oracle_mod.rgb_project(g["occ"], g["c2w"][v], g["intr"], g["grid_origin"],
                                                 float(g["voxel_size"]), g[f"img{v}"])
        assert np.array_equal(zyx, g[f"zyx{v}"])
        assert np.array_equal(uv, g[f"uv{v}"])
        assert colors.tobytes() == g[f"colors{v}"].tobytes()
        assert len(zyx) > 300


def test_dpf_diagnostics_match_reference_printout(oracle_mod):
    g = np.load(os.path.join(HERE, "golden", "reference_python_goldens.npz"))
    for v in range(3):
        d = oracle_mod.dpf_diagnostics(g["occ"], g["c2w"][v], g["intr"], g["grid_origin"], float(g["voxel_size"]), 64, 96)
        n_in, n_front, umin, umax, vmin, vmax = g[f"dpf{v}"]
        assert (d["n_in_bounds"], d["n_front"]) == (int(n_in), int(n_front))
        for got, ref in ((d["umin"], umin), (d["umax"], umax), (d["vmin"], vmin), (d["vmax"], vmax)):
            assert float(f"{got:.1f}") == ref        # the reference prints with one decimal


def test_host_mirror_diagnostics_match_reference_printout():
    import torch

    from debug_project_features import voxel_centre_diagnostics
    g = np.load(os.path.join(HERE, "golden", "reference_python_goldens.npz"))
    for v in range(3):
        d = voxel_centre_diagnostics(torch.from_numpy(g["occ"]), torch.from_numpy(g["c2w"][v]), torch.from_numpy(g["intr"]),
                                     torch.from_numpy(g["grid_origin"]), float(g["voxel_size"]), 96, 64)
        assert (d["n_in_bounds"], d["n_front"]) == (int(g[f"dpf{v}"][0]), int(g[f"dpf{v}"][1]))


def test_oracle_reproduces_s1_golden(oracle_mod):
    from make_oracle_goldens import S1, s1_inputs
    g = np.load(os.path.join(HERE, "golden", "s1_oracle_golden.npz"))
    s, feats = s1_inputs()
    assert int(s.occ.astype(np.int64).sum()) == int(g["occ_checksum"])
    assert feats.astype(np.float64).sum() == float(g["feats_checksum"])
    n_rows, C = s.n_vox + 1, S1["channels"]
    count = np.zeros(n_rows, np.int32)
    sums = np.zeros((n_rows, C), np.float32)
    r = oracle_mod.project_features(feats[None], s.occ[None].astype(np.int64), s.c2w.reshape(-1), s.intr[None], s.opts(),
                                    s.grid_origin, s.voxel_size, count, sums)
    assert np.array_equal(r["hits"][0], g["hits"]) and np.array_equal(count, g["count"])
    assert sums.tobytes() == g["sums"].tobytes()


def test_bso_mirror_matches_oracle_restatement(oracle_mod):
    import torch  # noqa: F401

    import build_sparse_occupancy as bso
    rng = np.random.default_rng(3)
    pts = (rng.uniform(-1, 1, size=(500, 3)) * np.array([2.0, 1.5, 1.0])).astype(np.float32)
    pts[10] = pts[3]                                  # duplicates: last one wins (Q12)
    for origin in ([-2.0, -1.5, -1.0], [0.3, -0.2, 0.1]):   # second origin makes some coords negative (Q11 shift)
        a = oracle_mod.build_occupancy(pts, origin, 0.25)
        b = bso.build_occupancy(pts, origin, 0.25).numpy()
        assert a.shape == b.shape and np.array_equal(a, b)


def test_ply_roundtrip(tmp_path):
    import build_sparse_occupancy as bso
    pts = np.array([[0.0, 0.0, 0.0], [0.04, 0.08, 0.12], [1.0, -2.0, 3.5]], np.float32)
    p = tmp_path / "grid_3vox_test.ply"
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment voxel_size 0.04\ncomment grid_origin -1.5 -2.25 0.125\n"
                "element vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
        for q in pts:
            f.write(f"{q[0]} {q[1]} {q[2]} 255 255 255\n")
    vs, origin, shape, n = bso.extract_voxel_params(str(p))
    assert (vs, origin, shape, n) == (0.04, [-1.5, -2.25, 0.125], None, 3)
    assert np.array_equal(bso.read_voxel_ply(str(p)), pts)
