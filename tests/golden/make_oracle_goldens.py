"""Generates tests/golden/s1_oracle_golden.npz: fixture S1 (8 views x 2000 voxels x 48x32x16)
run through the CPU oracle at kernel level (first-hit IDs, counts, fp32 sums) and at aggregator level
(the reference's fp16 per-view round trip + per-view counting, oracle.aggregate_views).  Inputs are not
stored: they are regenerated from the seed by synthetic_scene (pure numpy, deterministic).

Usage:  python tests/golden/make_oracle_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-semantic-segmentation_amd"))
from oracle import oracle  # noqa: E402
from synthetic_scene import make_features_np, make_scene  # noqa: E402

S1 = dict(n_vox=2000, n_views=8, width=48, height=32, channels=16, seed=101, room=(5.0, 4.0, 2.4))


def s1_inputs():
    s = make_scene(S1["n_vox"], S1["n_views"], S1["width"], S1["height"], seed=S1["seed"], room=S1["room"])
    feats = make_features_np(S1["n_views"], S1["height"], S1["width"], S1["channels"], seed=S1["seed"])
    return s, feats


def main():
    s, feats = s1_inputs()
    n_rows, C = s.n_vox + 1, S1["channels"]
    count = np.zeros(n_rows, np.int32)
    sums = np.zeros((n_rows, C), np.float32)
    r = oracle.project_features(feats[None], s.occ[None].astype(np.int64), s.c2w.reshape(-1), s.intr[None], s.opts(),
                                s.grid_origin, s.voxel_size, count, sums)
    per_view = []
    for v in range(s.n_views):
        c1 = np.zeros(n_rows, np.int32)
        s1 = np.zeros((n_rows, C), np.float32)
        oracle.project_features(feats[None, v:v + 1], s.occ[None].astype(np.int64), s.c2w[v].reshape(-1), s.intr[None],
                                s.opts(), s.grid_origin, s.voxel_size, c1, s1)
        per_view.append(oracle.dpf_select_outputs(s.occ, c1, s1))
    agg = oracle.aggregate_views(per_view, s.grid_origin.astype(np.float64), s.voxel_size)
    np.savez_compressed(os.path.join(HERE, "s1_oracle_golden.npz"), hits=r["hits"][0], count=count, sums=sums,
                        agg_xyz=agg["xyz"], agg_avg=agg["avg_feats"], agg_coords=agg["voxel_coords"],
                        agg_hits=agg["hit_count"], occ_checksum=np.int64(s.occ.astype(np.int64).sum()),
                        feats_checksum=np.float64(feats.astype(np.float64).sum()))
    print("wrote s1_oracle_golden.npz", r["hits"].shape, int(count.sum()), agg["avg_feats"].shape)


if __name__ == "__main__":
    main()
