"""Generates tests/golden/nearest_voxel_golden.npz by calling the REFERENCE's own
voxel_to_gaussian/voxeltoGaussian_logits.py::map_gaussians_to_voxels (sklearn KDTree, k = 1) in the build
container on seeded synthetic voxel positions and Gaussian centres.  Usage: python tests/golden/make_stage5_golden.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "3d-semantic-segmentation_amd"))
from synthetic_scene import make_scene  # noqa: E402


def inputs():
    s = make_scene(6000, 1, 8, 8, seed=5, room=(5.0, 4.0, 2.4))
    rng = np.random.default_rng(5)
    vox = s.points[rng.permutation(s.n_vox)[:4000]].astype(np.float32)          # aggregated voxels: a subset, any order
    near = vox[rng.integers(0, len(vox), 20000)] + rng.normal(0, 0.03, (20000, 3)).astype(np.float32)
    room = rng.uniform([-2.6, -2.1, -0.1], [2.6, 2.1, 2.5], (4000, 3)).astype(np.float32)   # interior, far from voxels
    far = rng.uniform(-30, 30, (500, 3)).astype(np.float32)                      # far outside the bounding box
    exact = vox[:300].copy()                                                     # centres sitting on voxels
    mu = np.concatenate([near, room, far, exact]).astype(np.float32)
    return vox, mu


def main():
    sys.path.insert(0, "/root/reference/voxel_to_gaussian")
    from voxeltoGaussian_logits import map_gaussians_to_voxels      # the reference function (build container only)
    vox, mu = inputs()
    idx = map_gaussians_to_voxels(torch.from_numpy(vox), torch.from_numpy(mu), batch_size=7000).numpy()
    np.savez_compressed(os.path.join(HERE, "nearest_voxel_golden.npz"), idx=idx.astype(np.int64),
                        vox_checksum=np.float64(vox.astype(np.float64).sum()), mu_checksum=np.float64(mu.astype(np.float64).sum()))
    print("wrote nearest_voxel_golden.npz", idx.shape, idx[:5])


if __name__ == "__main__":
    main()
