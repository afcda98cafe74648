"""Stage-5 front end: Gaussian -> nearest-voxel map, GPU kernel vs the reference's method
(sklearn KDTree(leaf_size=16).query(k=1), voxel_to_gaussian/voxeltoGaussian_logits.py:87-105) on the host cores."""
import json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "3d-semantic-segmentation_amd"))
import numpy as np, torch
from synthetic_scene import make_scene
from voxel_to_gaussian_map import map_gaussians_to_voxels
N, M = 200000, 2000000
s = make_scene(N, 1, 8, 8, seed=0)
rng = np.random.default_rng(0)
vox = torch.from_numpy(s.points)
mu = torch.from_numpy((s.points[rng.integers(0, N, M)] + rng.normal(0, 0.05, (M, 3))).astype(np.float32))
map_gaussians_to_voxels(vox, mu[:1000])
torch.cuda.synchronize()
t0 = time.perf_counter(); got = map_gaussians_to_voxels(vox, mu); dt = time.perf_counter() - t0
from sklearn.neighbors import KDTree
Ms = 200000
t0 = time.perf_counter(); tree = KDTree(vox.numpy(), leaf_size=16); ref = tree.query(mu[:Ms].numpy(), k=1, return_distance=False)[:, 0]
dtc = time.perf_counter() - t0
same = (got[:Ms].numpy() == ref).mean()
print(json.dumps({"workload": f"{N} voxels, {M} Gaussian centres", "gpu_ms_incl_bucketing_and_copies": round(dt * 1e3, 1),
                  "gpu_Mqueries_per_s": round(M / dt / 1e6, 1), "cpu_kdtree_Mqueries_per_s_1thread": round(Ms / dtc / 1e6, 3),
                  "index_agreement_on_sample": float(same)}))
