"""Generates tests/golden/reference_python_goldens.npz by RUNNING the reference's own Python in the build
container (it cannot travel to the GPU box, the vectors can).

  * debug_project_colors.py (RGB path, pure CPU once its `import project_features_cuda` resolves -- this
    repo's drop-in module is put on PYTHONPATH for that, nothing in it is called): full outputs
    projected_colors / projected_indices / pixel_indices  -> pins oracle_rgb_project and k_project_colors.
  * debug_project_features.py: its CPU diagnostics loop (lines 59-84) prints the in-bounds count and the u/v
    range before the script reaches `.cuda()` (which fails here: no GPU).  The printed numbers pin
    oracle_dpf_diagnostics / voxel_centre_diagnostics.

Usage (build container only):  python tests/golden/make_reference_goldens.py
"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(ROOT, "3d-semantic-segmentation_amd")
REF = "/root/reference/cuda_project_image_to_sparse_voxel"
sys.path.insert(0, PKG)
from synthetic_scene import make_scene  # noqa: E402


def main():
    s = make_scene(3000, 3, 96, 64, seed=7, room=(5.0, 4.0, 2.4))
    rng = np.random.default_rng(7)
    out = {}
    env = dict(os.environ, PYTHONPATH=PKG, TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD="1")
    with tempfile.TemporaryDirectory() as tmp:
        for v in range(s.n_views):
            img = rng.integers(0, 256, size=(64, 96, 3), dtype=np.uint8)
            data = {
                "encoded_2d_features": torch.zeros(1, 1, 64, 96, 4),
                "occupancy_3D": torch.from_numpy(s.occ),
                "intrinsicParams": torch.from_numpy(s.intr)[None, None],
                "viewMatrixInv": torch.from_numpy(s.c2w[v])[None, None],
                "grid_origin": torch.from_numpy(s.grid_origin),
                "voxel_size": float(s.voxel_size),
                "image": img,
            }
            td, po = os.path.join(tmp, f"td{v}.pt"), os.path.join(tmp, f"po{v}.pt")
            torch.save(data, td)
            subprocess.run([sys.executable, os.path.join(REF, "debug_project_colors.py"), "--tensor_data", td,
                            "--output", po], check=True, env=env, cwd=tmp, stdout=subprocess.DEVNULL)
            r = torch.load(po, weights_only=False)
            out[f"img{v}"] = img
            out[f"colors{v}"] = r["projected_colors"].numpy()
            out[f"zyx{v}"] = r["projected_indices"].numpy()
            out[f"uv{v}"] = r["pixel_indices"].numpy()
            # DPF diagnostics (the script then dies at .cuda(): expected, no GPU here)
            p = subprocess.run([sys.executable, os.path.join(REF, "debug_project_features.py"), "--tensor_data", td,
                                "--output", po], env=env, cwd=tmp, capture_output=True, text=True)
            m = re.search(r"Number of projected voxels in bounds: (\d+) / (\d+)", p.stdout)
            mu = re.search(r"u: min=(-?[\d.]+), max=(-?[\d.]+)", p.stdout)
            mv = re.search(r"v: min=(-?[\d.]+), max=(-?[\d.]+)", p.stdout)
            assert m and mu and mv, p.stdout[-2000:] + p.stderr[-2000:]
            out[f"dpf{v}"] = np.array([int(m.group(1)), int(m.group(2)), float(mu.group(1)), float(mu.group(2)),
                                       float(mv.group(1)), float(mv.group(2))])
    out.update(occ=s.occ, c2w=s.c2w, intr=s.intr, grid_origin=s.grid_origin, voxel_size=np.float64(s.voxel_size))
    np.savez_compressed(os.path.join(HERE, "reference_python_goldens.npz"), **out)
    print("wrote", os.path.join(HERE, "reference_python_goldens.npz"),
          {k: v.shape for k, v in out.items() if k.startswith(("colors", "dpf"))})


if __name__ == "__main__":
    main()
