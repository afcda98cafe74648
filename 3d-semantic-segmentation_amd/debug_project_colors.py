"""RGB projection of one view (counterpart of the reference's debug_project_colors.py).

The reference (cuda_project_image_to_sparse_voxel/debug_project_colors.py:16-89) imports the CUDA extension
but never calls it: the colour path is a per-voxel Python/numpy loop -- voxel-driven, nearest pixel, NO
occlusion test, float64 math (DPC:54-81).  ``project_colors_view`` runs that loop as the HIP kernel
k_project_colors (one lane per occupied voxel) and returns the reference's three tensors in the reference's row
order ((z,y,x) raster order of np.nonzero, DPC:50,58).
"""
import argparse
import os

import numpy as np
import torch

import project_features_cuda  # noqa: F401  (the reference imports it too, DPC:13)
import voxproj_host


def project_colors_view(occ_zyx, extr, intr4, grid_origin, voxel_size, img_np, device="cuda"):
    """Returns dict(projected_colors f32 [n,3], projected_indices i32 [n,3] (z,y,x), pixel_indices i32 [n,2] (u,v))."""
    dev = torch.device(device)
    occ = occ_zyx.to(dev, torch.int32).contiguous()
    n_rows = int(occ.max().item()) + 1
    img = torch.from_numpy(np.ascontiguousarray(img_np, dtype=np.uint8))[None].to(dev).contiguous()
    c2w = extr.reshape(1, 4, 4).to(dev, torch.float32).contiguous()
    intr = intr4.reshape(1, 4).to(dev, torch.float32).contiguous()
    csum = torch.zeros(n_rows, 3, dtype=torch.float32, device=dev)
    hits = torch.zeros(n_rows, dtype=torch.int32, device=dev)
    uv = torch.empty(1, n_rows, 2, dtype=torch.int32, device=dev)
    voxproj_host.project_colors_raw(occ, c2w, intr, [float(v) for v in grid_origin], float(voxel_size), img, csum, hits,
                                    pixel_uv=uv)
    # reference row order: raster (z,y,x) over occupied cells (DPC:50,58)
    zyx = (occ > 0).nonzero(as_tuple=False)
    ids = occ[zyx[:, 0], zyx[:, 1], zyx[:, 2]].long()
    keep = hits[ids] > 0
    zyx, ids = zyx[keep], ids[keep]
    # one contribution per voxel in a single-view call; (u,v) as sampled by the kernel (DPC:76)
    return dict(projected_colors=csum[ids].cpu(), projected_indices=zyx.int().cpu(), pixel_indices=uv[0, ids].cpu())


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--tensor_data", required=True)
    parser.add_argument("--output", default="proj_output.pt")
    args = parser.parse_args(argv)
    assert os.path.isfile(args.tensor_data), f"tensor_data not found: {args.tensor_data}"
    data = torch.load(args.tensor_data, map_location="cpu", weights_only=False)
    if "image" not in data:
        raise RuntimeError("Image array not found in tensor_data.pt. Please include the image for color projection.")
    out = project_colors_view(data["occupancy_3D"], data["viewMatrixInv"][0, 0], data["intrinsicParams"][0, 0],
                              data["grid_origin"], data["voxel_size"], data["image"])
    torch.save(out, args.output)
    print(f"Saved color projection output to {args.output}")


if __name__ == "__main__":
    main()
