"""Single-view projection driver (counterpart of the reference's debug_project_features.py).

The reference script (cuda_project_image_to_sparse_voxel/debug_project_features.py:17-258) loads
``tensor_data.pt``, prints diagnostics, calls ``project_features_cuda.project_features_cuda`` on the first
view and saves ``proj_output.pt``.  This module keeps that command line and the output file
(``projected_feats`` fp16 [n_hit,C] = per-view pixel SUMS, ``projected_indices`` int32 [n_hit,3] = (z,y,x)),
and exposes the steps as functions:

  build_id_to_zyx            reverse map voxel ID -> (z,y,x), -1 where an ID is not in the grid (DPF:35-45)
  voxel_centre_diagnostics   the per-voxel Python loop of DPF:59-84 (project every occupied voxel centre,
                             count those in front of the camera / inside the image) as one vectorised
                             float64 torch expression -- the "torch-CPU loop" of BASELINE config 1
  project_view               DPF:141-256: first view only (Q5), occ.long() with a batch axis, zero outputs
                             of max_id+1 rows (Q4), opts = [W,H,0.01,10.0,0.5*voxel_size] (Q8), the
                             extension call, hit rows -> (z,y,x) + fp16 sums

There is no CPU path for the projection itself: ``project_view`` needs a GPU.
"""
import argparse
import os

import torch

import project_features_cuda


def build_id_to_zyx(occ_zyx):
    """int64 [max_id+1,3] table of (z,y,x) per voxel ID, rows of absent IDs are -1 (DPF:35-45)."""
    occ = occ_zyx
    max_id = int(occ.max().item()) if occ.numel() else 0
    table = torch.full((max_id + 1, 3), -1, dtype=torch.long, device=occ.device)
    nz = occ.nonzero(as_tuple=False).long()
    if nz.numel() > 0:
        ids = occ[nz[:, 0], nz[:, 1], nz[:, 2]].long()
        table[ids] = nz
    return table


def voxel_centre_diagnostics(occ_zyx, c2w, intr4, grid_origin, voxel_size, img_w, img_h):
    """DPF:59-84 without the Python loop.  float64 math exactly as numpy promotes it there
    (float32 origin / pose / intrinsics promoted, python-float voxel_size).
    Returns dict(n_front, n_in_bounds, umin, umax, vmin, vmax)."""
    dev = occ_zyx.device
    zyx = (occ_zyx > 0).nonzero(as_tuple=False).to(torch.float64)
    world = grid_origin.to(dev, torch.float64)[None, :] + float(voxel_size) * zyx[:, [2, 1, 0]]
    m = c2w.to(dev).reshape(4, 4).to(torch.float64)
    d = world - m[:3, 3][None, :]
    cam = torch.stack([m[0, i] * d[:, 0] + m[1, i] * d[:, 1] + m[2, i] * d[:, 2] for i in range(3)], 1)   # R^T d
    front = cam[:, 2] > 0
    cam = cam[front]
    fx, fy, cx, cy = (intr4.reshape(-1)[i].to(dev, torch.float64) for i in range(4))
    u = fx * (cam[:, 0] / cam[:, 2]) + cx
    v = fy * (cam[:, 1] / cam[:, 2]) + cy
    inb = (u >= 0) & (u < img_w) & (v >= 0) & (v < img_h)
    res = dict(n_front=int(front.sum().item()), n_in_bounds=int(inb.sum().item()))
    if u.numel():
        res.update(umin=float(u.min()), umax=float(u.max()), vmin=float(v.min()), vmax=float(v.max()))
    return res


def project_view(feats, occ_zyx, intr, extr, grid_origin, voxel_size, device="cuda", id_to_zyx=None):
    """DPF:141-256 for one tensor_data record.

    feats f32 [1,V,H,W,C] (only view 0 is used, DPF:145), occ int [Z,Y,X], intr f32 [1,*,4], extr f32
    [1,V,4,4].  Returns dict(projected_feats f16 [n,C], projected_indices i32 [n,3], count i32 [max_id+1],
    sums f32 [max_id+1,C]) -- the first two are what the reference saves.
    """
    dev = torch.device(device)
    if id_to_zyx is None:
        id_to_zyx = build_id_to_zyx(occ_zyx.cpu())
    feats = feats[:, 0:1, ...].to(dev).contiguous()                       # DPF:142,145
    occ = occ_zyx.unsqueeze(0).to(dev).contiguous().long()                # DPF:143
    intr0 = intr[:, 0, :].to(dev).contiguous()                            # DPF:146
    extr0 = extr[:, 0, :, :].contiguous().view(-1).to(dev)                # DPF:147
    num_ids = int(occ.max().item()) + 1                                   # DPF:158-159
    _, _, H, W, C = feats.shape
    mapping2dto3d = torch.zeros((num_ids,), dtype=torch.int32, device=dev)
    proj_feats = torch.zeros((num_ids, C), dtype=torch.float32, device=dev)
    opts = torch.tensor([W, H, 0.01, 10.0, voxel_size * 0.5], dtype=torch.float32, device="cpu")   # DPF:167-169
    pred_mode = torch.tensor([False], dtype=torch.bool, device="cpu")
    grid_origin_cpu = grid_origin.detach().cpu().contiguous().to(torch.float32).view(-1)          # DPF:199
    project_features_cuda.project_features_cuda(feats, occ, extr0, intr0, opts, mapping2dto3d, proj_feats,
                                                pred_mode, grid_origin_cpu, float(voxel_size))
    nonzero = (mapping2dto3d > 0).nonzero(as_tuple=True)[0]              # DPF:237
    idx = id_to_zyx.to(nonzero.device)[nonzero].int().cpu()               # DPF:240
    pf = proj_feats[nonzero].cpu()                                        # DPF:243
    valid = idx[:, 0] != -1                                               # DPF:246-248
    return dict(projected_feats=pf[valid].to(torch.float16), projected_indices=idx[valid],   # DPF:252-256
                count=mapping2dto3d, sums=proj_feats)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--tensor_data", required=True)
    parser.add_argument("--output", default="proj_output.pt")
    args = parser.parse_args(argv)
    assert os.path.isfile(args.tensor_data), f"tensor_data not found: {args.tensor_data}"
    data = torch.load(args.tensor_data, map_location="cpu")
    feats, occ = data["encoded_2d_features"], data["occupancy_3D"]
    intr, extr = data["intrinsicParams"], data["viewMatrixInv"]
    grid_origin, voxel_size = data["grid_origin"], data["voxel_size"]
    _, _, H, W, _ = feats.shape
    d = voxel_centre_diagnostics(occ, extr[0, 0], intr[0, 0], grid_origin, voxel_size, W, H)
    if d["n_front"]:
        print(f"u: min={d['umin']:.1f}, max={d['umax']:.1f}")
        print(f"v: min={d['vmin']:.1f}, max={d['vmax']:.1f}")
        print(f"Number of projected voxels in bounds: {d['n_in_bounds']} / {d['n_front']}")
    else:
        print("No voxels projected in front of the camera.")
    out = project_view(feats, occ, intr, extr, grid_origin, voxel_size)
    torch.save({"projected_feats": out["projected_feats"], "projected_indices": out["projected_indices"]}, args.output)
    print(f"Saved filtered and compressed projection output to {args.output}")


if __name__ == "__main__":
    main()
