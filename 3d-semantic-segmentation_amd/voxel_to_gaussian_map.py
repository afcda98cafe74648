"""Gaussian -> nearest-voxel map on the GPU (the step right after the projector).

Counterpart of ``map_gaussians_to_voxels`` in the reference's voxel_to_gaussian/voxeltoGaussian_logits.py:87-105
(identical code in voxel_to_gaussian/voxeltoGaussian.py:84-93): for each Gaussian centre the index of its 1-NN
voxel, found there with ``sklearn.neighbors.KDTree(voxel_pos, leaf_size=16).query(k=1)`` in batches on the CPU.
Same signature and return type here (``(M,) int64`` CPU tensor); the search itself is the HIP kernel
k_nearest_voxel (exact, float64 distances like sklearn; on exact distance ties the lowest voxel index wins,
sklearn's choice there is unspecified).  PyTorch only buckets the voxel positions on a uniform grid
(sort by cell, prefix sums): plumbing, not the search.
"""
import ctypes

import torch

import voxproj_host


def _bucket(voxel_pos, device):
    pos = voxel_pos.to(device, torch.float32).contiguous()
    N = pos.shape[0]
    p64 = pos.double()
    lo, hi = p64.min(0).values, p64.max(0).values
    ext = (hi - lo).clamp_min(1e-9)
    # cell size ~ mean spacing of a volume-filling set, never more than ~8M cells
    h = float((ext.prod() / max(N, 1)) ** (1.0 / 3.0))
    h = max(h, float(ext.max()) / 200.0, 1e-9)
    dims = [int(v) for v in (torch.floor(ext / h).long() + 1).tolist()]
    cell = torch.floor((p64 - lo) / h).long()
    for a in range(3):
        cell[:, a].clamp_(0, dims[a] - 1)
    lin = (cell[:, 2] * dims[1] + cell[:, 1]) * dims[0] + cell[:, 0]
    order = torch.argsort(lin, stable=True)
    ncell = dims[0] * dims[1] * dims[2]
    counts = torch.bincount(lin, minlength=ncell)
    start = torch.zeros(ncell + 1, dtype=torch.int32, device=device)
    start[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return pos[order].contiguous(), order.to(torch.int32).contiguous(), start, [float(v) for v in lo.tolist()], h, dims


@torch.inference_mode()
def map_gaussians_to_voxels(voxel_pos, gaussian_mu, batch_size=200_000, device="cuda"):
    """For each Gaussian centre, index of its 1-NN voxel.  Returns (M,) int64 tensor on the CPU.

    ``batch_size`` is accepted for signature compatibility (the GPU search needs no batching).
    """
    dev = torch.device(device)
    if voxel_pos.shape[0] == 0:
        raise ValueError("voxel_pos is empty")
    pts, perm, start, origin, h, dims = _bucket(voxel_pos, dev)
    q = gaussian_mu.to(dev, torch.float32).contiguous()
    M = int(q.shape[0])
    out = torch.empty(M, dtype=torch.int64, device=dev)
    g = (ctypes.c_double * 3)(*origin)
    with torch.cuda.device(dev):
        voxproj_host.check(voxproj_host.lib().vp_nearest_voxel(
            pts.data_ptr(), perm.data_ptr(), start.data_ptr(), g, ctypes.c_double(h), dims[0], dims[1], dims[2],
            q.data_ptr(), M, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    return out.cpu()
