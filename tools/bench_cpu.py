"""The CPU baselines of bench.py's `cpu_baseline` legs, timed on the GPU box's host cores on a bounded sample:
the build's oracle (a port of the reference kernel -- test infrastructure, used here as the thing TIMED beside the GPU path,
never as part of it) and vectorised torch-CPU restatements of the reference's two Python projection loops.  Imported by bench.py
only, after its timed region."""
import os
import time

import numpy as np
import torch

def host_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota (a one-GPU box of
    the pool shows 256 logical CPUs but grants 16; 256 OpenMP threads on that share run the oracle 7x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


_cpu_maps = {}


def cpu_baseline(scene, C, n_views, n_threads, maps=4):
    """Time the CPU oracle (port of project_image_cuda_kernel.cu:24-92,157-187) on n_views views of the workload, `maps` at a time
    (every chunk has its own poses and reads the same `maps` synthetic feature maps: generating 1 GB maps on the host costs
    more than marching them, and the oracle's work does not depend on their values)."""
    from oracle import oracle
    from synthetic_scene import make_features_np
    key = (scene.height, scene.width, C, maps)
    if key not in _cpu_maps:
        _cpu_maps.clear()
        _cpu_maps[key] = make_features_np(maps, scene.height, scene.width, C, seed=0)[None]
    feats = _cpu_maps[key]
    n_rows = scene.n_vox + 1
    count = np.zeros(n_rows, np.int32)
    out = np.zeros((n_rows, C), np.float32)
    occ = scene.occ[None].astype(np.int64)
    dt = 0.0
    for a in range(0, n_views, maps):
        b = min(n_views, a + maps)
        t0 = time.perf_counter()
        oracle.project_features(feats[:, :b - a], occ, scene.c2w[a:b].reshape(-1), scene.intr[None], scene.opts(),
                                scene.grid_origin, scene.voxel_size, count, out, want_hits=False, nthreads=n_threads)
        dt += time.perf_counter() - t0
    return dict(value=round(scene.n_vox * n_views / dt / 1e6, 4), unit="Mvoxel-views/s", cores=n_threads, kind="port",
                sample=f"{n_views} of the workload's views at full resolution, all {scene.n_vox} voxels, "
                       f"{dt:.1f} s wall (OpenMP over pixel rows + channel slices; {os.cpu_count()} logical CPUs "
                       f"visible, {n_threads} granted to this process)")


def colour_projection_torch_cpu(occ_zyx, c2w, intr4, grid_origin, voxel_size, img):
    """The reference's colour loop (debug_project_colors.py:58-73: every occupied voxel centre through the pinhole model in
    numpy float64, in-front test, banker's rounding to the nearest pixel, image-bounds test, img[v, u] / 255) as one vectorised
    torch-CPU expression.  Returns (colors f32 [n,3], zyx i64 [n,3], uv i64 [n,2]) in the loop's raster order."""
    zyx = (occ_zyx > 0).nonzero(as_tuple=False)
    world = grid_origin.to(torch.float64)[None, :] + float(voxel_size) * zyx[:, [2, 1, 0]].to(torch.float64)      # DPC:60
    m = c2w.reshape(4, 4).to(torch.float64)
    d = world - m[:3, 3][None, :]
    cam = torch.stack([m[0, i] * d[:, 0] + m[1, i] * d[:, 1] + m[2, i] * d[:, 2] for i in range(3)], 1)            # R^T d, DPC:61-63
    fx, fy, cx, cy = (intr4.reshape(-1)[i].to(torch.float64) for i in range(4))
    front = cam[:, 2] > 0                                                                                         # DPC:65
    z = torch.where(front, cam[:, 2], torch.ones_like(cam[:, 2]))
    u = torch.round(fx * (cam[:, 0] / z) + cx)                                                                    # DPC:66-68 (half to even)
    v = torch.round(fy * (cam[:, 1] / z) + cy)
    ok = front & (u >= 0) & (u < img.shape[1]) & (v >= 0) & (v < img.shape[0])                                    # DPC:69
    ui, vi = u[ok].long(), v[ok].long()
    colors = (img[vi, ui].to(torch.float64) / 255.0).to(torch.float32)                                            # DPC:70,75
    return colors, zyx[ok], torch.stack([ui, vi], 1)


def distinct_image_lines(scene, views, H, W, dev):
    """Distinct 64-byte lines of the [V,H,W,3] uint8 images that hold a pixel some voxel samples in `views` -- the compulsory
    image traffic of one vp_project_colors call (H*W*3 is a multiple of 64 for config 5, so lines never span two images)."""
    occ = torch.from_numpy(scene.occ).to(dev)
    c2w = torch.from_numpy(scene.c2w).to(dev)
    intr = torch.from_numpy(scene.intr).to(dev)
    origin = torch.from_numpy(np.asarray(scene.grid_origin, dtype=np.float32)).to(dev)
    blank = torch.zeros(H, W, 3, dtype=torch.uint8, device=dev)
    n = 0
    for v in views:
        uv = colour_projection_torch_cpu(occ, c2w[v], intr, origin, scene.voxel_size, blank)[2]
        off = (uv[:, 1] * W + uv[:, 0]) * 3
        n += int(torch.unique(torch.cat([off // 64, (off + 2) // 64])).numel())
    return n


def cpu_colour_loop(scene, img_u8, n_views, n_threads):
    """cpu_torch_loop of the R4 leg: colour_projection_torch_cpu over n_views views on the box's host cores."""
    torch.set_num_threads(n_threads)
    occ = torch.from_numpy(scene.occ)
    c2w = torch.from_numpy(scene.c2w)
    intr = torch.from_numpy(scene.intr)
    origin = torch.from_numpy(np.asarray(scene.grid_origin, dtype=np.float32))
    img = torch.from_numpy(img_u8)
    colour_projection_torch_cpu(occ, c2w[0], intr, origin, scene.voxel_size, img)        # warm
    t0 = time.perf_counter()
    seen = 0
    for v in range(n_views):
        seen += colour_projection_torch_cpu(occ, c2w[v], intr, origin, scene.voxel_size, img)[0].shape[0]
    dt = time.perf_counter() - t0
    return dict(value=round(scene.n_vox * n_views / dt / 1e6, 3), unit="Mvoxel-views/s", cores=n_threads,
                what="the reference's colour loop (debug_project_colors.py:58-73) as one vectorised float64 torch-CPU expression per view",
                sample=f"{n_views} views x {scene.n_vox} voxels, {dt:.2f} s wall, {seen} voxel-views in the image")


def cpu_torch_loop(scene, n_views, n_threads):
    """The reference's only CPU projection loop (debug_project_features.py:59-84: every occupied voxel centre
    through the pinhole model, in-front and in-image tests) as the vectorised torch-CPU expression of
    debug_project_features.voxel_centre_diagnostics -- no occlusion test, no feature gather, so it is NOT the
    same work as the projector; reported beside cpu_baseline because north_star names it."""
    from debug_project_features import voxel_centre_diagnostics
    torch.set_num_threads(n_threads)
    occ = torch.from_numpy(scene.occ)
    c2w = torch.from_numpy(scene.c2w)
    intr = torch.from_numpy(scene.intr)
    origin = torch.from_numpy(np.asarray(scene.grid_origin, dtype=np.float32))
    voxel_centre_diagnostics(occ, c2w[0], intr, origin, scene.voxel_size, scene.width, scene.height)   # warm
    t0 = time.perf_counter()
    inb = 0
    for v in range(n_views):
        inb += voxel_centre_diagnostics(occ, c2w[v], intr, origin, scene.voxel_size, scene.width, scene.height)["n_in_bounds"]
    dt = time.perf_counter() - t0
    return dict(value=round(scene.n_vox * n_views / dt / 1e6, 3), unit="Mvoxel-views/s", cores=n_threads,
                what="voxel-centre projection + bounds test only (DPF:59-84), float64 torch-CPU, vectorised",
                sample=f"{n_views} views x {scene.n_vox} voxels, {dt:.2f} s wall, {inb} centres in bounds")
