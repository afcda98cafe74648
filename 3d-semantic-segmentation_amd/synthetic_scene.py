"""Synthetic scenes for the projector's parity tests and bench.

There is no network for ScanNet++ data, so every measured configuration uses this generator:

  * voxels  -- a hollow room (default interior 10.0 x 8.0 x 3.2 m, one-voxel-thick shell) plus random
               solid boxes standing on the floor, surface cells only, trimmed to exactly ``n_vox``
               occupied cells; IDs 1..N in lexicographic (x,y,z) order; dense int32 grid [Z,Y,X]
               exactly as build_sparse_occupancy.py would emit it (reference BSO:44-46);
  * cameras -- poses on an ellipse at 1.5 m height looking along the tangent (+- jitter), row-major
               camera->world matrices (the reference's viewMatrixInv, PTD:165-172), intrinsics of the
               ScanNet++ DSLR (camera_params/colmap_camera_params.sh:7-14) scaled to the image width;
  * features-- i.i.d. N(0,1), L2-normalised over C (LSeg-like unit vectors), one seed per view.

Ray options follow debug_project_features.py:167-169: dmin 0.01, dmax 10.0, step 0.5 * voxel_size.
"""
import math
from dataclasses import dataclass

import numpy as np

DSLR = dict(w=1752, h=1168, fx=623.966, fy=624.818, cx=876.0, cy=584.0)


@dataclass
class Scene:
    occ: np.ndarray            # int32 [Z,Y,X], 0 = empty, else 1-based ID
    points: np.ndarray         # float32 [N,3] world centres, row i <-> ID i+1
    grid_origin: np.ndarray    # float32 [3]
    voxel_size: float
    c2w: np.ndarray            # float32 [V,4,4] row-major camera->world
    intr: np.ndarray           # float32 [4] fx, fy, cx, cy
    width: int
    height: int

    @property
    def n_vox(self):
        return int(self.points.shape[0])

    @property
    def n_views(self):
        return int(self.c2w.shape[0])

    def opts(self, dmin=0.01, dmax=10.0):
        return np.array([self.width, self.height, dmin, dmax, 0.5 * self.voxel_size], dtype=np.float32)


def _shell_cells(nx, ny, nz):
    g = np.zeros((nx, ny, nz), bool)
    g[0], g[-1] = True, True
    g[:, 0], g[:, -1] = True, True
    g[:, :, 0], g[:, :, -1] = True, True
    return g


def make_cameras(n_views, rng, centre, semi=(3.0, 2.0), height=1.5):
    c2w = np.zeros((n_views, 4, 4), np.float64)
    for v in range(n_views):
        a = 2.0 * math.pi * v / max(n_views, 1)
        pos = np.array([centre[0] + semi[0] * math.cos(a), centre[1] + semi[1] * math.sin(a), height])
        tangent = np.array([-semi[0] * math.sin(a), semi[1] * math.cos(a)])
        yaw = math.atan2(tangent[1], tangent[0]) + rng.uniform(-0.3, 0.3)
        pitch = rng.uniform(-0.2, 0.2)
        f = np.array([math.cos(yaw) * math.cos(pitch), math.sin(yaw) * math.cos(pitch), math.sin(pitch)])
        right = np.cross(f, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(f, right)
        c2w[v, :3, 0], c2w[v, :3, 1], c2w[v, :3, 2], c2w[v, :3, 3] = right, down, f, pos
        c2w[v, 3, 3] = 1.0
    return c2w.astype(np.float32)


def make_scene(n_vox, n_views, width, height, seed=0, room=(10.0, 8.0, 3.2), voxel_size=None,
               blob_fraction=0.15):
    """Room-shell scene with exactly ``n_vox`` occupied cells (see module docstring)."""
    rng = np.random.default_rng(seed)
    a, b, c = room
    if voxel_size is None:
        area = 2.0 * (a * b + a * c + b * c)
        voxel_size = math.sqrt(area / ((1.0 - blob_fraction) * n_vox))
    vs = float(np.float32(voxel_size))
    nx, ny, nz = (max(3, int(round(d / vs))) for d in (a, b, c))
    g = _shell_cells(nx, ny, nz)
    if g.sum() > n_vox:
        raise ValueError(f"shell alone has {int(g.sum())} cells > n_vox={n_vox}; raise voxel_size")
    centre = (0.5 * nx * vs, 0.5 * ny * vs)
    # boxes on the floor, kept away from the camera corridor
    blob = np.zeros_like(g)
    order = []
    tries = 0
    while g.sum() + blob.sum() < n_vox and tries < 100000:
        tries += 1
        e = rng.uniform(0.3, 1.5, size=3)
        ex, ey, ez = (max(1, int(round(v / vs))) for v in e)
        x0 = int(rng.integers(1, max(2, nx - 1 - ex)))
        y0 = int(rng.integers(1, max(2, ny - 1 - ey)))
        bx, by = (x0 + 0.5 * ex) * vs - centre[0], (y0 + 0.5 * ey) * vs - centre[1]
        rad = math.hypot(bx / (0.3 * a), by / (0.25 * b))
        if abs(rad - 1.0) < 0.35 + 0.5 * max(e[0], e[1]) / 2.0:
            continue
        box = np.zeros_like(g)
        box[x0:x0 + ex, y0:y0 + ey, 1:1 + ez] = True
        inner = np.zeros_like(g)
        if ex > 2 and ey > 2 and ez > 1:
            inner[x0 + 1:x0 + ex - 1, y0 + 1:y0 + ey - 1, 1:ez] = True
        surf = box & ~inner & ~g & ~blob
        idx = np.argwhere(surf)
        blob |= surf
        order.append(idx)
    if g.sum() + blob.sum() < n_vox:
        raise RuntimeError("could not place enough blob cells")
    extra = int(g.sum() + blob.sum() - n_vox)
    if extra:
        last = np.concatenate(order[::-1], 0)[:extra]
        blob[last[:, 0], last[:, 1], last[:, 2]] = False
    occ_b = g | blob
    assert int(occ_b.sum()) == n_vox
    xyz = np.argwhere(occ_b)                      # lexicographic (x,y,z) = np.unique order
    origin = np.array([-0.5 * a - 0.03, -0.5 * b - 0.01, -0.02], dtype=np.float32)
    occ = np.zeros((nz, ny, nx), np.int32)
    occ[xyz[:, 2], xyz[:, 1], xyz[:, 0]] = np.arange(1, n_vox + 1, dtype=np.int32)
    points = (origin[None, :].astype(np.float64) + xyz * vs).astype(np.float32)
    s = width / DSLR["w"]
    intr = np.array([DSLR["fx"] * s, DSLR["fy"] * s, DSLR["cx"] * s, DSLR["cy"] * s], dtype=np.float32)
    cam_centre = (float(origin[0]) + centre[0], float(origin[1]) + centre[1])
    c2w = make_cameras(n_views, rng, cam_centre, semi=(0.3 * a, 0.25 * b), height=min(1.5, 0.47 * c))
    c2w[:, 2, 3] += origin[2]
    return Scene(occ=occ, points=points, grid_origin=origin, voxel_size=vs, c2w=c2w, intr=intr,
                 width=int(width), height=int(height))


def make_features_np(n_views, height, width, channels, seed=0):
    """float32 [V,H,W,C] unit vectors (numpy; small parity cases)."""
    out = np.empty((n_views, height, width, channels), np.float32)
    for v in range(n_views):
        r = np.random.default_rng(seed * 100003 + v)
        f = r.standard_normal((height, width, channels)).astype(np.float32)
        f /= np.linalg.norm(f, axis=-1, keepdims=True)
        out[v] = f
    return out


def make_features_torch(n_views, height, width, channels, device, seed=0, out=None, view_ids=None):
    """float32 [V,H,W,C] unit vectors generated on ``device`` (one generator seed per view).  ``view_ids``: the view
    index each of the V maps is seeded with (default 0 .. V-1), so that a rank holding views r, r+G, ... generates the
    very maps a single process holds for them."""
    import torch
    if out is None:
        out = torch.empty((n_views, height, width, channels), dtype=torch.float32, device=device)
    gen = torch.Generator(device=device)
    for v in range(n_views):
        gen.manual_seed(seed * 100003 + (v if view_ids is None else int(view_ids[v])))
        f = out[v]
        f.normal_(generator=gen)
        f /= f.norm(dim=-1, keepdim=True)
    return out
