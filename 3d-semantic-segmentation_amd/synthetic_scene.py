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

``trajectory=True`` (round 5) is the second generator mode: what a hand-held capture of a real room looks like to the
projector, as opposed to the benign room above (independent poses on a corridor kept clear of geometry, every ray hits,
at most ~9 k pixels per voxel and call).  The reference's authors ran a ScanNet++ DSLR trajectory
(aggregate_voxel_features_onthefly.py:101-106,209: the first 216 frames in file order, i.e. in capture order):

  * cameras -- ONE smooth path stepped frame by frame: consecutive views at most 5 cm and 3 degrees apart, scripted as
               shots -- close-up dwells 0.25-0.4 m in front of a wall or a box (0.5-1 cm and < 1 degree per frame: the same
               surface patch fills 40-60 consecutive frames), transits, and a long look through the missing wall segment;
  * geometry -- clutter fraction 0.5 (half of the occupied cells are boxes standing in the room, anywhere but within
               0.25 m of a camera position), and a large opening in the +x wall and in the ceiling above it: rays through
               them leave the grid and march to depthMax without a hit.
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


def _shell_cells(nx, ny, nz, openings=False):
    g = np.zeros((nx, ny, nz), bool)
    g[0], g[-1] = True, True
    g[:, 0], g[:, -1] = True, True
    g[:, :, 0], g[:, :, -1] = True, True
    if openings:
        # trajectory mode: the middle 60 % of the +x wall from 10 % to 95 % of its height, and the strip of ceiling next to it
        y0, y1 = int(round(0.2 * ny)), int(round(0.8 * ny))
        g[-1, y0:y1, int(round(0.1 * nz)):int(round(0.95 * nz))] = False
        g[int(round(0.7 * nx)):nx - 1, y0:y1, -1] = False
    return g


def _look_at(pos, fwd):
    """Row-major camera->world of a camera at ``pos`` looking along ``fwd`` (x right, y down, z forward; world z up)."""
    f = np.asarray(fwd, np.float64)
    f = f / np.linalg.norm(f)
    right = np.cross(f, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    down = np.cross(f, right)
    m = np.zeros((4, 4), np.float64)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, down, f, pos
    m[3, 3] = 1.0
    return m


def _dir(yaw_deg, pitch_deg):
    y, p = math.radians(yaw_deg), math.radians(pitch_deg)
    return np.array([math.cos(y) * math.cos(p), math.sin(y) * math.cos(p), math.sin(p)])


MAX_STEP_M, MAX_TURN_DEG = 0.05, 3.0          # consecutive views of a trajectory are never farther apart


def make_trajectory(n_views, room, seed=0):
    """Hand-held camera path inside a room of interior (a, b, c) metres whose min corner is the world origin: positions
    float64 [V,3], forward directions [V,3].  Stepped frame by frame through a script of shots (target position, target
    viewing direction, speed in m and degrees per frame, frames); a shot moves towards its targets at the given speeds and
    never faster than MAX_STEP_M / MAX_TURN_DEG per frame, a smooth tremor of a few millimetres and tenths of a degree on
    top.  The script is a closed loop (about 350-450 frames, by the size of the room) that repeats; shorter paths are its first frames."""
    a, b, c = room
    h = min(1.45, 0.5 * c)
    # target (x, y, z), target (yaw, pitch) in degrees, metres per frame, degrees per frame, frames (None: a transit, which
    # lasts until both targets are reached)
    script = [
        # close-up of the -y wall from 0.27 m, creeping sideways: one patch of wall fills these frames
        ((0.30 * a, 0.27, h), (-90.0, -4.0), 0.006, 0.25, 60),
        # pull back, turn along the wall and walk to the +x end
        ((0.42 * a, 0.80, h), (-35.0, -8.0), 0.045, 2.6, None),
        ((0.80 * a, 0.47 * b, h + 0.05), (0.0, 2.0), 0.045, 2.4, None),
        # the look through the opening in the +x wall (and up through the open strip of ceiling)
        ((0.80 * a + 0.5, 0.50 * b, h + 0.05), (6.0, 10.0), 0.012, 0.6, 50),
        # turn back into the room, down to the clutter: a close-up of whatever stands near the floor target
        ((0.60 * a, 0.62 * b, 0.75 * h), (150.0, -28.0), 0.045, 2.5, None),
        ((0.55 * a, 0.66 * b, 0.55 * h), (165.0, -38.0), 0.008, 0.4, 35),
        # on to the +y wall for a second, slanted close-up, then back to the start
        ((0.40 * a, b - 0.33, h), (100.0, -6.0), 0.045, 2.5, None),
        ((0.34 * a, b - 0.30, h), (80.0, -10.0), 0.007, 0.5, 40),
        ((0.27 * a, 0.30, h), (-90.0, -3.0), 0.045, 2.5, None),
    ]
    rng = np.random.default_rng(seed + 7919)
    ph = rng.uniform(0.0, 2.0 * math.pi, size=(2, 3))
    pos = np.array([0.27 * a, 0.30, h])
    fwd = _dir(-90.0, -3.0)
    P, F = np.zeros((n_views, 3)), np.zeros((n_views, 3))
    k = 0
    while k < n_views:
        for (tp, (tyaw, tpitch), sp, sa, frames) in script:
            tp = np.array(tp, np.float64)
            tf = _dir(tyaw, tpitch)
            done = 0
            while k < n_views:
                d = tp - pos
                dist = float(np.linalg.norm(d))
                ang = math.degrees(math.acos(max(-1.0, min(1.0, float(fwd @ tf)))))
                if (frames is None and dist < 0.02 and ang < 1.0) or (frames is not None and done >= frames) or done >= 400:
                    break
                step = min(sp, dist)
                if dist > 1e-9:
                    pos = pos + d * (step / dist)
                turn = min(sa, ang)
                if ang > 1e-6:
                    # rotate fwd towards tf by `turn` degrees (spherical interpolation)
                    w, th = turn / ang, math.radians(ang)
                    fwd = (math.sin((1.0 - w) * th) * fwd + math.sin(w * th) * tf) / math.sin(th)
                    fwd = fwd / np.linalg.norm(fwd)
                # hand-held tremor: smooth in the frame index, ~2 mm and ~0.15 degrees from frame to frame
                t = float(k)
                tremor_p = 0.004 * np.array([math.sin(0.31 * t + ph[0, 0]), math.sin(0.23 * t + ph[0, 1]), math.sin(0.41 * t + ph[0, 2])])
                yaw = math.degrees(math.atan2(fwd[1], fwd[0])) + 0.35 * math.sin(0.37 * t + ph[1, 0])
                pitch = math.degrees(math.asin(max(-1.0, min(1.0, float(fwd[2]))))) + 0.3 * math.sin(0.29 * t + ph[1, 1])
                P[k] = pos + tremor_p
                F[k] = _dir(yaw, pitch)
                k += 1
                done += 1
    return P, F


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


def make_scene(n_vox, n_views, width, height, seed=0, room=None, voxel_size=None,
               blob_fraction=None, trajectory=False):
    """Room-shell scene with exactly ``n_vox`` occupied cells (see module docstring).  ``trajectory=True``: the hand-held
    capture -- one smooth camera path with close-up dwells, clutter fraction 0.5, a wall segment and a strip of ceiling
    missing."""
    rng = np.random.default_rng(seed)
    if blob_fraction is None:
        blob_fraction = 0.5 if trajectory else 0.15
    if room is None:
        room = (6.0, 5.0, 2.8) if trajectory else (10.0, 8.0, 3.2)
    a, b, c = room
    if voxel_size is None:
        area = 2.0 * (a * b + a * c + b * c)
        voxel_size = math.sqrt(area / ((1.0 - blob_fraction) * n_vox))
    vs = float(np.float32(voxel_size))
    nx, ny, nz = (max(3, int(round(d / vs))) for d in (a, b, c))
    g = _shell_cells(nx, ny, nz, openings=trajectory)
    if g.sum() > n_vox:
        raise ValueError(f"shell alone has {int(g.sum())} cells > n_vox={n_vox}; raise voxel_size")
    centre = (0.5 * nx * vs, 0.5 * ny * vs)
    traj = None
    if trajectory:
        # the path first (it only needs the room): the clutter is then placed anywhere but within 0.25 m of a camera
        traj = make_trajectory(n_views, (nx * vs, ny * vs, nz * vs), seed=seed)
        cam_cells = traj[0] / vs
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
        if trajectory:
            # distance from every camera position to the box (cells): keep 0.25 m clear, no more -- boxes may stand right
            # in front of a camera, which is what a close-up of clutter is
            lo = np.array([x0, y0, 1.0]) - 0.5
            hi = np.array([x0 + ex, y0 + ey, 1.0 + ez]) - 0.5
            dd = np.maximum(np.maximum(lo - cam_cells, cam_cells - hi), 0.0)
            if float(np.sqrt((dd * dd).sum(1)).min()) * vs < 0.25:
                continue
        else:
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
    if trajectory:
        # cell (i,j,k) has its centre at origin + (i,j,k)*vs: the room's interior starts half a cell below the origin
        c2w = np.stack([_look_at(origin.astype(np.float64) - 0.5 * vs + traj[0][v], traj[1][v]) for v in range(n_views)]).astype(np.float32)
    else:
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
