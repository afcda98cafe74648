"""ctypes front end of the CPU ORACLE (test infrastructure, not a product path).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The arithmetic lives in projector_oracle.c (which cites the
reference lines it restates); this file only marshals numpy arrays and adds the
pure-Python restatements of the reference's *Python* callers:

  build_occupancy          build_sparse_occupancy.py:30-53
  dpf_select_outputs       debug_project_features.py:35-45, 237-256
  aggregate_views          aggregate_voxel_features_onthefly.py:307-313, 381-451
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_LIB: another build of the same source (the sanitizer build of `make -C oracle sanitize`)
_LIB_PATH = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build(force=False):
    """Compile oracle/_build/liboracle.so with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "projector_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_rgb_project.restype = ctypes.c_int64
    return _lib


def host_cores():
    """Threads worth starting: the affinity mask capped by the cgroup CPU quota (containers often show every
    logical CPU of the host but grant a fraction; oversubscribed OpenMP teams run several times slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct)) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def project_features(feats, occ, vmi, intr, opts, grid_origin, voxel_size, count, out,
                     want_hits=True, want_f64=False, want_steps=False, nthreads=0):
    """project_features_cuda(...) semantics on numpy arrays; count/out accumulate in place.

    feats f32 [B,V,H,W,C]; occ i64 [B,Z,Y,X]; vmi f32 [B*V*16]; intr f32 [B,4]; opts f32 [5];
    grid_origin f32 [3]; count i32 [n_rows]; out f32 [n_rows,C].
    Returns dict(rc, hits i32 [B,V,H,W] or None, out64 or None, steps or None).
    """
    feats = _c(feats, np.float32)
    occ = _c(occ, np.int64)
    vmi = _c(vmi, np.float32).reshape(-1)
    intr = _c(intr, np.float32)
    opts = _c(opts, np.float32)
    go = _c(grid_origin, np.float32)
    B, V, H, W, C = feats.shape
    assert occ.ndim == 4 and occ.shape[0] == B
    _, Z, Y, X = occ.shape
    Wo, Ho = int(np.float32(opts[0]) + np.float32(0.5)), int(np.float32(opts[1]) + np.float32(0.5))
    assert (Wo, Ho) == (W, H), "oracle requires opts W,H to equal the tensor's"
    assert count.dtype == np.int32 and out.dtype == np.float32 and out.flags.c_contiguous
    n_rows = count.shape[0]
    assert out.shape == (n_rows, C)
    hits = np.zeros((B, V, H, W), np.int32) if want_hits else None
    out64 = np.zeros((n_rows, C), np.float64) if want_f64 else None
    steps = np.zeros((B, V, H, W), np.int32) if want_steps else None
    rc = lib().oracle_project_features(
        _p(feats, ctypes.c_float), _p(occ, ctypes.c_int64), _p(vmi, ctypes.c_float),
        _p(intr, ctypes.c_float), _p(opts, ctypes.c_float), _p(go, ctypes.c_float),
        ctypes.c_float(voxel_size), B, V, C, Z, Y, X,
        _p(count, ctypes.c_int32), _p(out, ctypes.c_float), ctypes.c_int64(n_rows),
        _p(hits, ctypes.c_int32), _p(out64, ctypes.c_double), _p(steps, ctypes.c_int32),
        int(nthreads) or host_cores())
    return dict(rc=rc, hits=hits, out64=out64, steps=steps)


def first_hit(occ, vmi, intr, opts, grid_origin, voxel_size, B, V, want_steps=False, nthreads=0):
    """First-hit ID image i32 [B,V,H,W] only (no feature traffic)."""
    occ = _c(occ, np.int64)
    vmi = _c(vmi, np.float32).reshape(-1)
    intr = _c(intr, np.float32)
    opts = _c(opts, np.float32)
    go = _c(grid_origin, np.float32)
    _, Z, Y, X = occ.shape
    W, H = int(np.float32(opts[0]) + np.float32(0.5)), int(np.float32(opts[1]) + np.float32(0.5))
    hits = np.zeros((B, V, H, W), np.int32)
    steps = np.zeros((B, V, H, W), np.int32) if want_steps else None
    lib().oracle_first_hit(_p(occ, ctypes.c_int64), _p(vmi, ctypes.c_float), _p(intr, ctypes.c_float),
                           _p(opts, ctypes.c_float), _p(go, ctypes.c_float), ctypes.c_float(voxel_size),
                           B, V, Z, Y, X, _p(hits, ctypes.c_int32), _p(steps, ctypes.c_int32),
                           int(nthreads) or host_cores())
    return (hits, steps) if want_steps else hits


def ray(m16, intr4, dmin, dmax, x, y):
    out = np.zeros(9, np.float32)
    lib().oracle_ray(_p(_c(m16, np.float32).reshape(-1), ctypes.c_float), _p(_c(intr4, np.float32), ctypes.c_float),
                     ctypes.c_float(dmin), ctypes.c_float(dmax), int(x), int(y), _p(out, ctypes.c_float))
    return out


def rgb_project(occ_zyx, c2w, intr4, grid_origin, voxel_size, img):
    """debug_project_colors.py:54-81.  Returns (colors f32 [n,3], zyx i32 [n,3], uv i32 [n,2])."""
    occ = _c(occ_zyx, np.int32)
    Z, Y, X = occ.shape
    img = _c(img, np.uint8)
    ih, iw = img.shape[:2]
    nmax = int((occ > 0).sum())
    colors = np.zeros((nmax, 3), np.float32)
    zyx = np.zeros((nmax, 3), np.int32)
    uv = np.zeros((nmax, 2), np.int32)
    n = lib().oracle_rgb_project(_p(occ, ctypes.c_int32), Z, Y, X,
                                 _p(_c(c2w, np.float32).reshape(-1), ctypes.c_float),
                                 _p(_c(intr4, np.float32), ctypes.c_float),
                                 _p(_c(grid_origin, np.float32), ctypes.c_float), ctypes.c_double(voxel_size),
                                 _p(img, ctypes.c_uint8), ih, iw,
                                 _p(colors, ctypes.c_float), _p(zyx, ctypes.c_int32), _p(uv, ctypes.c_int32))
    return colors[:n], zyx[:n], uv[:n]


def dpf_diagnostics(occ_zyx, c2w, intr4, grid_origin, voxel_size, img_h, img_w):
    """debug_project_features.py:59-84.  Returns dict(n_front, n_in_bounds, umin, umax, vmin, vmax)."""
    occ = _c(occ_zyx, np.int32)
    Z, Y, X = occ.shape
    nf = ctypes.c_int64(0)
    nb = ctypes.c_int64(0)
    st = np.zeros(4, np.float64)
    lib().oracle_dpf_diagnostics(_p(occ, ctypes.c_int32), Z, Y, X,
                                 _p(_c(c2w, np.float32).reshape(-1), ctypes.c_float),
                                 _p(_c(intr4, np.float32), ctypes.c_float),
                                 _p(_c(grid_origin, np.float32), ctypes.c_float), ctypes.c_double(voxel_size),
                                 int(img_h), int(img_w), ctypes.byref(nf), ctypes.byref(nb),
                                 _p(st, ctypes.c_double))
    return dict(n_front=nf.value, n_in_bounds=nb.value, umin=st[0], umax=st[1], vmin=st[2], vmax=st[3])


# ----------------------------------------------------------------------------------------------
# Pure-Python restatements of the reference's Python callers (small inputs only).
# ----------------------------------------------------------------------------------------------

def build_occupancy(points_xyz, grid_origin, voxel_size):
    """build_sparse_occupancy.py:30-53: points f32 [N,3] -> dense int32 [Z,Y,X], ID = row index + 1.

    coords = np.round((pts - origin) / voxel_size) (half-to-even, float32 pts and origin, python-float
    voxel_size keeps float32 under NumPy 2) :32; shift to zero if any coordinate is negative :36-39;
    dims = max + 1 :41; later rows overwrite earlier ones on collisions :45-46.
    """
    pts = np.asarray(points_xyz, dtype=np.float32)
    origin = np.array(grid_origin, dtype=np.float32)
    coords = np.round((pts - origin) / voxel_size).astype(np.int64)
    min_coord = coords.min(axis=0)
    if not np.all(min_coord >= 0):
        coords -= min_coord
    dims = coords.max(axis=0) + 1
    occ = np.zeros(dims[::-1], dtype=np.int32)
    for i, c in enumerate(coords):
        occ[tuple(c[::-1])] = i + 1
    return occ


def dpf_select_outputs(occ_zyx, count, sums):
    """debug_project_features.py:35-45,237-256: hit IDs -> (z,y,x) via the reverse map, fp16 sums.

    Returns (projected_feats f16 [n,C], projected_indices i32 [n,3]) in ascending-ID order, dropping
    IDs that the grid no longer holds (overwritten duplicates keep -1 in the map, :245-248).
    """
    occ = np.asarray(occ_zyx)
    max_id = int(occ.max()) if occ.size else 0
    id_to_zyx = np.full((max_id + 1, 3), -1, np.int64)
    nz = np.argwhere(occ != 0)
    if nz.size:
        id_to_zyx[occ[nz[:, 0], nz[:, 1], nz[:, 2]]] = nz
    hit_ids = np.nonzero(np.asarray(count) > 0)[0]
    idx = id_to_zyx[hit_ids].astype(np.int32)
    feats = np.asarray(sums)[hit_ids]
    valid = idx[:, 0] != -1
    with np.errstate(over="ignore"):                       # sums beyond 65504 become inf, as torch's .to(float16) makes them
        return feats[valid].astype(np.float16), idx[valid]


def aggregate_views(per_view_outputs, grid_origin, voxel_size):
    """aggregate_voxel_features_onthefly.py:307-313 and 381-451.

    per_view_outputs: iterable of (projected_feats f16 [n,C], projected_indices i32 [n,3]).
    The running sum is a *float16* tensor per (z,y,x) key (first feat.clone(), then += in fp16, :309-312),
    hit_count counts VIEWS (:313); avg = sum / count in fp16 (:385); key order is first
    insertion (:393).  Returns dict(xyz f32 [n,3], avg_feats f16 [n,C], voxel_coords i32 [n,3],
    hit_count i32 [n]).
    """
    sums, counts = {}, {}
    for feats, indices in per_view_outputs:
        for idx, feat in zip(indices, feats):
            k = tuple(int(v) for v in idx)
            if k not in sums:
                sums[k] = np.array(feat, dtype=np.float16, copy=True)
                counts[k] = 0
            else:
                sums[k] = (sums[k] + feat.astype(np.float16)).astype(np.float16)
            counts[k] += 1
    keys = list(sums.keys())
    if not keys:
        return dict(xyz=np.zeros((0, 3), np.float32), avg_feats=np.zeros((0, 0), np.float16),
                    voxel_coords=np.zeros((0, 3), np.int32), hit_count=np.zeros((0,), np.int32))
    # torch: half tensor / python int -> half (computed in float, rounded once)
    avg = np.stack([(sums[k].astype(np.float32) / np.float32(counts[k])).astype(np.float16) for k in keys], 0)
    go = np.array(grid_origin, dtype=np.float64)
    xyz = np.array([np.array([k[2], k[1], k[0]]) * voxel_size + go for k in keys], dtype=np.float32)
    return dict(xyz=xyz, avg_feats=avg, voxel_coords=np.array(keys, dtype=np.int32),
                hit_count=np.array([counts[k] for k in keys], dtype=np.int32))
