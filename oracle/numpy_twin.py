"""Independent numpy restatement of the reference kernel (ORACLE cross-check; test infrastructure).

Written separately from projector_oracle.c, vectorised over pixels with an explicit loop over ray
steps, every operand held as np.float32 so each numpy ufunc is one IEEE binary32 operation (numpy
never contracts a*b+c).  Follows project_image_cuda_kernel.cu:157-187 (ray set-up) and :24-92
(march / accumulate); see projector_oracle.c for the full citation list and arithmetic contract.
tests/test_oracle_twin.py requires the two restatements to agree bit for bit.
"""
import numpy as np

F = np.float32


def _roundf(x):
    """C roundf: half away from zero, exact (x - trunc(x) is exact in binary32)."""
    t = np.trunc(x)
    d = np.abs(x - t)
    return t + np.where(d >= F(0.5), np.copysign(F(1.0), x), F(0.0)).astype(F)


def _normalize(x, y, z):
    inv = F(1.0) / np.sqrt(x * x + y * y + z * z)      # rsqrtf := 1/sqrtf (cutil_math.h:81-84)
    return x * inv, y * inv, z * inv


def rays(m, intr4, dmin, dmax, width, height):
    """camDir, camPos, worldDir for every pixel of one view; arrays are [H,W] float32."""
    m = np.asarray(m, F).reshape(4, 4)
    fx, fy, mx, my = (F(v) for v in intr4)
    dmin, dmax = F(dmin), F(dmax)
    ux = np.arange(width, dtype=np.uint32).astype(F)[None, :].repeat(height, 0)
    uy = np.arange(height, dtype=np.uint32).astype(F)[:, None].repeat(width, 1)
    depth = F(1.0) * (dmax - dmin) + dmin
    cx = depth * ((ux - mx) / fx)
    cy = depth * ((uy - my) / fy)
    cz = np.full_like(cx, depth)
    cdx, cdy, cdz = _normalize(cx, cy, cz)
    zero, one = F(0.0), F(1.0)
    pos = [m[i, 0] * zero + m[i, 1] * zero + m[i, 2] * zero + m[i, 3] * one for i in range(3)]
    w = [m[i, 0] * cdx + m[i, 1] * cdy + m[i, 2] * cdz + m[i, 3] * zero for i in range(3)]
    wdx, wdy, wdz = _normalize(*w)
    return (cdx, cdy, cdz), pos, (wdx, wdy, wdz)


def first_hit(occ_zyx, m, intr4, opts, grid_origin, voxel_size):
    """First-hit ID image int32 [H,W] of one view against a dense [Z,Y,X] grid."""
    occ = np.asarray(occ_zyx)
    dimz, dimy, dimx = occ.shape
    width = int(F(opts[0]) + F(0.5))
    height = int(F(opts[1]) + F(0.5))
    dmin, dmax, inc = F(opts[2]), F(opts[3]), F(opts[4])
    fx, fy, mx, my = (F(v) for v in intr4)
    ox, oy, oz = (F(v) for v in grid_origin)
    vs = F(voxel_size)
    (cdx, cdy, cdz), pos, (wdx, wdy, wdz) = rays(m, intr4, dmin, dmax, width, height)
    d2r = F(1.0) / cdz
    t = d2r * dmin
    t_end = d2r * dmax
    hit = np.zeros((height, width), np.int32)
    done = np.zeros((height, width), bool)
    with np.errstate(all="ignore"):
        while True:
            active = (~done) & (t < t_end)
            if not active.any():
                break
            px, py, pz = pos[0] + t * wdx, pos[1] + t * wdy, pos[2] + t * wdz
            ix = _roundf((px - ox) / vs)
            iy = _roundf((py - oy) / vs)
            iz = _roundf((pz - oz) / vs)
            camx, camy, camz = cdx * t, cdy * t, cdz * t
            u = fx * (camx / camz) + mx
            v = fy * (camy / camz) + my
            inb = (u >= 0) & (u < F(width)) & (v >= 0) & (v < F(height))
            ing = (ix >= 0) & (iy >= 0) & (iz >= 0) & (ix < dimx) & (iy < dimy) & (iz < dimz)
            ok = active & inb & ing
            jx = np.where(ok, ix, 0).astype(np.int64)
            jy = np.where(ok, iy, 0).astype(np.int64)
            jz = np.where(ok, iz, 0).astype(np.int64)
            ids = occ[jz, jy, jx].astype(np.int64).astype(np.int32)
            newhit = ok & (ids != 0)
            hit[newhit] = ids[newhit]
            done |= newhit
            done |= ~active
            t = np.where(done, t, t + inc).astype(F)
    return hit


def accumulate(hit, feats_hwc, count, out):
    """count[id] += 1 and out[id,:] += feat in (y,x) raster order, float32 adds."""
    H, W = hit.shape
    flat = hit.reshape(-1)
    f = feats_hwc.reshape(H * W, -1)
    for i in np.nonzero(flat)[0]:
        count[flat[i]] += 1
        out[flat[i]] = out[flat[i]] + f[i]
