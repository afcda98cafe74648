"""How far can a real CUDA build of the reference be from the arithmetic contract?  The contract (oracle/projector_oracle.c)
reads the reference source as IEEE operations: rsqrtf = 1/sqrtf, no FMA.  nvcc builds the device code with the GPU's
approximate rsqrtf (<= 2 ulp) and contracts mul+add into FMA, which this image cannot reproduce bit for bit.  This test
builds the oracle a second time "device-like" (FMA contraction on, correctly rounded rsqrt nudged by -2..+2 ulp) and
measures the fraction of pixels whose first-hit voxel differs from the contract's: the expected order of divergence
between this repo's bit-exact-to-the-contract kernel and the reference's CUDA binary.  It is a handful of boundary rays
per ten thousand, never a structural difference; DESIGN.md quotes the numbers printed here."""
import ctypes
import os

import numpy as np
import pytest

from synthetic_scene import make_scene

HERE = os.path.dirname(os.path.abspath(__file__))
DEV = os.path.join(os.path.dirname(HERE), "oracle", "_build", "liboracle_devlike.so")


def _first_hit(lib, s, V):
    W, H = s.width, s.height
    occ = np.ascontiguousarray(s.occ[None].astype(np.int64))
    hits = np.zeros((1, V, H, W), np.int32)
    p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))   # noqa: E731
    vmi = np.ascontiguousarray(s.c2w[:V].reshape(-1), np.float32)
    intr = np.ascontiguousarray(s.intr[None], np.float32)
    opts = np.ascontiguousarray(s.opts(), np.float32)
    go = np.ascontiguousarray(s.grid_origin, np.float32)
    lib.oracle_first_hit(p(occ, ctypes.c_int64), p(vmi, ctypes.c_float), p(intr, ctypes.c_float), p(opts, ctypes.c_float),
                         p(go, ctypes.c_float), ctypes.c_float(s.voxel_size), 1, V, *occ.shape[1:], p(hits, ctypes.c_int32), None, 0)
    return hits


def test_device_like_builds_change_only_boundary_rays(oracle_mod):
    oracle_mod.build()
    if not os.path.exists(DEV):
        pytest.skip("device-like oracle variant not built")
    base, dev = oracle_mod.lib(), ctypes.CDLL(DEV)
    assert base.oracle_is_devlike() == 0 and dev.oracle_is_devlike() == 1
    rows = []
    for n_vox, W, H, V in ((20000, 242, 137, 6), (80000, 484, 274, 3)):
        s = make_scene(n_vox, 100, W, H, seed=0)
        ref = _first_hit(base, s, V)
        assert (ref > 0).mean() > 0.99
        for bias in (0, -2, 2):
            dev.oracle_set_rsqrt_bias(bias)
            got = _first_hit(dev, s, V)
            diff = got != ref
            frac = float(diff.mean())
            # where: the first pixel row / column -- there the reference's own (u,v) bounds test (K.cu:53-61)
            # sits within rounding distance of 0 and a different rounding flips a whole ray between "hits" and "never
            # hits" -- and everywhere else, where a sample lands within rounding distance of a cell boundary
            border = np.zeros_like(diff)
            border[..., 0, :] = True
            border[..., :, 0] = True
            inner = float((diff & ~border).sum() / (~border).sum())
            rows.append((n_vox, f"{W}x{H}", bias, round(frac, 6), round(inner, 6), round(float((diff & border).sum() / border.sum()), 4)))
            assert inner < 1e-3, rows[-1]
            assert (((got > 0) == (ref > 0)) | border).mean() > 0.9999      # away from the border only the voxel changes
    print("\nfirst-hit pixels differing from the IEEE contract")
    print("    (voxels, image, rsqrt ulp bias, fraction of all pixels, of interior pixels, of first-row/column pixels):")
    for r in rows:
        print("   ", r)
