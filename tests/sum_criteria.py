"""The acceptance criterion for feature SUMS, in one place: no per-test solidity thresholds.

north_star's bar is "per-voxel feature outputs match the reference kernel to 1e-4 relative (fp32 accumulate)".  A float32 sum of n
addends cannot promise a relative error on an element that is a cancellation residue of its addends, whatever the order -- the
reference's own atomics in arrival order do not either -- so the bar is stated in the form that is independent of cancellation,
the forward-error bound of a float32 sum:

    |got - ref64|  <=  K * sqrt(n) * 2^-24 * sum|addend|          for EVERY element of EVERY row            (F)

with n = the voxel's pixel count, ref64 and sum|addend| accumulated in float64 from the same pixels, K = 4 (stated once, here).
Any order of float32 additions satisfies (n - 1) * 2^-24 * sum|addend| in the worst case; typically the error is ~0.5 * 2^-24 *
sum|addend| when the addends' signs are random (these feature maps) and ~0.33 sqrt(n) * 2^-24 * sum|addend| when they share a sign,
so (F) is the worst case up to n = 17 and >= 12 standard deviations beyond.  What (F) cannot see on a very long row (a lost pixel
is an error of ~sum|addend| / n: below the bound once n > 26 000) the other assertions do: pixel counts and view counts are compared
exactly in every test, and (S) below has no sqrt(n).  On top of (F):

    (R)  every element within 1e-4 of its ROW's largest element (the bar at row level);
    (E)  every element that is at least 1 % of its row's largest element (one threshold, everywhere) and whose bound (F) is itself
         below 1e-4 * |ref64| is within 1e-4 of |ref64| -- where float32 can promise the element-wise bar, it is asserted;
    (S)  for rows summed in PARTS (split voxels), where the oracle's serial float32 sum is available:
         |got - ref64| <= 2 * |oracle32 - ref64| + 4 * 2^-24 * sum|addend| element-wise -- a split row is no worse than the serial
         float32 order it replaces (the additive term is ~8 standard deviations of the serial order's own error, for the elements on
         which the serial sum happens to land exactly).
"""
import numpy as np
import torch

U32 = 2.0 ** -24
K_FWD = 4.0
SOLID = 1e-2
BAR = 1e-4


def _t(x, dev, dtype=torch.float64):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=dev, dtype=dtype)


def abs_sums_from_hits(hits_views, feats_views, n_rows, dev):
    """(ref64, abs64) of float64 scatter-adds of the hit pixels' rows and of their absolute values.
    hits_views: int array [V,H,W] of first-hit IDs (0 = miss); feats_views: tensor or array [V,H,W,C]."""
    V = hits_views.shape[0]
    C = feats_views.shape[-1]
    ref = torch.zeros(n_rows, C, dtype=torch.float64, device=dev)
    ab = torch.zeros(n_rows, C, dtype=torch.float64, device=dev)
    for v in range(V):
        ids = torch.from_numpy(np.ascontiguousarray(hits_views[v]).reshape(-1).astype(np.int64)).to(dev)
        rows = _t(feats_views[v], dev).reshape(-1, C)
        ref.index_add_(0, ids, rows)
        ab.index_add_(0, ids, rows.abs())
        del rows
    ref[0] = 0
    ab[0] = 0
    return ref, ab


def assert_sums(got, ref64, abs64, count, split=None, oracle32=None, dev="cuda:0"):
    """got f32 [n,C]; ref64 / abs64 f64 [n,C]; count int [n] (pixels per row); split: bool [n] rows summed in parts (for (S), with
    oracle32 f32 [n,C]).  Returns dict of the largest observed ratios (for the test's own record)."""
    dev = torch.device(dev)
    got = _t(got, dev)
    ref64, abs64 = _t(ref64, dev), _t(abs64, dev)
    n = _t(count, dev)[:, None]
    err = (got - ref64).abs()
    bound = K_FWD * n.sqrt() * U32 * abs64
    bad = err > bound
    assert not bool(bad.any()), (f"(F) {int(bad.sum())} elements beyond {K_FWD} sqrt(n) 2^-24 sum|addend|; worst ratio "
                                 f"{float((err / bound.clamp_min(1e-300))[bad].max()):.3g}")
    row_mag = ref64.abs().amax(dim=1, keepdim=True)
    touched = (n > 0)[:, 0]
    assert float(got[~touched].abs().max().item() if bool((~touched).any()) else 0.0) == 0.0, "rows without pixels must stay untouched"
    rel_row = err[touched] / row_mag[touched].clamp_min(1e-300)
    assert float(rel_row.max().item()) <= BAR, f"(R) {float(rel_row.max().item()):.3e} of the row's magnitude"
    promised = touched[:, None] & (ref64.abs() >= SOLID * row_mag) & (bound <= BAR * ref64.abs())
    rel_el = err[promised] / ref64.abs()[promised]
    assert rel_el.numel() == 0 or float(rel_el.max().item()) <= BAR, f"(E) {float(rel_el.max().item()):.3e}"
    res = dict(fwd=float((err / bound.clamp_min(1e-300)).max().item()), rel_row=float(rel_row.max().item()),
               rel_el=float(rel_el.max().item()) if rel_el.numel() else 0.0, promised=int(promised.sum().item()),
               solid=int((touched[:, None] & (ref64.abs() >= SOLID * row_mag)).sum().item()))
    if split is not None and oracle32 is not None:
        sp = _t(split, dev, torch.bool)
        if bool(sp.any()):
            e_or = (_t(oracle32, dev)[sp] - ref64[sp]).abs()
            lim = 2.0 * e_or + 4.0 * U32 * abs64[sp]
            worse = err[sp] > lim
            assert not bool(worse.any()), (f"(S) {int(worse.sum())} elements of split rows are worse than twice the serial float32 "
                                           f"order's error + 4 * 2^-24 * sum|addend|")
            res["split_vs_serial"] = float((err[sp] / lim.clamp_min(1e-300)).max().item())
    return res


def assert_sums_vs_oracle(got, r, feats, count, split=None, oracle32=None, dev="cuda:0"):
    """assert_sums for a call the host oracle ran with want_f64: r["out64"] is the float64 accumulation, r["hits"] [B,V,H,W] the
    first-hit IDs; feats [B,V,H,W,C] (numpy or tensor, any float dtype) the feature maps the call read."""
    hits = np.asarray(r["hits"])
    H, W = hits.shape[-2:]
    f = feats.reshape((-1, H, W, feats.shape[-1]))
    _, abs64 = abs_sums_from_hits(hits.reshape(-1, H, W), f, r["out64"].shape[0], torch.device(dev))
    return assert_sums(got, r["out64"], abs64, count, split=split, oracle32=oracle32, dev=dev)
