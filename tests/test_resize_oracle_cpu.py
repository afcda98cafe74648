"""The up-sampler's oracle (oracle/resize_oracle.py = OpenCV's published INTER_LINEAR rule in numpy float32) against two
independent evaluations of the same sampling rule: bilinear interpolation with half-pixel centres and edge clamp in
float64, and torch's interpolate(bilinear, align_corners=False) -- the library the reference itself uses on the colour
path (prepare_tensor_data_color.py:102-105).  cv2 itself is absent from the reference tree and from this image, so this
is the pin this row can have ("parity unpinned" for the resize)."""
import numpy as np
import pytest
import torch

from oracle import resize_oracle as ro


CASES = [((5, 6, 9), (12, 18)), ((3, 7, 5), (7, 5)), ((4, 9, 13), (15, 21)), ((2, 1, 1), (4, 3)), ((2, 8, 8), (4, 4)),
         ((6, 36, 54), (58, 87)), ((3, 10, 10), (27, 16))]


@pytest.mark.parametrize("shape,size", CASES)
def test_resize_rule_agrees_with_float64_bilinear_and_torch(shape, size):
    rng = np.random.default_rng(sum(shape) + sum(size))
    arr = rng.standard_normal(shape).astype(np.float16)
    H, W = size
    got = ro.resize_linear_f32(arr.astype(np.float32), H, W)
    ref64 = ro.bilinear_f64(arr, H, W)
    scale = np.abs(arr.astype(np.float64)).max()
    assert np.abs(got - ref64).max() <= 4e-6 * scale                  # float32 evaluation of the same three lerps
    t = torch.nn.functional.interpolate(torch.from_numpy(arr.astype(np.float32))[None], size=(H, W), mode="bilinear",
                                        align_corners=False)[0].numpy()
    # torch derives the source coordinate in float32 (area_pixel_compute_scale<float>), OpenCV in double: the weights
    # differ by ~1e-6, the values by a few 1e-6 of the data's scale
    assert np.abs(got - t).max() <= 2e-5 * scale
    # after the cast back to the file's dtype (PTD:126) the float32 and float64 evaluations differ by at most one fp16
    # ulp of the value (plus the float32 evaluation error itself where the value is a near-zero cancellation residue)
    a16, b16 = got.astype(np.float16), ref64.astype(np.float16)
    ulp = np.spacing(np.abs(b16).astype(np.float16)).astype(np.float64)
    assert (np.abs(a16.astype(np.float64) - b16.astype(np.float64)) <= ulp + 4e-6 * scale).all()
    assert (a16 != b16).mean() < 0.02


def test_coefficients_follow_the_published_edge_rules():
    # 2x up-sampling of 4 columns: fx = (dx + .5)/2 - .5 -> -0.25, 0.25, 0.75, ... ; sx < 0 is clamped with weight 0,
    # the last source column is copied (sx = w-1, fx = 0)
    s, f = ro.linear_coefficients(8, 4, zero_at_edges=True)
    assert s.tolist() == [0, 0, 0, 1, 1, 2, 2, 3]
    assert np.allclose(f, [0, 0.25, 0.75, 0.25, 0.75, 0.25, 0.75, 0.0])
    s, f = ro.linear_coefficients(8, 4, zero_at_edges=False)            # rows: no zeroing, the row index is clipped later
    assert s.tolist() == [-1, 0, 0, 1, 1, 2, 2, 3] and np.allclose(f, [0.75, 0.25, 0.75, 0.25, 0.75, 0.25, 0.75, 0.25])
    # identity size: every output pixel is its source pixel
    arr = np.random.default_rng(0).standard_normal((3, 5, 7)).astype(np.float32)
    assert np.array_equal(ro.resize_linear_f32(arr, 5, 7), arr)
    # constants stay constant bit for bit (a0 + a1 == 1 exactly for these weights is not needed: v*a0 + v*a1)
    c = np.full((1, 4, 4), 0.5, np.float32)
    assert np.array_equal(ro.resize_linear_f32(c, 8, 8), np.full((1, 8, 8), 0.5, np.float32))


def test_upsample_features_layout_and_dtype_round_trip():
    arr = np.random.default_rng(1).standard_normal((5, 6, 9)).astype(np.float16)
    up = ro.upsample_features(arr, 12, 18)
    assert up.shape == (12, 18, 5) and up.dtype == np.float32
    assert np.array_equal(up, up.astype(np.float16).astype(np.float32))          # PTD:126: fp16-representable values
    kept = ro.upsample_features(arr, 12, 18, keep_dtype=True)
    assert kept.dtype == np.float16 and np.array_equal(kept.astype(np.float32), up)
