"""CPU ORACLE for the feature-map up-sampler (test infrastructure, NOT a product path).

Restates, in numpy float32, the arithmetic the reference applies to every LSeg feature map before the projector sees
it (cuda_project_image_to_sparse_voxel/prepare_tensor_data.py:119-127,152,183-185):

    for c in range(C): arr_upsampled[c] = cv2.resize(arr[c].astype(np.float32), (W, H), interpolation=cv2.INTER_LINEAR)
    arr = arr_upsampled.astype(arr.dtype)          # back to the file's dtype (float16 for LSeg maps)      PTD:126
    torch.from_numpy(arr).float() ... permute(0, 1, 3, 4, 2)   # float32, channels-last                  PTD:152,183

cv2 is a third-party dependency that is absent from /root/reference and from this image, and the reference does not pin
its version ("not listed", cuda_requirement.txt:10).  PARITY STATUS: this row is therefore pinned to OpenCV's PUBLISHED
algorithm for INTER_LINEAR on CV_32F images (modules/imgproc/src/resize.cpp: resizeGeneric_ with HResizeLinear /
VResizeLinear and the coefficient set-up of cv::resize), not to a run of cv2 itself -- "parity unpinned" in the sense of
the task statement, and DESIGN.md says so.  The rule:

    inv_scale_x = (double)W / w;  scale_x = 1. / inv_scale_x                      (likewise y)
    fx = (float)((dx + 0.5) * scale_x - 0.5);  sx = cvFloor(fx);  fx -= sx
    if sx < 0:       sx = 0,     fx = 0          (xmin bookkeeping)
    if sx >= w - 1:  sx = w - 1, fx = 0          (xmax bookkeeping: those columns copy S[sx] * 1)
    alpha = (1.f - fx, fx)                        float32
    fy, sy likewise, but WITHOUT zeroing at the edges: the two source rows are clip(sy, 0, h-1) and clip(sy+1, 0, h-1)
    beta = (1.f - fy, fy)
    horizontal pass: D[dx] = S[sx]*alpha0 + S[sx+1]*alpha1             (float32 multiplies and adds, rounded separately)
    vertical pass:   dst   = D0*beta0 + D1*beta1

OpenCV's SIMD builds may evaluate the vertical pass as fma(D0, beta0, D1*beta1); that last-bit, build-dependent choice is
not part of the rule.  tests/test_resize_oracle_cpu.py checks this restatement against an independent float64
evaluation of bilinear interpolation with half-pixel centres and edge clamp (the fp16 results may differ by one fp16
ulp where float32 and float64 round differently) and, when torch is importable, against
torch.nn.functional.interpolate(bilinear, align_corners=False), the same sampling rule in another library
(prepare_tensor_data_color.py:102-105 uses it on the colour path).

Only tests/ may import this module.
"""
import numpy as np


def linear_coefficients(dst_n, src_n, zero_at_edges):
    """(index int32 [dst_n], frac float32 [dst_n]) of cv::resize's INTER_LINEAR set-up along one axis."""
    scale = 1.0 / (float(dst_n) / float(src_n))                       # python floats are IEEE doubles
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if zero_at_edges:
        lo = s < 0
        s[lo], f[lo] = 0, 0.0
        hi = s >= src_n - 1
        s[hi], f[hi] = src_n - 1, 0.0
    return s, f


def resize_linear_f32(src, H, W):
    """src float32 [C,h,w] -> float32 [C,H,W], OpenCV's INTER_LINEAR rule (see module docstring)."""
    src = np.ascontiguousarray(src, dtype=np.float32)
    C, h, w = src.shape
    sx, fx = linear_coefficients(W, w, zero_at_edges=True)
    sy, fy = linear_coefficients(H, h, zero_at_edges=False)
    a0, a1 = (np.float32(1.0) - fx).astype(np.float32), fx
    b0, b1 = (np.float32(1.0) - fy).astype(np.float32), fy
    two = sx < w - 1
    sx1 = np.minimum(sx + 1, w - 1)
    # horizontal pass on every source row: D[c, y, dx]
    left, right = src[:, :, sx], src[:, :, sx1]
    D = np.where(two[None, None, :], left * a0[None, None, :] + right * a1[None, None, :], left * np.float32(1.0)).astype(np.float32)
    y0, y1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)
    out = D[:, y0, :] * b0[None, :, None] + D[:, y1, :] * b1[None, :, None]
    return out.astype(np.float32)


def upsample_features(arr, H, W, keep_dtype=False):
    """prepare_tensor_data.py:119-127,152,183-185: [C,h,w] (float16 or float32) -> channels-last [H,W,C], float32 or, with
    keep_dtype, the file's own dtype (same values: PTD:126 casts back before PTD:152 widens)."""
    arr = np.asarray(arr)
    up = resize_linear_f32(arr.astype(np.float32), H, W).astype(arr.dtype)          # PTD:125-126
    if not keep_dtype:
        up = up.astype(np.float32)                                                   # PTD:152
    return np.ascontiguousarray(np.transpose(up, (1, 2, 0)))                         # PTD:183-185


def bilinear_f64(arr, H, W):
    """Independent check: bilinear interpolation with half-pixel centres and edge clamp, all in float64."""
    a = np.asarray(arr, dtype=np.float64)
    C, h, w = a.shape
    x = (np.arange(W) + 0.5) * (w / W) - 0.5
    y = (np.arange(H) + 0.5) * (h / H) - 0.5
    x = np.clip(x, 0, w - 1)
    y = np.clip(y, 0, h - 1)
    x0 = np.floor(x).astype(int); x1 = np.minimum(x0 + 1, w - 1); tx = x - x0
    y0 = np.floor(y).astype(int); y1 = np.minimum(y0 + 1, h - 1); ty = y - y0
    top = a[:, y0][:, :, x0] * (1 - tx) + a[:, y0][:, :, x1] * tx
    bot = a[:, y1][:, :, x0] * (1 - tx) + a[:, y1][:, :, x1] * tx
    return top * (1 - ty)[None, :, None] + bot * ty[None, :, None]
