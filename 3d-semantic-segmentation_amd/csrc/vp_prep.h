// vp_prep.h -- the data-format rows either side of the projector: the feature-map up-sampler that
// replaces prepare_tensor_data.py:119-127,183-185 and the occupancy builder that replaces build_sparse_occupancy.py:30-53.
// Included by voxproj.hip only.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// Feature-map up-sampler.  The reference resizes every channel of the fp16 [C,h,w] LSeg map with
// cv2.resize(..., INTER_LINEAR) in float32, casts the result back to the file's dtype and only then widens it to
// float32 and permutes to channels-last (PTD:119-127,152,183-185).  Here: one transpose pass [C,h,w] -> [h,w,C]
// (k_chw_to_hwc, 64x64 tiles through LDS) and one wavefront per output pixel that reads the four source rows
// (16 B per lane, served by L2: every source row is used by ~(H/h)*(W/w) output pixels) and writes the C-wide output
// row once.
//
// Arithmetic (the parity spec of this row -- OpenCV's published INTER_LINEAR rule for CV_32F, resize.cpp):
//   scale_x = 1.0 / ((double)W / w)                              (double, as cv::resize derives it from dsize)
//   fx = (float)((dx + 0.5) * scale_x - 0.5);  sx = floor(fx);  fx -= sx
//   sx < 0      -> sx = 0, fx = 0;     sx >= w-1 -> sx = w-1, fx = 0  (the tap at sx+1 then has weight 0 and is not read)
//   a0 = 1.f - fx, a1 = fx                                       (float)
//   rows: fy likewise WITHOUT the edge zeroing; sy0 = clamp(sy, 0, h-1), sy1 = clamp(sy+1, 0, h-1); b0 = 1.f - fy, b1 = fy
//   r0 = S[sy0][sx]*a0 + S[sy0][sx+1]*a1;  r1 = S[sy1][sx]*a0 + S[sy1][sx+1]*a1    (horizontal pass, float32)
//   out = r0*b0 + r1*b1                                                               (vertical pass, float32)
//   every multiply and add rounds separately (no FMA: the library is built -ffp-contract=off); then the cast back to
//   the source dtype (PTD:126; binary16 round-to-nearest-even) and, for a float32 destination, the exact widening.
// OpenCV's SIMD builds may fuse the vertical pass (v_muladd); that build-dependent last-bit choice is outside the spec,
// see DESIGN.md.  oracle/resize_oracle.py restates the same arithmetic in numpy float32.
// ------------------------------------------------------------------------------------------------
template <typename T, int N>
struct VecOf { typedef T type __attribute__((ext_vector_type(N))); };
template <typename T>
struct VecOf<T, 1> { typedef T type; };

template <typename T>
__global__ __launch_bounds__(256) void k_chw_to_hwc(const T *__restrict__ src, T *__restrict__ dst, int C, long long P)
{
    // src [C,P] -> dst [P,C]; 64x64 tile, 256 threads: 16 rows of 64 per pass (any shape; element-wise accesses)
    __shared__ T tile[64][65];
    const long long p0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r;
        const long long p = p0 + tx;
        if (c < C && p < P) tile[r][tx] = src[(long long)c * P + p];
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const long long p = p0 + r;
        const int c = c0 + tx;
        if (c < C && p < P) dst[p * C + c] = tile[tx][r];
    }
}

// The same transpose with 16-byte accesses on both sides (P and C multiples of E = 16 / sizeof(T), 16-byte aligned
// pointers): a lane reads E consecutive positions of one channel and writes E consecutive channels of one position, so
// a 64x64 tile moves with 1/E of the memory instructions of the element-wise kernel above.
template <typename T>
__global__ __launch_bounds__(256) void k_chw_to_hwc_v16(const T *__restrict__ src, T *__restrict__ dst, int C, long long P)
{
    constexpr int E = 16 / (int)sizeof(T);          // elements per 16 bytes: 8 (binary16) or 4 (float)
    constexpr int SEG = 64 / E;                     // 16-byte segments per 64-element tile row
    constexpr int PITCH = 64 + E;                   // keeps every row of the tile 16-byte aligned and staggers the banks
    typedef typename VecOf<T, E>::type V;
    __shared__ __attribute__((aligned(16))) T tile[64 * PITCH];     // [channel][position]
    const long long p0 = (long long)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int seg = threadIdx.x % SEG, row = threadIdx.x / SEG;     // 256 / SEG rows per pass
    for (int r = row; r < 64; r += 256 / SEG) {
        const int c = c0 + r;
        const long long p = p0 + seg * E;
        if (c < C && p < P) *reinterpret_cast<V *>(&tile[r * PITCH + seg * E]) = *reinterpret_cast<const V *>(src + (long long)c * P + p);
    }
    __syncthreads();
    for (int r = row; r < 64; r += 256 / SEG) {                      // now r = position inside the tile, seg = channel segment
        const long long p = p0 + r;
        const int c = c0 + seg * E;
        if (c < C && p < P) {
            V v;
#pragma unroll
            for (int e = 0; e < E; e++) v[e] = tile[(seg * E + e) * PITCH + r];
            *reinterpret_cast<V *>(dst + p * C + c) = v;
        }
    }
}

// one wavefront per output pixel, lanes over channels: VEC consecutive channels per lane (16-byte loads when
// VEC * sizeof(TS) == 16), consecutive lanes = consecutive channel groups, so every source and destination row moves as
// whole contiguous lines
constexpr int UPS_PIX = 8;   // consecutive output pixels per wavefront (fewer, longer-lived waves: the launch rate of
                             // one-pixel waves, not memory, bounded the first version of this kernel)

// NV > 0: the lane owns NV channel groups of 64*VEC (C <= 64*VEC*NV) and keeps the window in registers; NV = 0: any C, every
// tap loaded where it is used.
template <typename TS, typename TD, int VEC, int NV_>
__global__ __launch_bounds__(256) void k_upsample_hwc(const TS *__restrict__ src, TD *__restrict__ dst, int C, int h, int w,
                                                      int H, int W, double scale_x, double scale_y)
{
    const long long HW = (long long)H * W;
    const long long pix0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * UPS_PIX;
    const int lane = threadIdx.x & 63;
    typedef typename VecOf<TS, VEC>::type VS;
    typedef typename VecOf<TD, VEC>::type VD;
    // Sliding window over the source columns: consecutive output pixels of a row use the same source column or the next
    // one (up-sampling), so the window's right column becomes its left column and only one new column (two rows) is
    // loaded -- ~14 row loads per 8 pixels instead of 32.  All of it is wave-uniform control flow.
    constexpr int NV = NV_ > 0 ? NV_ : 1;
    constexpr bool small = NV_ > 0;
    VS l0[NV], l1[NV], r0v[NV], r1v[NV];
    int cur_dy = -1, cur_sx = -2;
    for (int q = 0; q < UPS_PIX; q++) {
        const long long pix = pix0 + q;
        if (pix >= HW) return;
        const int dy = (int)(pix / W), dx = (int)(pix - (long long)dy * W);
        float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { sx = 0; fx = 0.f; }
        if (sx >= w - 1) { sx = w - 1; fx = 0.f; }
        float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= (float)sy;
        const int sy0 = min(max(sy, 0), h - 1), sy1 = min(max(sy + 1, 0), h - 1);
        const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
        const bool two = sx < w - 1;      // at the right border OpenCV copies S[sx] (x >= xmax: D[dx] = S[sx] * 1)
        const TS *s00 = src + ((long long)sy0 * w + sx) * C, *s10 = src + ((long long)sy1 * w + sx) * C;
        TD *o = dst + pix * C;
        if (small) {
            const bool slide = (dy == cur_dy) && (sx == cur_sx + 1);
            if (dy != cur_dy || sx != cur_sx) {
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const int c = (k * 64 + lane) * VEC;
                    if (c < C) {
                        if (slide) { l0[k] = r0v[k]; l1[k] = r1v[k]; }
                        else { l0[k] = *reinterpret_cast<const VS *>(s00 + c); l1[k] = *reinterpret_cast<const VS *>(s10 + c); }
                        if (two) { r0v[k] = *reinterpret_cast<const VS *>(s00 + C + c); r1v[k] = *reinterpret_cast<const VS *>(s10 + C + c); }
                    }
                }
                cur_dy = dy; cur_sx = sx;
            }
        }
#pragma unroll
        for (int k = 0; k < NV; k++) {
            for (int c = (k * 64 + lane) * VEC; c < C; c += 64 * VEC * NV) {
                VS p00, p10, p01, p11;
                if (small) { p00 = l0[k]; p10 = l1[k]; p01 = two ? r0v[k] : p00; p11 = two ? r1v[k] : p10; }
                else {
                    p00 = *reinterpret_cast<const VS *>(s00 + c); p10 = *reinterpret_cast<const VS *>(s10 + c);
                    p01 = p00; p11 = p10;
                    if (two) { p01 = *reinterpret_cast<const VS *>(s00 + C + c); p11 = *reinterpret_cast<const VS *>(s10 + C + c); }
                }
                VD res;
#pragma unroll
                for (int e = 0; e < VEC; e++) {
                    float q00, q01, q10, q11;
                    if constexpr (VEC == 1) { q00 = (float)p00; q01 = (float)p01; q10 = (float)p10; q11 = (float)p11; }
                    else { q00 = (float)p00[e]; q01 = (float)p01[e]; q10 = (float)p10[e]; q11 = (float)p11[e]; }
                    const float r0 = two ? q00 * a0 + q01 * a1 : q00 * 1.f;
                    const float r1 = two ? q10 * a0 + q11 * a1 : q10 * 1.f;
                    const float v = r0 * b0 + r1 * b1;
                    const TS back = (TS)v;                   // PTD:126 arr_upsampled.astype(arr.dtype)
                    if constexpr (VEC == 1) res = (TD)back;  // PTD:152 .float() (exact) or kept in the file's dtype
                    else res[e] = (TD)back;
                }
                __builtin_nontemporal_store(res, reinterpret_cast<VD *>(o + c));   // written once, read by another kernel later
                if (small) break;                            // one group per k when the window holds the whole row
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Occupancy builder (build_sparse_occupancy.py:30-53).  Two kernels around one 24-byte read-back (the grid's extent
// decides the size of the tensor the caller allocates):
//   k_voxel_coords      coords[i] = rint((pts[i] - origin) / voxel_size) in float32, round half to even (BSO:32;
//                       numpy keeps float32: float32 array op python float), and the min / max per axis (BSO:35,40)
//   k_scatter_occupancy occ[z,y,x] = i + 1 with "the last vertex wins" (BSO:45-46) = largest ID wins = atomicMax
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_voxel_coords(const float *__restrict__ pts, long long N, float ox, float oy, float oz,
                                                      float vs, int *__restrict__ coords, int *minmax, int *bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int c[3] = {0, 0, 0};
    const bool live = i < N;
    if (live) {
        const float q[3] = {(pts[i * 3 + 0] - ox) / vs, (pts[i * 3 + 1] - oy) / vs, (pts[i * 3 + 2] - oz) / vs};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float r = rintf(q[k]);
            if (!(fabsf(r) < 1073741824.0f)) { atomicOr(bad, 1); c[k] = 0; }   // NaN / beyond any grid: reported
            else c[k] = (int)r;
            coords[i * 3 + k] = c[k];
        }
    }
    // min/max per axis: wave-level butterfly, then the workgroup's four waves through LDS -- one atomic pair per workgroup
    // and axis (six hot addresses: one atomic per wave was most of this kernel's 0.1 ms on 87 k points)
    __shared__ int red[4][6];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int lo = live ? c[k] : 2147483647, hi = live ? c[k] : -2147483647 - 1;
        for (int off = 32; off > 0; off >>= 1) {
            lo = min(lo, __shfl_xor(lo, off));
            hi = max(hi, __shfl_xor(hi, off));
        }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][k] = lo; red[threadIdx.x >> 6][3 + k] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        const int lo = min(min(red[0][k], red[1][k]), min(red[2][k], red[3][k]));
        const int hi = max(max(red[0][3 + k], red[1][3 + k]), max(red[2][3 + k], red[3][3 + k]));
        if (lo <= hi) {
            atomicMin(&minmax[k], lo);
            atomicMax(&minmax[3 + k], hi);
        }
    }
}

__global__ __launch_bounds__(256) void k_scatter_occupancy(const int *__restrict__ coords, long long N, int sx, int sy, int sz,
                                                           int dimz, int dimy, int dimx, int *occ, int *bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int x = coords[i * 3 + 0] - sx, y = coords[i * 3 + 1] - sy, z = coords[i * 3 + 2] - sz;
    if ((unsigned)x >= (unsigned)dimx || (unsigned)y >= (unsigned)dimy || (unsigned)z >= (unsigned)dimz) { atomicOr(bad, 1); return; }
    atomicMax(&occ[((long long)z * dimy + y) * dimx + x], (int)(i + 1));
}

}  // namespace
