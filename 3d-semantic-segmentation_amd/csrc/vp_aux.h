// vp_aux.h -- kernels beside the feature path: RGB projection (config 5), Gaussian -> nearest-voxel map (stage 5),
// streaming-read ceiling probe.  Included by voxproj.hip only.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// RGB path (BASELINE config 5): the reference's debug_project_colors.py:54-81 is a per-voxel Python loop --
// voxel-driven, nearest pixel, NO occlusion test, numpy float64 arithmetic.
//
//   k_color_cells     one pass over the dense grid: cell_of_id[id] = cell (IDs are unique per cell when the grid comes
//                     from build_sparse_occupancy.py:44-46; an ID found in two cells raises CST_DUP, an ID outside
//                     [1, n_rows) CST_BADID)
//   k_project_colors  one lane per OCCUPIED voxel (the compact ID list, not the dense grid): the lane walks the views
//                     of the call in order, so the voxel's float32 colour sum is accumulated in view order exactly like
//                     aggregate_voxel_colors_onthefly.py:134-140 does (one contribution per view, no atomics), and
//                     writes the pixel (u, v) it sampled per view (DPC:76, `pixel_indices`).
// Arithmetic contract: oracle_rgb_project in oracle/projector_oracle.c (separate multiplies and adds in float64, IEEE
// divide, round-half-even).  img/255.0 is taken from a 256-entry table of exactly that float64 quotient rounded to
// float32 (DPC:70,75), built in LDS by the workgroup: three correctly-rounded float64 divisions fewer per voxel-view.
// ------------------------------------------------------------------------------------------------
enum { CST_BADID = 0, CST_DUP = 1 };
#ifndef COLOR_UNROLL
#define COLOR_UNROLL 2     // views whose pixel gathers are in flight together per lane of k_project_colors
#endif
constexpr int COLOR_CHUNK = 64;   // views whose pose and intrinsics are staged in LDS together

__global__ __launch_bounds__(256) void k_color_cells(const int *__restrict__ occ, long long cells, int *cell_of_id,
                                                     long long n_rows, int *status)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += stride) {
        const int id = occ[i];
        if (id <= 0) continue;                            // DPC:50 (occ > 0)
        if (id >= n_rows) { atomicOr(&status[CST_BADID], 1); continue; }
        if (atomicMax(&cell_of_id[id], (int)i) >= 0) atomicOr(&status[CST_DUP], 1);
    }
}

__global__ __launch_bounds__(256, 8) void k_project_colors(const int *__restrict__ cell_of_id, int dimy, int dimx,
                                                        const float *__restrict__ c2w, const float *__restrict__ intr,
                                                        int V, float ox, float oy, float oz, double vs,
                                                        const unsigned char *__restrict__ img, int img_h, int img_w,
                                                        float *color_sum, int *hit_count, int *first_view,
                                                        int *pixel_uv, long long n_rows, int view_base)
{
    __shared__ float lut[256];
    // pose and intrinsics of COLOR_CHUNK views at a time, widened to float64 once per workgroup: per view R^T row by row
    // (m0 m4 m8 | m1 m5 m9 | m2 m6 m10), the camera position (m3 m7 m11), fx fy cx cy.  Read as LDS broadcasts; fetched
    // through the scalar unit one element at a time where it was needed -- six dependent scalar-load round trips per view and
    // wavefront -- the kernel spent 55 % of its wave cycles parked (profiles/r05_r4_counters.txt).
    __shared__ double cam[COLOR_CHUNK][16];
    lut[threadIdx.x] = (float)((double)threadIdx.x / 255.0);                                                    // DPC:70,75
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool live = id < n_rows;
    const int cell = (live && id > 0) ? cell_of_id[id] : -1;        // row 0 is the dummy of the 1-based IDs
    if (live && cell < 0) {
        if (pixel_uv)
            for (int v = 0; v < V; v++) {
                pixel_uv[((long long)v * n_rows + id) * 2 + 0] = -1;
                pixel_uv[((long long)v * n_rows + id) * 2 + 1] = -1;
            }
        live = false;
    }
    const int z = live ? cell / (dimy * dimx) : 0;
    const int r = live ? cell - z * (dimy * dimx) : 0;
    const int y = r / dimx, x = r - y * dimx;
    const double wx = (double)ox + vs * (double)x, wy = (double)oy + vs * (double)y, wz = (double)oz + vs * (double)z;   // DPC:60
    const long long img_bytes = (long long)V * img_h * img_w * 3;
    float sr = 0.f, sg = 0.f, sb = 0.f;
    int hc = 0, fv = 0;
    if (live) {
        sr = color_sum[id * 3 + 0]; sg = color_sum[id * 3 + 1]; sb = color_sum[id * 3 + 2];
        hc = hit_count[id];
        fv = first_view ? first_view[id] : 0;
    }
    for (int vc = 0; vc < V; vc += COLOR_CHUNK) {
        __syncthreads();      // the table and the previous chunk are no longer being read
        for (int i = threadIdx.x; i < COLOR_CHUNK * 16; i += 256) {
            const int v = vc + (i >> 4), k = i & 15;
            if (v < V) {
                // k = 0..8: R^T (column c of c2w's rotation as row c), 9..11: position, 12..15: intrinsics
                const int src = k < 9 ? (k % 3) * 4 + k / 3 : (k - 9) * 4 + 3;
                cam[i >> 4][k] = k < 12 ? (double)c2w[(long long)v * 16 + src] : (double)intr[v * 4 + (k - 12)];
            }
        }
        __syncthreads();
        if (!live) continue;
        const int nv = min(COLOR_CHUNK, V - vc);
        // Views are taken COLOR_UNROLL at a time: the pixel addresses of the group are computed first, their loads go out
        // together, and the colours are added afterwards in view order (the float32 sums are the same bits as one view at a time).
        for (int v0 = 0; v0 < nv; v0 += COLOR_UNROLL) {
            int ui[COLOR_UNROLL], vi[COLOR_UNROLL];
#pragma unroll
            for (int j = 0; j < COLOR_UNROLL; j++) {
                ui[j] = -1; vi[j] = -1;
                if (v0 + j < nv) {
                    const double *m = cam[v0 + j];
                    const double dx = wx - m[9], dy = wy - m[10], dz = wz - m[11];                                  // DPC:61-63
                    const double cx = m[0] * dx + m[1] * dy + m[2] * dz;                                            // R^T d
                    const double cy = m[3] * dx + m[4] * dy + m[5] * dz;
                    const double cz = m[6] * dx + m[7] * dy + m[8] * dz;
                    if (cz > 0.0) {                                                                                 // DPC:65
                        const double u = m[12] * (cx / cz) + m[14];                                                 // DPC:66-67
                        const double w = m[13] * (cy / cz) + m[15];
                        const double ur = rint(u), vr = rint(w);                                                    // DPC:68 (half to even)
                        if (ur >= 0.0 && ur < (double)img_w && vr >= 0.0 && vr < (double)img_h) {                  // DPC:69
                            ui[j] = (int)ur; vi[j] = (int)vr;
                        }
                    }
                }
            }
            // the pixel's three bytes with ONE load (an unaligned dword; the fourth byte belongs to the next pixel and is
            // dropped): 64 lanes x 3 byte loads to 64 different cache lines kept the CU's one address unit busy longer than
            // the arithmetic took.  Branch-free, so that the loads of the group go out back to back: a lane without a pixel
            // loads the first dword of the images, and the dword of the very last pixel of the last image starts one byte
            // early (nothing may be read behind the buffer).
            unsigned pix[COLOR_UNROLL];
#pragma unroll
            for (int j = 0; j < COLOR_UNROLL; j++) {
                const long long off = ui[j] >= 0 ? (((long long)(vc + v0 + j) * img_h + vi[j]) * img_w + ui[j]) * 3 : 0;
                if (img_bytes >= 4) {
                    const long long ld = min(off, img_bytes - 4);
                    unsigned w4;
                    __builtin_memcpy(&w4, img + ld, 4);
                    pix[j] = w4 >> (8 * (int)(off - ld));
                } else {      // a single pixel in all: byte by byte
                    pix[j] = (unsigned)img[off] | ((unsigned)img[off + 1] << 8) | ((unsigned)img[off + 2] << 16);
                }
            }
#pragma unroll
            for (int j = 0; j < COLOR_UNROLL; j++) {
                const int v = vc + v0 + j;
                if (ui[j] >= 0) {
                    sr += lut[pix[j] & 255u];                                                                       // AGGC:139
                    sg += lut[(pix[j] >> 8) & 255u];
                    sb += lut[(pix[j] >> 16) & 255u];
                    hc += 1;                                                                                        // AGGC:140
                    fv = min(fv, view_base + v);
                }
                if (pixel_uv && v0 + j < nv) {
                    pixel_uv[((long long)v * n_rows + id) * 2 + 0] = ui[j];
                    pixel_uv[((long long)v * n_rows + id) * 2 + 1] = vi[j];
                }
            }
        }
    }
    if (!live) return;
    color_sum[id * 3 + 0] = sr; color_sum[id * 3 + 1] = sg; color_sum[id * 3 + 2] = sb;
    hit_count[id] = hc;
    if (first_view) first_view[id] = fv;
}

// ------------------------------------------------------------------------------------------------
// Stage-5 front end: nearest voxel of every Gaussian centre.  The reference builds an sklearn
// KDTree over the voxel positions and queries k = 1 (voxel_to_gaussian/voxeltoGaussian_logits.py:87-105, same
// code at voxeltoGaussian.py:84-93); distances there are float64 sums of squared float32 differences.  Here the
// voxel positions are bucketed on a uniform grid (sorted by cell on the host side) and each lane searches
// Chebyshev shells of cells around its query until the best squared distance (same float64 arithmetic) is
// no larger than what any unexplored shell could offer: a point in a cell k shells away is at least (k-1)*h
// away.  Exact; ties go to the lowest voxel index.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nearest_voxel(const float *__restrict__ pts, const int *__restrict__ perm,
                                                       const int *__restrict__ cell_start, double gx, double gy,
                                                       double gz, double h, int nx, int ny, int nz,
                                                       const float *__restrict__ q, long long M, long long *out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double qx = (double)q[i * 3 + 0], qy = (double)q[i * 3 + 1], qz = (double)q[i * 3 + 2];
    // query cell in (possibly out-of-range) grid coordinates
    const double fx = floor((qx - gx) / h), fy = floor((qy - gy) / h), fz = floor((qz - gz) / h);
    const double lim = 1.0e9;
    const long long cx = (long long)fmin(fmax(fx, -lim), lim), cy = (long long)fmin(fmax(fy, -lim), lim),
                    cz = (long long)fmin(fmax(fz, -lim), lim);
    // first shell that can touch the grid
    long long r0 = 0;
    r0 = max(r0, max(-cx, cx - (nx - 1)));
    r0 = max(r0, max(-cy, cy - (ny - 1)));
    r0 = max(r0, max(-cz, cz - (nz - 1)));
    const long long rmax = r0 + (long long)max(nx, max(ny, nz)) + 1;
    double best = INFINITY;
    long long best_idx = -1;
    for (long long r = r0; r <= rmax; r++) {
        const long long z0 = max(cz - r, 0ll), z1 = min(cz + r, (long long)nz - 1);
        const long long y0 = max(cy - r, 0ll), y1 = min(cy + r, (long long)ny - 1);
        const long long x0 = max(cx - r, 0ll), x1 = min(cx + r, (long long)nx - 1);
        for (long long z = z0; z <= z1; z++)
            for (long long y = y0; y <= y1; y++) {
                const bool face = (llabs(z - cz) == r) || (llabs(y - cy) == r);
                for (long long x = x0; x <= x1; x++) {
                    if (!face && llabs(x - cx) != r) {          // interior of the shell: jump to the far side
                        if (x < cx + r && cx + r <= x1) x = cx + r - 1;
                        else break;
                        continue;
                    }
                    const long long c = (z * ny + y) * nx + x;
                    for (int k = cell_start[c]; k < cell_start[c + 1]; k++) {
                        const double dx = qx - (double)pts[(long long)k * 3 + 0];
                        const double dy = qy - (double)pts[(long long)k * 3 + 1];
                        const double dz = qz - (double)pts[(long long)k * 3 + 2];
                        const double d2 = dx * dx + dy * dy + dz * dz;
                        const long long idx = perm[k];
                        if (d2 < best || (d2 == best && idx < best_idx)) { best = d2; best_idx = idx; }
                    }
                }
            }
        // everything in shells > r is at least r*h away
        const double bound = (double)r * h;
        if (best_idx >= 0 && best <= bound * bound) break;
    }
    out[i] = best_idx;
}

// measurement aid (bench.py): plain streaming read of a buffer with 16-byte non-temporal loads, the on-box
// ceiling the gather's achieved bandwidth is quoted against next to the nominal HBM peak
__global__ __launch_bounds__(256) void k_stream_read(const float *__restrict__ src, long long n_vec4, float *sink)
{
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const v4f_ *p = reinterpret_cast<const v4f_ *>(src);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (; i + 3 * stride < n_vec4; i += 4 * stride) {
        const v4f_ x0 = __builtin_nontemporal_load(p + i), x1 = __builtin_nontemporal_load(p + i + stride);
        const v4f_ x2 = __builtin_nontemporal_load(p + i + 2 * stride), x3 = __builtin_nontemporal_load(p + i + 3 * stride);
        a0 += x0.x + x0.y + x0.z + x0.w; a1 += x1.x + x1.y + x1.z + x1.w;
        a2 += x2.x + x2.y + x2.z + x2.w; a3 += x3.x + x3.y + x3.z + x3.w;
    }
    for (; i < n_vec4; i += stride) {
        const v4f_ x0 = __builtin_nontemporal_load(p + i);
        a0 += x0.x + x0.y + x0.z + x0.w;
    }
    const float a = (a0 + a1) + (a2 + a3);
    if (a == 1.2345678e30f) sink[0] = a;   // keeps the loads alive
}

}  // namespace
