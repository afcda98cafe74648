// vp_aux.h -- kernels beside the feature path: RGB projection (config 5), Gaussian -> nearest-voxel map (stage 5),
// streaming-read ceiling probe.  Included by voxproj.hip only.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// RGB path (BASELINE config 5): the reference's debug_project_colors.py:54-81 is a per-voxel Python loop --
// voxel-driven, nearest pixel, NO occlusion test, numpy float64 arithmetic.
//
//   k_color_cells<1>  one pass over the dense grid in Z-ORDER (4x4x4 blocks, one per wavefront load, the blocks along the
//                     Morton curve of their coordinates): cell_of_id[id] = cell (IDs are unique per cell when the grid comes
//                     from build_sparse_occupancy.py:44-46; an ID found in two cells raises CST_DUP, an ID outside
//                     [1, n_rows) CST_BADID) and the number of occupied cells of every wavefront's stretch of the curve
//   k_color_scan      exclusive scan of those counts (one workgroup)
//   k_color_cells<2>  the same walk again: the occupied cells' {ID, cell} appended in curve order -- the voxel list
//   k_project_colors  one lane per entry of that list: the lane walks the views
//                     of the call in order, so the voxel's float32 colour sum is accumulated in view order exactly like
//                     aggregate_voxel_colors_onthefly.py:134-140 does (one contribution per view, no atomics), and
//                     writes the pixel (u, v) it sampled per view (DPC:76, `pixel_indices`).
// Why a list in curve order (round 5): which lane sums which voxel changes no bit of any output, but it decides which image
// lines a wavefront, a workgroup and an XCD touch together.  With the lanes in ID order -- the scan order of the grid,
// BSO:44-46 -- k_project_colors took 2.02 ms per 1000 views of config 5; with the same IDs relabelled along the curve on the
// host 1.47-1.51 ms, and every coarser grouping in between (blocks of 4 / 8 / 16 / 32 cells in scan order: 1.66 / 1.62 / 1.55 /
// 1.49 ms; profiles/r05_ab_colour_order.log): locality pays at every scale, so the list follows the curve itself.
// Arithmetic contract: oracle_rgb_project in oracle/projector_oracle.c (separate multiplies and adds in float64, IEEE
// divide, round-half-even).  img/255.0 is taken from a 256-entry table of exactly that float64 quotient rounded to
// float32 (DPC:70,75), built in LDS by the workgroup: three correctly-rounded float64 divisions fewer per voxel-view.
// ------------------------------------------------------------------------------------------------
enum { CST_BADID = 0, CST_DUP = 1, CST_NOCC = 2 };
#ifndef COLOR_UNROLL
#define COLOR_UNROLL 2     // views per group of k_project_colors: their pixel loads go out together
#endif
constexpr int COLOR_CHUNK = 64;   // views whose pose and intrinsics are staged in LDS together
constexpr int COLOR_WALKERS = 8192;   // wavefronts that share the curve (their counts are one workgroup's scan)

struct ColorCurve {      // the Morton curve over the grid's 4x4x4 blocks: bits per axis (an axis that runs out of bits drops out)
    int bx, by, bz;      // bits of the block coordinates
    int nbx, nby, nbz;   // blocks per axis
    long long slots;     // 2^(bx + by + bz) positions on the curve (those outside the grid are skipped)
};

__device__ __forceinline__ void curve_block(const ColorCurve &c, unsigned s, int &x, int &y, int &z)
{
    x = y = z = 0;
    const int top = max(c.bx, max(c.by, c.bz));
    for (int l = 0; l < top; l++) {
        if (l < c.bx) { x |= (int)(s & 1u) << l; s >>= 1; }
        if (l < c.by) { y |= (int)(s & 1u) << l; s >>= 1; }
        if (l < c.bz) { z |= (int)(s & 1u) << l; s >>= 1; }
    }
}

// PASS 1: cell_of_id, the error flags and counts[w] = occupied cells on wavefront w's stretch of the curve.
// PASS 2: counts[] now holds the exclusive scan; {ID, cell} of the stretch's occupied cells go to list[counts[w] ...].
// A wavefront decodes 64 curve positions at a time (one per lane) and then takes the blocks inside the grid eight at a time,
// lane = cell of the block: eight independent 64-cell loads in flight.
template <int PASS>
__global__ __launch_bounds__(256) void k_color_cells(const int *__restrict__ occ, int dimz, int dimy, int dimx, ColorCurve c,
                                                     int *cell_of_id, long long n_rows, int *status, int *counts, int2 *list)
{
    const int lane = threadIdx.x & 63;
    const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long s0 = c.slots * w / nw, s1 = c.slots * (w + 1) / nw;
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    const long long base = PASS == 2 ? counts[w] : 0;
    int n = 0;
    for (long long sb = s0; sb < s1; sb += 64) {
        int bxv, byv, bzv;
        curve_block(c, (unsigned)(sb + lane), bxv, byv, bzv);
        unsigned long long todo = __ballot(sb + lane < s1 && bxv < c.nbx && byv < c.nby && bzv < c.nbz);
        while (todo) {
            int id[8], cell[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                id[u] = 0; cell[u] = 0;
                if (todo) {
                    const int k = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const int x = __builtin_amdgcn_readlane(bxv, k) * 4 + lx, y = __builtin_amdgcn_readlane(byv, k) * 4 + ly;
                    const int z = __builtin_amdgcn_readlane(bzv, k) * 4 + lz;
                    if (x < dimx && y < dimy && z < dimz) {
                        cell[u] = (z * dimy + y) * dimx + x;
                        id[u] = occ[cell[u]];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                bool ok = id[u] > 0;                                  // DPC:50 (occ > 0)
                if (ok && id[u] >= n_rows) { ok = false; if (PASS == 1) atomicOr(&status[CST_BADID], 1); }
                if (PASS == 1) {
                    if (ok && atomicMax(&cell_of_id[id[u]], cell[u]) >= 0) atomicOr(&status[CST_DUP], 1);
                    n += __popcll(__ballot(ok));
                } else {
                    const unsigned long long m = __ballot(ok);
                    const long long pos = base + n + __popcll(m & ((1ull << lane) - 1ull));
                    if (ok && pos < n_rows) list[pos] = make_int2(id[u], cell[u]);      // pos < n_rows whenever the IDs are unique
                    n += __popcll(m);
                }
            }
        }
    }
    if (PASS == 1 && lane == 0) counts[w] = n;
}

// counts[0 .. n) -> their exclusive scan in place, the total to status[CST_NOCC].  One workgroup of 1024 lanes.
__global__ __launch_bounds__(1024) void k_color_scan(int *counts, int n, int *status)
{
    __shared__ int part[1024];
    const int per = (n + 1023) / 1024, lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += counts[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int i = lo; i < hi; i++) { const int v = counts[i]; counts[i] = run; run += v; }
    if (threadIdx.x == 1023) status[CST_NOCC] = part[1023];
}

// UV: pixel_uv is written (DPC's diagnostics); TINY: the images hold fewer than four bytes in all (one pixel), read byte by byte.
template <bool UV, bool TINY>
__global__ __launch_bounds__(256, 8) void k_project_colors(const int *__restrict__ cell_of_id, const int2 *__restrict__ list,
                                                        const int *__restrict__ status, int dimy, int dimx,
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
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    // rows no cell carries (row 0, the dummy of the 1-based IDs -- SURVEY Q4 -- among them) are seen by no view
    if (UV && t < n_rows && (t == 0 || cell_of_id[t] < 0))
        for (int v = 0; v < V; v++) {
            pixel_uv[((long long)v * n_rows + t) * 2 + 0] = -1;
            pixel_uv[((long long)v * n_rows + t) * 2 + 1] = -1;
        }
    const bool live = t < status[CST_NOCC];      // entry t of the voxel list
    const int2 entry = live ? list[t] : make_int2(0, 0);
    const long long id = entry.x;
    const int cell = entry.y;
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
        const int nv = min(COLOR_CHUNK, V - vc);
        if (!live) continue;
        // Views are taken COLOR_UNROLL at a time and the groups are software-pipelined: while the pixel loads of one group
        // are in flight the lane projects the voxel into the views of the NEXT group (~150 float64 instructions per view in sight,
        // two IEEE divisions among them) and sends their loads; the colours are added afterwards in view order (the float32
        // sums are the same bits as one view at a time).  Two register sets take turns, and the loads of a group go out in
        // straight-line code, so that the wait in front of the sums counts loads instead of draining them (deeper rings were
        // slower: profiles/r05_ab_colour_order.log).
        // A pixel's three bytes come with ONE load (an unaligned dword; the fourth byte belongs to the next pixel and is
        // dropped): 64 lanes x 3 byte loads to 64 different cache lines kept the CU's one address unit busy longer than
        // the arithmetic took.  A lane without a pixel loads the first dword of the images, and the dword of the very last
        // pixel of the last image starts one byte early (nothing may be read behind the buffer).
        // `seen`: bit j = view j of the group sees the voxel; bits 8.. = the byte shift of load j (non-zero for that last pixel only).
        const auto send = [&](int v0, unsigned (&pix)[COLOR_UNROLL]) -> unsigned {
            unsigned seen = 0;
#pragma unroll
            for (int j = 0; j < COLOR_UNROLL; j++) {
                int ui = -1, vi = -1;
                const bool in_chunk = v0 + j < nv;
                const double *m = cam[min(v0 + j, nv - 1)];
                const double dx = wx - m[9], dy = wy - m[10], dz = wz - m[11];                                  // DPC:61-63
                const double cx = m[0] * dx + m[1] * dy + m[2] * dz;                                            // R^T d
                const double cy = m[3] * dx + m[4] * dy + m[5] * dz;
                const double cz = m[6] * dx + m[7] * dy + m[8] * dz;
                if (in_chunk && cz > 0.0) {                                                                   // DPC:65
                    const double u = m[12] * (cx / cz) + m[14];                                                 // DPC:66-67
                    const double w = m[13] * (cy / cz) + m[15];
                    const double ur = rint(u), vr = rint(w);                                                    // DPC:68 (half to even)
                    if (ur >= 0.0 && ur < (double)img_w && vr >= 0.0 && vr < (double)img_h) {                  // DPC:69
                        ui = (int)ur; vi = (int)vr;
                    }
                }
                if (UV && in_chunk) {
                    pixel_uv[((long long)(vc + v0 + j) * n_rows + id) * 2 + 0] = ui;
                    pixel_uv[((long long)(vc + v0 + j) * n_rows + id) * 2 + 1] = vi;
                }
                const long long off = ui >= 0 ? (((long long)(vc + v0 + j) * img_h + vi) * img_w + ui) * 3 : 0;
                if (!TINY) {
                    const long long ld = min(off, img_bytes - 4);
                    __builtin_memcpy(&pix[j], img + ld, 4);          // shifted when it is used: nothing here waits for the load
                    seen |= (unsigned)(off - ld) << (8 + 2 * j);
                } else {
                    pix[j] = (unsigned)img[off] | ((unsigned)img[off + 1] << 8) | ((unsigned)img[off + 2] << 16);
                }
                seen |= (ui >= 0 ? 1u : 0u) << j;
            }
            return seen;
        };
        const auto add = [&](int v0, const unsigned (&pix)[COLOR_UNROLL], unsigned seen) {
#pragma unroll
            for (int j = 0; j < COLOR_UNROLL; j++)
                if (seen >> j & 1u) {
                    const unsigned px = pix[j] >> (8 * (seen >> (8 + 2 * j) & 3u));
                    sr += lut[px & 255u];                                                                           // AGGC:139
                    sg += lut[(px >> 8) & 255u];
                    sb += lut[(px >> 16) & 255u];
                    hc += 1;                                                                                        // AGGC:140
                    fv = min(fv, view_base + vc + v0 + j);
                }
        };
        unsigned pix_a[COLOR_UNROLL], pix_b[COLOR_UNROLL];
        unsigned seen_a = send(0, pix_a), seen_b;
        for (int v0 = 0; v0 < nv; v0 += 2 * COLOR_UNROLL) {          // views past the chunk's end are sent to pixel 0 and dropped
            seen_b = send(v0 + COLOR_UNROLL, pix_b);
            add(v0, pix_a, seen_a);
            seen_a = send(v0 + 2 * COLOR_UNROLL, pix_a);
            add(v0 + COLOR_UNROLL, pix_b, seen_b);
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
