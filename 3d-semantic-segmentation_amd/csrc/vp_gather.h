// vp_gather.h -- phase 2: one wavefront per voxel gathers and sums the feature rows of the pixels that first-hit it
// (k_gather).  A voxel that collected more pixels than the heavy threshold in the call is cut into PARTS (round 5): each
// part is an item of the same work list, summed by one wavefront into a partial row, and a small follow-up kernel
// (k_combine_parts) adds the partial rows to the voxel's row in a fixed order.  One-view calls have a kernel of their
// own (k_gather_one): a fixed grid of wavefronts dealt the parts and the size-ordered list (round 6: parts sized on the device
// from the view's hit total; round 5's path -- a workgroup of four wavefronts per large voxel -- is kept as the A/B arm).
// Included by voxproj.hip only.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// phase 2: one wavefront per voxel
// ------------------------------------------------------------------------------------------------
template <int K, int VEC>
struct Acc {
    float a[K * VEC];
};

// VEC == 8 selects the fp16 feature-map mode (8 halves = 16 B per lane per chunk; values are widened exactly and
// summed in fp32 in the same order, so the outputs equal the fp32 path's on the same data).  Feature pointers are
// carried as `const float *`; this advances one by `elems` ELEMENTS of the mode's input type.
template <int VEC>
__device__ __forceinline__ const float *feat_ptr(const float *base, long long elems)
{
    return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + elems * (VEC == 8 ? 2 : 4));
}

// Scan the pixel box [x0,x1]x[y0,y1] of one view's ID image for pixels whose first hit is `id`, in
// raster order, and add their feature rows (channels cb .. cb+64*K*VEC) to acc.  64 lanes cover a
// tile of tw x (64/tw) pixels, tw = smallest power of two >= box width (capped at 64), so tiles
// and the lanes inside a tile are visited in raster order.
// ID of the pixel this lane covers in the FIRST 64-pixel tile of the box (0 outside the box): the load scan_box would
// issue for that tile, split off so that a caller can have the first tiles of several views in flight at once
__device__ __forceinline__ int first_tile_id(const int *__restrict__ hv, int W, int x0, int y0, int x1, int y1, int lane)
{
    const int bw = x1 - x0 + 1;
    const int lg = bw >= 64 ? 6 : (bw <= 1 ? 0 : 32 - __builtin_clz(bw - 1));
    const int px = x0 + (lane & ((1 << lg) - 1)), py = y0 + (lane >> lg);
    return ((px <= x1) && (py <= y1)) ? hv[py * W + px] : 0;
}

template <int K, int VEC, int U>
__device__ __forceinline__ void scan_box(const float *__restrict__ fv, const int *__restrict__ hv,
                                         int W, int C, int id, int x0, int y0, int x1, int y1,
                                         int cb, int lane, Acc<K, VEC> &acc, int &found,
                                         bool have_first = false, int h_first = 0)
{
    const int bw = x1 - x0 + 1;
    const int lg = bw >= 64 ? 6 : (bw <= 1 ? 0 : 32 - __builtin_clz(bw - 1));
    const int tw = 1 << lg, th = 64 >> lg;
    const int lx = lane & (tw - 1), ly = lane >> lg;
    for (int ty = y0; ty <= y1; ty += th) {
        const int py = ty + ly;
        for (int tx = x0; tx <= x1; tx += tw) {
            const int px = tx + lx;
            const bool inb = (px <= x1) && (py <= y1);
            const int pix = py * W + px;
            int h;
            if (have_first && ty == y0 && tx == x0) h = h_first;     // prefetched by the caller (first_tile_id)
            else h = inb ? hv[pix] : 0;
            unsigned long long m = __ballot(h == id);
            found += __popcll(m);
            while (m) {
                int n = 0;
                long long off[U];
#pragma unroll
                for (int j = 0; j < U; j++) {
                    off[j] = 0;
                    if (m) {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        off[j] = (long long)__builtin_amdgcn_readlane(pix, l) * C + cb;
                        n = j + 1;
                    }
                }
                if constexpr (VEC == 8) {
                    typedef _Float16 v8h_ __attribute__((ext_vector_type(8)));
                    v8h_ r[U][K];
#pragma unroll
                    for (int j = 0; j < U; j++)
                        if (j < n) {
#pragma unroll
                            for (int k = 0; k < K; k++) {
                                const int ch = (k * 64 + lane) * 8;
                                if (cb + ch < C)
                                    r[j][k] = __builtin_nontemporal_load(reinterpret_cast<const v8h_ *>(
                                        reinterpret_cast<const char *>(fv) + (off[j] + ch) * 2));
                                else
                                    r[j][k] = (v8h_)(_Float16)0;
                            }
                        }
#pragma unroll
                    for (int j = 0; j < U; j++)
                        if (j < n) {
#pragma unroll
                            for (int k = 0; k < K; k++)
#pragma unroll
                                for (int e = 0; e < 8; e++) acc.a[k * 8 + e] += (float)r[j][k][e];
                        }
                } else if constexpr (VEC == 4) {
                    float4 r[U][K];
#pragma unroll
                    for (int j = 0; j < U; j++)
                        if (j < n) {
#pragma unroll
                            for (int k = 0; k < K; k++) {
                                const int ch = (k * 64 + lane) * 4;
                                // feature rows are read exactly once: non-temporal loads keep them out of L2/MALL
                                // (+12 % gather bandwidth measured against plain loads)
                                typedef float v4f_ __attribute__((ext_vector_type(4)));
                                if (cb + ch < C) {
                                    const v4f_ t_ = __builtin_nontemporal_load(reinterpret_cast<const v4f_ *>(fv + off[j] + ch));
                                    r[j][k] = make_float4(t_.x, t_.y, t_.z, t_.w);
                                } else {
                                    r[j][k] = make_float4(0.f, 0.f, 0.f, 0.f);
                                }
                            }
                        }
#pragma unroll
                    for (int j = 0; j < U; j++)
                        if (j < n) {
#pragma unroll
                            for (int k = 0; k < K; k++) {
                                acc.a[k * 4 + 0] += r[j][k].x;
                                acc.a[k * 4 + 1] += r[j][k].y;
                                acc.a[k * 4 + 2] += r[j][k].z;
                                acc.a[k * 4 + 3] += r[j][k].w;
                            }
                        }
                } else {
                    float r[U][K];
#pragma unroll
                    for (int j = 0; j < U; j++)
                        if (j < n) {
#pragma unroll
                            for (int k = 0; k < K; k++) {
                                const int ch = k * 64 + lane;
                                r[j][k] = (cb + ch < C) ? fv[off[j] + ch] : 0.f;
                            }
                        }
#pragma unroll
                    for (int j = 0; j < U; j++)
                        if (j < n) {
#pragma unroll
                            for (int k = 0; k < K; k++) acc.a[k] += r[j][k];
                        }
                }
            }
        }
    }
}

// Conservative pixel box of a voxel cube (centre c, half edge h) in view ve; returns false if empty.
// Every ray sample has camera depth >= depthMin (t >= depthMin/camDir.z, K.cu:31-32), so the cube is
// clipped against the plane z = zn = 0.98*depthMin before it is projected: vertices in front of the
// plane are projected as they are, edges crossing it contribute their intersection point.
__device__ __forceinline__ bool voxel_box(const ViewEntry &ve, float fx, float fy, float mx, float my,
                                          float cxw, float cyw, float czw, float h, float zn, int W, int H,
                                          int &x0, int &y0, int &x1, int &y1)
{
    x0 = 0; y0 = 0; x1 = W - 1; y1 = H - 1;
    if (ve.ok == 0.0f) return true;
    const float dx = cxw - ve.pos[0], dy = cyw - ve.pos[1], dz = czw - ve.pos[2];
    const float camx = ve.inv[0] * dx + ve.inv[1] * dy + ve.inv[2] * dz;
    const float camy = ve.inv[3] * dx + ve.inv[4] * dy + ve.inv[5] * dz;
    const float camz = ve.inv[6] * dx + ve.inv[7] * dy + ve.inv[8] * dz;
    const float ez = h * (fabsf(ve.inv[6]) + fabsf(ve.inv[7]) + fabsf(ve.inv[8]));
    if (!(camz + ez > zn)) return false;                   // cube entirely nearer than any sample
    float umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
    float qx[8], qy[8], qz[8];
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const float a = (s & 1) ? h : -h, b = (s & 2) ? h : -h, c = (s & 4) ? h : -h;
        qx[s] = camx + ve.inv[0] * a + ve.inv[1] * b + ve.inv[2] * c;
        qy[s] = camy + ve.inv[3] * a + ve.inv[4] * b + ve.inv[5] * c;
        qz[s] = camz + ve.inv[6] * a + ve.inv[7] * b + ve.inv[8] * c;
        if (qz[s] >= zn) {
            const float u = fx * (qx[s] / qz[s]) + mx, v = fy * (qy[s] / qz[s]) + my;
            umin = fminf(umin, u); umax = fmaxf(umax, u);
            vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
        }
    }
    if (!(camz - ez >= zn)) {
        // some vertices are behind the plane: add the 12 edges' crossings with z = zn
#pragma unroll
        for (int s = 0; s < 8; s++) {
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
                const int o = s ^ (1 << ax);
                if (o < s) continue;
                const bool fs = qz[s] >= zn, fo = qz[o] >= zn;
                if (fs == fo) continue;
                const float tt = (zn - qz[s]) / (qz[o] - qz[s]);
                const float ix = qx[s] + tt * (qx[o] - qx[s]), iy = qy[s] + tt * (qy[o] - qy[s]);
                const float u = fx * (ix / zn) + mx, v = fy * (iy / zn) + my;
                umin = fminf(umin, u); umax = fmaxf(umax, u);
                vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
            }
        }
        // crossing points are computed with cancellation: widen by 2 % of the box and 2 px
        const float pu = 0.02f * (umax - umin) + 2.0f, pv = 0.02f * (vmax - vmin) + 2.0f;
        umin -= pu; umax += pu; vmin -= pv; vmax += pv;
    }
    if (!(umin == umin) || !(umax == umax) || !(vmin == vmin) || !(vmax == vmax)) return true;
    if (!(umin <= umax) || !(vmin <= vmax)) return true;   // nothing in front although the depth test passed
    const float fW = (float)W, fH = (float)H;
    if (umax < -2.0f || vmax < -2.0f || umin > fW + 1.0f || vmin > fH + 1.0f) return false;
    x0 = max(0, (int)floorf(fmaxf(umin, 0.0f)) - 1);
    y0 = max(0, (int)floorf(fmaxf(vmin, 0.0f)) - 1);
    x1 = min(W - 1, (int)ceilf(fminf(umax, fW)) + 1);
    y1 = min(H - 1, (int)ceilf(fminf(vmax, fH)) + 1);
    return x0 <= x1 && y0 <= y1;
}

struct GatherArgs {
    const float *feats;
    const int *hit;
    const ViewEntry *viewtab;
    const float *intr;
    const int *cell_of_id;
    const int *cnt_call;
    const int *heavy_list;   // one-view calls: IDs whose pixel count exceeds heavy_t (appended by phase 1)
    const int *n_heavy;
    int parts_on;            // k_gather_one: the call's work list has part items (one-view calls that split large voxels)
    int *host_word;          // nullable (blocking one-view calls): pinned host word that receives "split voxels of this call" when the
    int host_seq;            // gather starts, tagged with the call's sequence number -- the host then launches k_combine_parts only if any
    int slot_cap;            // part slots of the buffer set: the consumers never walk past it, whatever the counters say
    int row_lo, row_hi;      // phase 2 of this launch covers the voxel IDs in [row_lo, row_hi) (VP_OPT_ROW_BEGIN / _END)
    int heavy_blocks;        // k_gather_one, VP_OPT_ONE_VIEW_SPLIT = 0 only: workgroups that take the march's heavy voxels before they join the deal
    const int *work;         // work list of this call: WORK_CLASSES arrays of n_rows voxel IDs, by size class (k_worklist)
    const int *work_n;       // voxels per class; [WORK_CLASSES] = parts planned, [WORK_CLASSES + 1] = split voxels (ST_NPARTS, ST_NSPLIT)
    const int4 *parts;       // part items of the split voxels: {id, part, P, first slot of the voxel}; the item's index is its slot
    const int4 *split;       // split voxels: {id, first slot, P, pixels in the call}
    int4 *pmeta;             // per slot: {pixels found, views that contributed, first such view, last such view}
    float *prow;             // per slot: the part's C-wide partial row
    int *count;
    int *views_hit;          // nullable: += number of views of this call in which the voxel got >= 1 pixel
    float *out;
    int *status;
};

constexpr int GW_MERGED = 4;     // wavefronts of a workgroup that share one voxel: k_combine_parts, its whole-image redo, and the A/B arm of k_gather_one
constexpr int COMBINE_BLOCKS = 512; // grid of k_combine_parts (a workgroup per split voxel at a time)
// One-view calls size their parts on the device, from the number of pixels the view's rays hit (k_worklist): the smallest part,
// and how many parts' worth of pixels a voxel must exceed to be cut (VP_OPT_PART_PIXELS / VP_OPT_ONE_VIEW_SPLIT fix them)
#ifndef VP_ONE_VIEW_PART_MIN
#define VP_ONE_VIEW_PART_MIN 32
#endif
#ifndef VP_ONE_VIEW_T_RATIO
#define VP_ONE_VIEW_T_RATIO 2
#endif
#ifndef VP_ONE_VIEW_T_FLOOR_SMALL
#define VP_ONE_VIEW_T_FLOOR_SMALL 256
#endif
constexpr int ONE_VIEW_PART_MIN = VP_ONE_VIEW_PART_MIN, ONE_VIEW_T_RATIO = VP_ONE_VIEW_T_RATIO, ONE_VIEW_T_FLOOR_SMALL = VP_ONE_VIEW_T_FLOOR_SMALL;

// What k_worklist plans with.  Calls of more than one view: heavy_t == part_t and part_px are the host's (project_impl).
// One-view calls that split (round 6): no voxel is shared by a workgroup any more, heavy_t == part_t again, and both numbers may
// be left to the device: dyn_px_min > 0 -> part_px = max(dyn_px_min, 2 * hits / slots), with `hits` the pixels of the view whose
// ray hit a voxel (counted by the march, ST_NHIT) -- about one part per wavefront the machine holds on a frame that is all large
// voxels, parts of 32 pixels on a frame that is mostly misses, where the longest single item IS the launch (a wavefront alone
// pulls ~5 GB/s: 320 rows of 2 KiB last 128 us); dyn_t_ratio > 0 -> part_t = heavy_t = max(dyn_t_ratio * part_px, dyn_t_floor): a
// voxel is worth cutting only when one wavefront would need a good part of the launch's duration for it -- ~26 us are 64 pixels
// with 4 rows in flight per wavefront, and views of up to 262144 pixels (8 rows in flight; a quarter-resolution frame is ten
// thousand voxels of a dozen pixels, bounded by round trips per voxel, not by its longest voxel) gain nothing below 256.
struct PlanArgs {
    int heavy_t, part_t, part_px;
    int count_heavy;      // add the split voxels to ST_NHEAVY (0: the march counts the voxels above heavy_t, one-view calls without parts)
    int dyn_px_min, dyn_t_ratio, dyn_t_floor;
    int cell_in_item;     // parts[].w = the voxel's cell in batch 0 instead of its first slot (one-view calls: B == 1)
};

// Views whose first ID tile is fetched together by the one-wavefront gather (template argument G of k_gather; 1 = one view
// at a time).  fp16 rows: 4 (-1 % pipelined, round 2).  fp32 rows: 4 for small images (a voxel of R1's 484x274 views gathers
// half the rows per view an R2 voxel does, so the dependent tile fetch in front of them weighs twice as much: -1.4 % per
// pipelined R1 call), 1 otherwise (968x548: equal alone, +0.3 % pipelined) -- profiles/r03_ab_id_tile_grouping_fp32.log.
constexpr int GATHER_G16 = 4;
constexpr long long GATHER_G32_SMALL_IMAGE = 262144;     // pixels per view up to which fp32 calls use G = 4

// Output rows are read once and written once per call by the wavefront that owns the voxel: no reuse inside a launch.
// The finished row is stored WRITE-THROUGH (buffer_store_dwordx4 with the sc0 sc1 cache-policy bits): stores that leave
// dirty lines in L2 cost this read-bound kernel far more than their bytes, and on some (feature pool, output rows)
// placement pairs several times more (profiles/r03_probe_row_stores_microbenchmark.log: +3.9 % per launch for one plain 2-KiB store per 272 rows
// read, +12.7 % on a bad pair; +2.3 % / +4.6 % written through; the same through uncached memory or a non-temporal store).
// In the gather itself: -0.3 ... -1.5 % per launch (profiles/r03_ab_output_row_store_policy.log).  The row's load stays a
// plain load (a non-temporal load of the row was measured slower).  Visibility is that of a plain store: the line goes
// to memory at system scope and the next kernel reads it from there.
typedef int v4i_ __attribute__((ext_vector_type(4)));
constexpr int OUT_ST_POLICY = 0x11;     // aux bits of the raw buffer store on gfx940+: sc0 (bit 0) | sc1 (bit 4)

__device__ __forceinline__ float4 ld_out4(const float *p)
{
    return *reinterpret_cast<const float4 *>(p);
}

// store 16 bytes at row + byte_off, row wave-uniform (one buffer descriptor per row chunk: base = the row, no bound)
__device__ __forceinline__ void st_out4(__amdgpu_buffer_rsrc_t row, int byte_off, float4 v)
{
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    const v4f_ t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_, t), row, byte_off, 0, OUT_ST_POLICY);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_row_rsrc(float *orow)
{
    // raw buffer, stride 0, num_records = 2^31 bytes (the callers bound their accesses themselves), gfx9 dword 3 for 32-bit data
    return __builtin_amdgcn_make_buffer_rsrc(orow, 0, 0x7fffffff, 0x00020000);
}

template <int K, int VEC>
__device__ __forceinline__ void acc_load(Acc<K, VEC> &acc, const float *orow, int cb, int C, int lane)
{
#pragma unroll
    for (int k = 0; k < K; k++) {
        if constexpr (VEC == 8) {
            const int ch = (k * 64 + lane) * 8;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float4 o = (cb + ch < C) ? ld_out4(orow + ch + h * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc.a[k * 8 + h * 4 + 0] = o.x; acc.a[k * 8 + h * 4 + 1] = o.y; acc.a[k * 8 + h * 4 + 2] = o.z; acc.a[k * 8 + h * 4 + 3] = o.w;
            }
        } else if constexpr (VEC == 4) {
            const int ch = (k * 64 + lane) * 4;
            const float4 o = (cb + ch < C) ? ld_out4(orow + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc.a[k * 4 + 0] = o.x; acc.a[k * 4 + 1] = o.y; acc.a[k * 4 + 2] = o.z; acc.a[k * 4 + 3] = o.w;
        } else {
            const int ch = k * 64 + lane;
            acc.a[k] = (cb + ch < C) ? orow[ch] : 0.f;
        }
    }
}

// WT: write-through stores (the multi-view launches); false: plain stores -- the one-view-per-call variant of k_gather stays
// at 96 VGPRs = 5 wavefronts per SIMD that way (the buffer form costs 5 registers), which is worth more to a 0.25-ms launch.
template <int K, int VEC, bool WT>
__device__ __forceinline__ void acc_store(const Acc<K, VEC> &acc, float *orow, int cb, int C, int lane)
{
    if constexpr (!WT) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            if constexpr (VEC == 8) {
                const int ch = (k * 64 + lane) * 8;
                if (cb + ch < C) {
#pragma unroll
                    for (int h = 0; h < 2; h++)
                        *reinterpret_cast<float4 *>(orow + ch + h * 4) = make_float4(acc.a[k * 8 + h * 4 + 0], acc.a[k * 8 + h * 4 + 1], acc.a[k * 8 + h * 4 + 2], acc.a[k * 8 + h * 4 + 3]);
                }
            } else if constexpr (VEC == 4) {
                const int ch = (k * 64 + lane) * 4;
                if (cb + ch < C)
                    *reinterpret_cast<float4 *>(orow + ch) = make_float4(acc.a[k * 4 + 0], acc.a[k * 4 + 1], acc.a[k * 4 + 2], acc.a[k * 4 + 3]);
            } else {
                const int ch = k * 64 + lane;
                if (cb + ch < C) orow[ch] = acc.a[k];
            }
        }
        return;
    }
    const __amdgpu_buffer_rsrc_t row = out_row_rsrc(orow);
#pragma unroll
    for (int k = 0; k < K; k++) {
        if constexpr (VEC == 8) {
            const int ch = (k * 64 + lane) * 8;
            if (cb + ch < C) {
#pragma unroll
                for (int h = 0; h < 2; h++)
                    st_out4(row, (ch + h * 4) * 4, make_float4(acc.a[k * 8 + h * 4 + 0], acc.a[k * 8 + h * 4 + 1], acc.a[k * 8 + h * 4 + 2], acc.a[k * 8 + h * 4 + 3]));
            }
        } else if constexpr (VEC == 4) {
            const int ch = (k * 64 + lane) * 4;
            if (cb + ch < C)
                st_out4(row, ch * 4, make_float4(acc.a[k * 4 + 0], acc.a[k * 4 + 1], acc.a[k * 4 + 2], acc.a[k * 4 + 3]));
        } else {
            const int ch = k * 64 + lane;
            if (cb + ch < C) orow[ch] = acc.a[k];
        }
    }
}

// world-space centre of voxel `id` in batch b; false if the grid of batch b does not hold the ID
__device__ __forceinline__ bool voxel_centre(const GatherArgs &g, const Params &p, int b, int id,
                                             float &cxw, float &cyw, float &czw)
{
    const int cell = g.cell_of_id[(long long)b * p.n_rows + id];
    if (cell < 0) return false;
    const int czi = cell / (p.dimy * p.dimx);
    const int rem = cell - czi * (p.dimy * p.dimx);
    const int cyi = rem / p.dimx, cxi = rem - cyi * p.dimx;
    cxw = p.ox + (float)cxi * p.vs; cyw = p.oy + (float)cyi * p.vs; czw = p.oz + (float)czi * p.vs;
    return true;
}

__device__ __forceinline__ float box_half_edge(const Params &p)
{
    return 0.5f * fabsf(p.vs) * 1.02f +
           1e-6f * (fabsf(p.ox) + fabsf(p.oy) + fabsf(p.oz) + fabsf(p.vs) * (p.dimx + p.dimy + p.dimz));
}

// camera depth below which no ray sample exists (t starts at depthMin/camDir.z); a non-positive or
// non-finite depthMin degrades to a tiny positive plane (boxes grow, results stay exact)
__device__ __forceinline__ float near_plane(const Params &p)
{
    const float zn = 0.98f * p.dmin;
    return (zn > 1e-6f && zn < 1e30f) ? zn : 1e-6f;
}

// Normal role: one wavefront sums all pixels of one voxel, in (b, v, y, x) order, starting from the
// row already in `out` -- bit-identical to the oracle's serial accumulation.
template <int K, int VEC, int U, int G, bool WT>
__device__ __forceinline__ void gather_voxel_wave(const GatherArgs &g, const Params &p, int id, int expected, int lane)
{
    const int W = p.width, H = p.height, C = p.C;
    const long long HW = (long long)H * W;
    const float hh = box_half_edge(p);
    const float zn = near_plane(p);
    constexpr int CB = 64 * K * VEC;
    for (int cb = 0; cb < C; cb += CB) {
        Acc<K, VEC> acc;
        float *orow = g.out + (long long)id * C + cb;
        acc_load<K, VEC>(acc, orow, cb, C, lane);
        const Acc<K, VEC> acc0 = acc;
        int found = 0, nviews = 0;
        for (int b = 0; b < p.B && found < expected; b++) {
            float cxw, cyw, czw;
            if (!voxel_centre(g, p, b, id, cxw, cyw, czw)) continue;
            const float fx = g.intr[b * 4 + 0], fy = g.intr[b * 4 + 1], mx = g.intr[b * 4 + 2], my = g.intr[b * 4 + 3];
            for (int vbase = 0; vbase < p.V && found < expected; vbase += 64) {
                const int v = vbase + lane;
                int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
                bool ne = false;
                if (v < p.V) ne = voxel_box(g.viewtab[b * p.V + v], fx, fy, mx, my, cxw, cyw, czw, hh, zn, W, H, x0, y0, x1, y1);
                unsigned long long vm = __ballot(ne);
                // Views are taken G at a time: the ID-image loads of the first tile of all G views go out together (one
                // memory latency instead of G dependent ones -- most boxes are a single 64-pixel tile, and many views of
                // a voxel yield no pixel at all), then the views are summed one after the other, in order.
                while (vm && found < expected) {
                    // the group: the lowest G set bits of vm
                    unsigned long long mg = 0ull;
#pragma unroll
                    for (int q = 0; q < G; q++)
                        if (vm) { mg |= vm & (~vm + 1ull); vm &= vm - 1; }
                    int gh[G];
                    if constexpr (G > 1) {
                        unsigned long long t = mg;
#pragma unroll
                        for (int q = 0; q < G; q++) {
                            gh[q] = 0;
                            if (t) {
                                const int l = __builtin_ctzll(t);
                                t &= t - 1;
                                gh[q] = first_tile_id(g.hit + ((long long)b * p.V + vbase + l) * HW, W, __builtin_amdgcn_readlane(x0, l),
                                                      __builtin_amdgcn_readlane(y0, l), __builtin_amdgcn_readlane(x1, l),
                                                      __builtin_amdgcn_readlane(y1, l), lane);
                            }
                        }
                    }
                    int q = 0;
#pragma unroll 1
                    for (unsigned long long t = mg; t && found < expected; q++) {
                        const int l = __builtin_ctzll(t);
                        t &= t - 1;
                        const int bx0 = __builtin_amdgcn_readlane(x0, l), by0 = __builtin_amdgcn_readlane(y0, l);
                        const int bx1 = __builtin_amdgcn_readlane(x1, l), by1 = __builtin_amdgcn_readlane(y1, l);
                        int hq = 0;
                        if constexpr (G > 1) {
                            hq = gh[0];
#pragma unroll
                            for (int k = 1; k < G; k++) hq = (q == k) ? gh[k] : hq;
                        }
                        const long long bv = (long long)b * p.V + vbase + l;
                        const int before = found;
                        scan_box<K, VEC, U>(feat_ptr<VEC>(g.feats, bv * HW * C), g.hit + bv * HW, W, C, id, bx0, by0, bx1, by1, cb, lane,
                                            acc, found, G > 1, hq);
                        nviews += found > before;
                    }
                }
            }
        }
        if (found != expected) {
            // the search boxes missed pixels (an ID labelling several cells, a degenerate pose...):
            // redo this voxel over whole images.  Correctness never depends on the boxes.
            if (lane == 0 && cb == 0) atomicAdd(&g.status[ST_BOXMISS], 1);
            acc = acc0;
            found = 0;
            nviews = 0;
            for (long long bv = 0; bv < (long long)p.B * p.V; bv++) {
                const int before = found;
                scan_box<K, VEC, U>(feat_ptr<VEC>(g.feats, bv * HW * C), g.hit + bv * HW, W, C, id, 0, 0, W - 1, H - 1, cb, lane, acc, found);
                nviews += found > before;
            }
        }
        acc_store<K, VEC, WT>(acc, orow, cb, C, lane);
        if (cb == 0 && lane == 0) {
            g.count[id] += found;   // K.cu:77 (one add of the per-call total)
            if (g.views_hit) g.views_hit[id] += nviews;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Split voxels (round 5).  A voxel that collected more than heavy_t pixels in the call -- a surface patch a hand-held camera
// stares at from 0.3 m for sixty consecutive frames collects 10^5 and more -- is not one wavefront's job: at the ~5 GB/s one
// wavefront pulls, 200 MB of rows last longer than the whole launch.  Rounds 1-4 gave such a voxel to ONE workgroup of the
// launch's first 128 (four wavefronts splitting each view's box rows, combined through LDS view by view): enough for the
// benign room (76 voxels of <= 8.8 k pixels per pass), 0.25-0.27 of peak on a trajectory whose close-ups put 70-100 % of a
// call's pixels into such voxels (profiles/r05_before_*.log).  Now:
//   * k_worklist PLANS: a voxel with c > heavy_t pixels becomes P = ceil(c / part_px) parts; each part is an item
//     {id, part, P} in its own slot, and the parts lead the work list (they are the longest items);
//   * a PART is a contiguous piece of the voxel's pixel sequence in (b, v, y, x) order, cut by search-box AREA: with
//     A = the summed area of the voxel's pixel boxes over all views of the call, part k owns the box rows whose first
//     pixel's running area index lies in [A k / P, A (k+1) / P) -- every row of every view belongs to exactly one part,
//     whatever the boxes are.  One wavefront sums its rows in order from zero into a partial row (gather_part_wave);
//   * k_combine_parts (one workgroup per split voxel, after the gather on the same stream) adds the partial rows to the
//     row in `out` in slot order -- a fixed tree, so results are reproducible run to run and independent of which
//     wavefront ran when; they differ from the serial order in the last bits only (inside the 1e-4 bar, tested per
//     element); pixel counts add exactly; a view shared by two neighbouring parts is counted once.  If the parts found
//     fewer pixels than the march counted (the boxes are only hints), the voxel is redone over whole images.
// No float atomics, no inter-workgroup hand-off inside a launch (the partial rows cross a kernel boundary).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long wave_sum_nonneg(int v)
{
    long long s = 0;
    unsigned long long m = __ballot(v != 0);
    while (m) {
        const int l = __builtin_ctzll(m);
        m &= m - 1;
        s += __builtin_amdgcn_readlane(v, l);
    }
    return s;
}

template <int K, int VEC, int U>
__device__ __forceinline__ void gather_part_wave(const GatherArgs &g, const Params &p, int slot, int lane)
{
    const int4 it = g.parts[slot];
    const int id = it.x, part = it.y, P = it.z;
    const int W = p.width, H = p.height, C = p.C;
    const long long HW = (long long)H * W;
    const float hh = box_half_edge(p);
    const float zn = near_plane(p);
    constexpr int CB = 64 * K * VEC;
    // pass 1: the voxel's total box area over the call's views
    long long A = 0;
    for (int b = 0; b < p.B; b++) {
        float cxw, cyw, czw;
        if (!voxel_centre(g, p, b, id, cxw, cyw, czw)) continue;
        const float fx = g.intr[b * 4 + 0], fy = g.intr[b * 4 + 1], mx = g.intr[b * 4 + 2], my = g.intr[b * 4 + 3];
        for (int vbase = 0; vbase < p.V; vbase += 64) {
            const int v = vbase + lane;
            int x0 = 0, y0 = 0, x1 = -1, y1 = -1, area = 0;
            if (v < p.V && voxel_box(g.viewtab[b * p.V + v], fx, fy, mx, my, cxw, cyw, czw, hh, zn, W, H, x0, y0, x1, y1))
                area = (x1 - x0 + 1) * (y1 - y0 + 1);
            A += wave_sum_nonneg(area);
        }
    }
    const long long lo = A * part / P, hi = A * (part + 1) / P;      // A < 2^47, P <= 2^15
    int found0 = 0, nviews = 0, first_v = -1, last_v = -1;
    for (int cb = 0; cb < C; cb += CB) {
        Acc<K, VEC> acc;
#pragma unroll
        for (int i = 0; i < K * VEC; i++) acc.a[i] = 0.f;
        int found = 0;
        long long a0 = 0;      // area of the boxes in front of the current view
        for (int b = 0; b < p.B && a0 < hi; b++) {
            float cxw, cyw, czw;
            if (!voxel_centre(g, p, b, id, cxw, cyw, czw)) continue;
            const float fx = g.intr[b * 4 + 0], fy = g.intr[b * 4 + 1], mx = g.intr[b * 4 + 2], my = g.intr[b * 4 + 3];
            for (int vbase = 0; vbase < p.V && a0 < hi; vbase += 64) {
                const int v = vbase + lane;
                int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
                bool ne = false;
                if (v < p.V) ne = voxel_box(g.viewtab[b * p.V + v], fx, fy, mx, my, cxw, cyw, czw, hh, zn, W, H, x0, y0, x1, y1);
                unsigned long long vm = __ballot(ne);
                while (vm && a0 < hi) {
                    const int l = __builtin_ctzll(vm);
                    vm &= vm - 1;
                    const int bx0 = __builtin_amdgcn_readlane(x0, l), by0 = __builtin_amdgcn_readlane(y0, l);
                    const int bx1 = __builtin_amdgcn_readlane(x1, l), by1 = __builtin_amdgcn_readlane(y1, l);
                    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
                    const long long av = (long long)bw * bh;
                    // rows r of this box with lo <= a0 + r*bw < hi.  A box that ends before lo owns none of them -- most views of a
                    // part in the middle of a long voxel: no division for those; the others divide 32-bit numbers (lo - a0, hi - a0
                    // < av < 2^31; two 64-bit divisions per view and part were a third of a part's instructions)
                    const long long a1 = a0 + av;
                    const long long b0 = a0;
                    a0 = a1;
                    if (a1 <= lo) continue;
                    const int r_lo = lo > b0 ? (int)(((unsigned)(lo - b0) + (unsigned)bw - 1u) / (unsigned)bw) : 0;
                    const int r_hi = hi < a1 ? (int)(((unsigned)(hi - b0) + (unsigned)bw - 1u) / (unsigned)bw) : bh;
                    if (r_lo >= r_hi) continue;
                    const long long bv = (long long)b * p.V + vbase + l;
                    const int before = found;
                    scan_box<K, VEC, U>(feat_ptr<VEC>(g.feats, bv * HW * C), g.hit + bv * HW, W, C, id, bx0, by0 + r_lo, bx1,
                                        by0 + r_hi - 1, cb, lane, acc, found);
                    if (cb == 0 && found > before) {
                        nviews++;
                        if (first_v < 0) first_v = (int)bv;
                        last_v = (int)bv;
                    }
                }
            }
        }
        acc_store<K, VEC, false>(acc, g.prow + (long long)slot * C + cb, cb, C, lane);
        if (cb == 0) found0 = found;
    }
    if (lane == 0) g.pmeta[slot] = make_int4(found0, nviews, first_v, last_v);
}

// A part of a ONE-VIEW call (B*V == 1): part k of P owns the rows [ceil(h k / P), ceil(h (k+1) / P)) of the voxel's pixel box of h
// rows -- consecutive, disjoint, all of them, whatever P (more parts than rows: some own none) --, with the box computed once, by
// every lane alike, from the cell k_worklist left in the item (one dependent load fewer in front of the first row; 32-bit
// arithmetic: h k < 2^31 for any image the library accepts with P <= 8192 slots).  No cell or no box: the part finds nothing and
// k_combine_parts redoes the voxel over the whole image.
template <int K, int VEC, int U>
__device__ __forceinline__ void gather_part_one(const GatherArgs &g, const Params &p, int slot, int lane)
{
    const int4 it = g.parts[slot];
    const int id = it.x, part = it.y, P = it.z, cell = it.w;
    const int W = p.width, H = p.height, C = p.C;
    constexpr int CB = 64 * K * VEC;
    int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
    if (cell >= 0) {
        const int czi = cell / (p.dimy * p.dimx);
        const int rem = cell - czi * (p.dimy * p.dimx);
        const int cyi = rem / p.dimx, cxi = rem - cyi * p.dimx;
        int a0, a1, a2, a3;
        if (voxel_box(g.viewtab[0], g.intr[0], g.intr[1], g.intr[2], g.intr[3], p.ox + (float)cxi * p.vs, p.oy + (float)cyi * p.vs,
                      p.oz + (float)czi * p.vs, box_half_edge(p), near_plane(p), W, H, a0, a1, a2, a3)) { x0 = a0; y0 = a1; x1 = a2; y1 = a3; }
    }
    const int bh = y1 - y0 + 1;
    int r_lo = 0, r_hi = 0;
    if (bh > 0 && x1 >= x0) {
        const unsigned up = (unsigned)P;
        r_lo = (int)(((unsigned long long)(unsigned)bh * (unsigned)part + up - 1u) / up);
        r_hi = (int)(((unsigned long long)(unsigned)bh * ((unsigned)part + 1u) + up - 1u) / up);
    }
    int found0 = 0;
    for (int cb = 0; cb < C; cb += CB) {
        Acc<K, VEC> acc;
#pragma unroll
        for (int i = 0; i < K * VEC; i++) acc.a[i] = 0.f;
        int found = 0;
        if (r_lo < r_hi) scan_box<K, VEC, U>(g.feats, g.hit, W, C, id, x0, y0 + r_lo, x1, y0 + r_hi - 1, cb, lane, acc, found);
        acc_store<K, VEC, false>(acc, g.prow + (long long)slot * C + cb, cb, C, lane);
        if (cb == 0) found0 = found;
    }
    if (lane == 0) g.pmeta[slot] = make_int4(found0, found0 > 0 ? 1 : 0, found0 > 0 ? 0 : -1, found0 > 0 ? 0 : -1);
}

// Workgroup role: the GW wavefronts of a workgroup share one voxel.  Used by k_combine_parts to redo a split voxel whose parts came
// up short (whole images) and by k_gather_one's A/B arm (VP_OPT_ONE_VIEW_SPLIT = 0: every voxel above heavy_t pixels in the call).  Per view the box rows are cut into GW contiguous ranges,
// each wavefront sums its range in raster order, and the partial rows are combined through LDS in
// wavefront order -- a fixed summation tree, so results are reproducible run to run (they differ from
// the serial order in the last bits only, well inside the 1e-4 bar).
template <int K, int VEC, int U, int GW>
__device__ bool gather_voxel_block(const GatherArgs &g, const Params &p, int id, int expected,
                                   float (*part)[64 * K * VEC], int *part_found, bool whole_image)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tid = threadIdx.x;
    const int W = p.width, H = p.height, C = p.C;
    const long long HW = (long long)H * W;
    const float hh = box_half_edge(p);
    const float zn = near_plane(p);
    constexpr int CB = 64 * K * VEC;
    constexpr int R = (CB + GW * 64 - 1) / (GW * 64);   // running-sum channels per thread
    int found_total = 0, nviews = 0;
    for (int cb = 0; cb < C; cb += CB) {
        float run[R];
        float *orow = g.out + (long long)id * C + cb;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int c = tid + r * GW * 64;
            run[r] = (c < CB && cb + c < C) ? orow[c] : 0.f;
        }
        found_total = 0;
        nviews = 0;
        for (int b = 0; b < p.B && found_total < expected; b++) {
            float cxw = 0.f, cyw = 0.f, czw = 0.f;
            if (!whole_image && !voxel_centre(g, p, b, id, cxw, cyw, czw)) continue;
            const float fx = g.intr[b * 4 + 0], fy = g.intr[b * 4 + 1], mx = g.intr[b * 4 + 2], my = g.intr[b * 4 + 3];
            for (int vbase = 0; vbase < p.V && found_total < expected; vbase += 64) {
                const int v = vbase + lane;
                int x0 = 0, y0 = 0, x1 = W - 1, y1 = H - 1;
                bool ne = v < p.V;
                if (ne && !whole_image) ne = voxel_box(g.viewtab[b * p.V + v], fx, fy, mx, my, cxw, cyw, czw, hh, zn, W, H, x0, y0, x1, y1);
                unsigned long long vm = __ballot(ne);   // identical in every wavefront of the workgroup
                while (vm && found_total < expected) {
                    const int l = __builtin_ctzll(vm);
                    vm &= vm - 1;
                    const int bx0 = __builtin_amdgcn_readlane(x0, l), by0 = __builtin_amdgcn_readlane(y0, l);
                    const int bx1 = __builtin_amdgcn_readlane(x1, l), by1 = __builtin_amdgcn_readlane(y1, l);
                    const long long bv = (long long)b * p.V + vbase + l;
                    const int per = (by1 - by0 + GW) / GW;
                    const int ry0 = by0 + w * per, ry1 = min(by1, ry0 + per - 1);
                    Acc<K, VEC> acc;
#pragma unroll
                    for (int i = 0; i < K * VEC; i++) acc.a[i] = 0.f;
                    int f = 0;
                    if (ry0 <= ry1)
                        scan_box<K, VEC, U>(feat_ptr<VEC>(g.feats, bv * HW * C), g.hit + bv * HW, W, C, id, bx0, ry0, bx1, ry1, cb, lane, acc, f);
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        if constexpr (VEC == 8) {
#pragma unroll
                            for (int h = 0; h < 2; h++)
                                *reinterpret_cast<float4 *>(&part[w][(k * 64 + lane) * 8 + h * 4]) =
                                    make_float4(acc.a[k * 8 + h * 4 + 0], acc.a[k * 8 + h * 4 + 1], acc.a[k * 8 + h * 4 + 2], acc.a[k * 8 + h * 4 + 3]);
                        } else if constexpr (VEC == 4) {
                            *reinterpret_cast<float4 *>(&part[w][(k * 64 + lane) * 4]) =
                                make_float4(acc.a[k * 4 + 0], acc.a[k * 4 + 1], acc.a[k * 4 + 2], acc.a[k * 4 + 3]);
                        } else {
                            part[w][k * 64 + lane] = acc.a[k];
                        }
                    }
                    if (lane == 0) part_found[w] = f;
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const int c = tid + r * GW * 64;
                        if (c < CB) {
#pragma unroll
                            for (int ww = 0; ww < GW; ww++) run[r] += part[ww][c];
                        }
                    }
                    int fview = 0;
#pragma unroll
                    for (int ww = 0; ww < GW; ww++) fview += part_found[ww];
                    found_total += fview;
                    nviews += fview > 0;
                    __syncthreads();
                }
            }
        }
        if (found_total != expected) return false;   // caller retries over whole images; nothing stored yet
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int c = tid + r * GW * 64;
            if (c < CB && cb + c < C) orow[c] = run[r];
        }
    }
    if (tid == 0) {
        g.count[id] += found_total;
        if (g.views_hit) g.views_hit[id] += nviews;
    }
    return true;
}

// Work list of a call: the voxels that received pixels (and are not heavy), binned by size class -- class k holds the
// voxels with floor(log2(pixels)) == k + 3 (clamped to 0 .. WORK_CLASSES-1).  k_gather walks the classes from the largest
// down, so the long voxels start first and the short ones fill the tail (longest-processing-time-first).  The grid is
// still sized for every row (the host does not know how many voxels a call touches): the wavefronts beyond the end of the
// list read the eight class counters and exit -- no box, no ID image, no output row is touched for an untouched voxel.
// Inside a class the IDs keep roughly their ascending order (neighbouring voxels read neighbouring pixels).  One lane per
// ID, appends aggregated per workgroup.  (Each class has room for every row -- 8 x n_rows ints per buffer set, 64 B of
// workspace per voxel row -- because the class sizes are only known once every workgroup has counted.)
constexpr int WL_PER_THREAD = 16;   // IDs per lane of k_worklist: a 256-thread workgroup bins 4096 voxel IDs

// [row_lo, row_hi): the IDs this call's phase 2 gathers (VP_OPT_ROW_BEGIN / _END; [1, n_rows) when no range is set).
// part_px > 0: the launch also PLANS the split voxels (see "Split voxels" above): a voxel with more than part_t pixels
// gets P = ceil(c / part_px) consecutive part slots and an entry in the split list; slots and entries are handed out per
// workgroup (LDS counters, two global atomics per workgroup).  part_px and part_t >= part_px are such that the parts of a
// call cannot outnumber slot_cap (project_impl; PlanArgs for the numbers derived here); the guard below only keeps a broken
// promise from writing out of bounds, and the consumers clamp what they walk to slot_cap.
// One-view calls without parts (VP_OPT_ONE_VIEW_SPLIT = 0): the voxels above heavy_t are in the march's heavy list (a workgroup
// of k_gather_one sums each) and stay out of this list, ST_NHEAVY is the march's count.  part_px == 0: nothing is split.
__global__ __launch_bounds__(256) void k_worklist(const int *__restrict__ cnt_call, PlanArgs plan, long long n_rows,
                                                  int *__restrict__ work, int *status, int wl_blocks,
                                                  const float *__restrict__ vmi, ViewEntry *viewtab, int n_views,
                                                  long long row_lo, long long row_hi, int4 *__restrict__ parts,
                                                  int4 *__restrict__ split, int slot_cap, int *sticky,
                                                  const int *__restrict__ cell_of_id, const int *__restrict__ hit_waves, int n_hit_waves)
{
    if ((int)blockIdx.x >= wl_blocks) {
        // trailing workgroups: the call's view table (phase 2's world->camera maps), one thread per view -- riding on
        // this launch saves one kernel launch per call
        const int v = ((int)blockIdx.x - wl_blocks) * 256 + (int)threadIdx.x;
        if (v < n_views) view_entry(vmi, viewtab, v);
        return;
    }
    int heavy_t = plan.heavy_t, part_t = plan.part_t, part_px = plan.part_px;
    __shared__ int n_cls[WORK_CLASSES + 2], base_cls[WORK_CLASSES + 2], n_split2, hits_lds;
    // this workgroup's 4096 pixel counts: requested first, so that they travel together with the loads of the hit total below
    // (the launch is a chain of dependent round trips -- counts, the classes' global bases, the list entries -- of 1.5 us each)
    const long long id0 = (long long)blockIdx.x * (256 * WL_PER_THREAD);
    int cv[WL_PER_THREAD];
#pragma unroll
    for (int j = 0; j < WL_PER_THREAD; j++) {
        // consecutive lanes take consecutive IDs, so the ranks inside a class follow the ID order closely
        const long long id = id0 + (long long)j * 256 + threadIdx.x;
        cv[j] = (id >= row_lo && id < row_hi) ? cnt_call[id] : 0;
    }
    if (plan.dyn_px_min > 0) {
        // the march has finished: every workgroup adds up its per-wavefront hit counts (a few thousand ints from L2) -- no
        // atomic on one hot word in the march, no second launch.  2 * hits / part_px <= slot_cap bounds the parts (see above)
        if (threadIdx.x == 0) hits_lds = 0;
        __syncthreads();
        // (16 ints per thread and round, all four loads in flight: one dependent load per round was 5 us of this launch)
        int mine = 0;
        const int4 *hw4 = reinterpret_cast<const int4 *>(hit_waves);      // 256-byte aligned, a multiple of four ints
        const int n4 = n_hit_waves >> 2;
        for (int j = threadIdx.x; j < n4; j += 1024) {
            int4 a[4];
#pragma unroll
            for (int u = 0; u < 4; u++) a[u] = j + u * 256 < n4 ? hw4[j + u * 256] : make_int4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; u++) mine += a[u].x + a[u].y + a[u].z + a[u].w;
        }
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&hits_lds, mine);
        __syncthreads();
        const long long hits = hits_lds;
        if (blockIdx.x == 0 && threadIdx.x == 0) status[ST_NHIT] = (int)hits;
        part_px = (int)max((long long)max(part_px, plan.dyn_px_min), (2 * hits + slot_cap - 1) / slot_cap);
    }
    if (plan.dyn_t_ratio > 0) heavy_t = part_t = (int)min(max((long long)part_px * plan.dyn_t_ratio, (long long)plan.dyn_t_floor), 2147483646ll);
    if (part_px > 0 && part_t < part_px) { part_t = part_px; heavy_t = max(heavy_t, part_t); }
    int *work_n = status + ST_WORK0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { status[ST_HEAVY_T] = heavy_t; status[ST_PART_T] = part_t; status[ST_PART_PX] = part_px; }
    // Appends are aggregated per WORKGROUP through LDS: a handful of global atomics per 4096 IDs.  (Returning integer
    // atomics on a few hot addresses are exactly what slows a concurrently running gather -- DESIGN.md section 2.)
    // Counters WORK_CLASSES and WORK_CLASSES + 1: part slots and split voxels.
    if (threadIdx.x < WORK_CLASSES + 2) n_cls[threadIdx.x] = 0;
    if (threadIdx.x == 0) n_split2 = 0;
    __syncthreads();
    int cls[WL_PER_THREAD], rank[WL_PER_THREAD];
#pragma unroll
    for (int j = 0; j < WL_PER_THREAD; j++) {
        const int c = cv[j];
        cls[j] = -1;
        rank[j] = 0;
        if (c > 0 && c <= heavy_t) {
            cls[j] = min(WORK_CLASSES - 1, max(0, 28 - __builtin_clz(c)));     // floor(log2 c) - 3
            rank[j] = atomicAdd(&n_cls[cls[j]], 1);
        } else if (c > part_t && part_px > 0) {
            cls[j] = WORK_CLASSES;
            rank[j] = atomicAdd(&n_cls[WORK_CLASSES], (c + part_px - 1) / part_px);
            atomicAdd(&n_cls[WORK_CLASSES + 1], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < WORK_CLASSES + 2) {
        const int n = n_cls[threadIdx.x];
        base_cls[threadIdx.x] = n > 0 ? atomicAdd(&work_n[threadIdx.x], n) : 0;
        if (threadIdx.x == WORK_CLASSES + 1 && n > 0 && plan.count_heavy) atomicAdd(&status[ST_NHEAVY], n);      // the counter tests and the bench read
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WL_PER_THREAD; j++) {
        const int id = (int)(id0 + (long long)j * 256 + threadIdx.x);
        if (cls[j] >= 0 && cls[j] < WORK_CLASSES) {
            work[(long long)cls[j] * n_rows + base_cls[cls[j]] + rank[j]] = id;
        } else if (cls[j] == WORK_CLASSES) {
            const int c = cv[j];
            const int P = (c + part_px - 1) / part_px;
            const int base = base_cls[WORK_CLASSES] + rank[j];
            const int sidx = base_cls[WORK_CLASSES + 1] + atomicAdd(&n_split2, 1);
            if (base + P > slot_cap || sidx >= slot_cap) {      // cannot happen (see above); loud rather than out of bounds
                atomicOr(&status[ST_BADID], 1);
                *(volatile int *)&sticky[ST_STICKY_BADID] = 1;
                continue;
            }
            split[sidx] = make_int4(id, base, P, c);
            const int w4 = plan.cell_in_item ? cell_of_id[id] : base;
            for (int q = 0; q < P; q++) parts[base + q] = make_int4(id, q, P, w4);
        }
    }
}

// The gather of every call with more than one view.  Wavefront w of the grid takes item w of the call's work list: first
// the parts of the split voxels (the longest items), then the voxels of the size classes from the largest class down --
// long items start first, short ones fill the tail.  104-112 VGPRs allocated = 4 wavefronts per SIMD -- in pipelined mode a
// gain, because a fifth gather wave would take the room the next call's march needs (forcing <= 96 registers with
// __launch_bounds__(256, 5), one allocation: pipelined +2.5 % fp16 / +2 % R1 / -0.4 % fp32; serial phases -1.0 .. -1.4 %).
// The grid is sized for every row plus every part slot (the host does not know how many a call uses): the wavefronts
// beyond the end of the list read the ten counters and exit.
template <int K, int VEC, int U, int G>
__global__ __launch_bounds__(256) void k_gather(GatherArgs g, Params p)
{
    // The gather is HBM-bound: what matters is that its few instructions (address arithmetic, load issue) go out
    // the moment data returns.  Raised wave priority lets it win instruction arbitration against the issue-bound
    // march waves of the next call that share the SIMD in pipelined mode.
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63;
    long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_parts = min(g.work_n[WORK_CLASSES], g.slot_cap);
    if (w < n_parts) {
        gather_part_wave<K, VEC, U>(g, p, (int)w, lane);
        return;
    }
    w -= n_parts;
    int id = 0;
#pragma unroll
    for (int k = WORK_CLASSES - 1; k >= 0; k--) {
        const int n = g.work_n[k];
        if (id == 0 && w < n) id = g.work[(long long)k * p.n_rows + w];
        w -= n;
    }
    if (id == 0) return;
    const int expected = g.cnt_call[id];
    gather_voxel_wave<K, VEC, U, G, true>(g, p, id, expected, lane);
}

// After k_gather on the same stream: one workgroup per split voxel at a time adds the voxel's partial rows to its row in
// `out`, in slot order.  The four wavefronts take four consecutive runs of the P slots (U rows in flight each), their
// sums meet in LDS and are added to the old row in wavefront order: a fixed tree for a given P.  Pixel counts add exactly;
// a view that two neighbouring parts share is counted once (parts cover ascending, contiguous stretches of the view
// sequence, so only the boundary views can repeat).  Parts that found fewer pixels than the march counted: the search
// boxes missed some (an ID labelling several cells, a degenerate pose) -- the voxel is redone over whole images.
template <int K, int VEC, int U>
__global__ __launch_bounds__(256) void k_combine_parts(GatherArgs g, Params p)
{
    __shared__ __attribute__((aligned(16))) float part[GW_MERGED][64 * K * VEC];
    __shared__ int part_found[GW_MERGED];
    __shared__ int meta[GW_MERGED][4];
    const int n_split = min(g.work_n[WORK_CLASSES + 1], g.slot_cap);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, tid = threadIdx.x;
    const int C = p.C;
    constexpr int CB = 64 * K * VEC;
    constexpr int R = (CB + GW_MERGED * 64 - 1) / (GW_MERGED * 64);
    for (int j = blockIdx.x; j < n_split; j += gridDim.x) {
        const int4 sp = g.split[j];
        const int id = sp.x, base = sp.y, P = sp.z, expected = sp.w;
        if (base < 0 || P <= 0 || (long long)base + P > g.slot_cap) continue;      // never written by k_worklist (its guard fired: ST_BADID is up)
        const int q0 = (int)((long long)P * w / GW_MERGED), q1 = (int)((long long)P * (w + 1) / GW_MERGED);
        {   // this wavefront's run of slots: pixels found, distinct views, first and last contributing view
            int f = 0, nv = 0, first = -1, last = -1;
            for (int q = q0; q < q1; q++) {
                const int4 m = g.pmeta[base + q];
                f += m.x;
                if (m.y > 0) {
                    nv += m.y - (m.z == last ? 1 : 0);
                    if (first < 0) first = m.z;
                    last = m.w;
                }
            }
            if (lane == 0) { meta[w][0] = f; meta[w][1] = nv; meta[w][2] = first; meta[w][3] = last; }
        }
        __syncthreads();
        int found = 0, nviews = 0;
        {
            int last = -1;
#pragma unroll
            for (int ww = 0; ww < GW_MERGED; ww++) {
                found += meta[ww][0];
                if (meta[ww][1] > 0) {
                    nviews += meta[ww][1] - (meta[ww][2] == last ? 1 : 0);
                    last = meta[ww][3];
                }
            }
        }
        __syncthreads();
        if (found != expected) {
            if (tid == 0) atomicAdd(&g.status[ST_BOXMISS], 1);
            gather_voxel_block<K, VEC, U, GW_MERGED>(g, p, id, expected, part, part_found, true);      // count / views_hit included
            __syncthreads();
            continue;
        }
        for (int cb = 0; cb < C; cb += CB) {
            Acc<K, VEC> acc;
#pragma unroll
            for (int i = 0; i < K * VEC; i++) acc.a[i] = 0.f;
            for (int q = q0; q < q1; q += U) {
                Acc<K, VEC> r[U];
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (q + u < q1) acc_load<K, VEC>(r[u], g.prow + (long long)(base + q + u) * C + cb, cb, C, lane);
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (q + u < q1) {
#pragma unroll
                        for (int i = 0; i < K * VEC; i++) acc.a[i] += r[u].a[i];
                    }
            }
#pragma unroll
            for (int k = 0; k < K; k++) {
                if constexpr (VEC == 8) {
#pragma unroll
                    for (int h = 0; h < 2; h++)
                        *reinterpret_cast<float4 *>(&part[w][(k * 64 + lane) * 8 + h * 4]) =
                            make_float4(acc.a[k * 8 + h * 4 + 0], acc.a[k * 8 + h * 4 + 1], acc.a[k * 8 + h * 4 + 2], acc.a[k * 8 + h * 4 + 3]);
                } else if constexpr (VEC == 4) {
                    *reinterpret_cast<float4 *>(&part[w][(k * 64 + lane) * 4]) =
                        make_float4(acc.a[k * 4 + 0], acc.a[k * 4 + 1], acc.a[k * 4 + 2], acc.a[k * 4 + 3]);
                } else {
                    part[w][k * 64 + lane] = acc.a[k];
                }
            }
            __syncthreads();
            float *orow = g.out + (long long)id * C + cb;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const int c = tid + r * GW_MERGED * 64;
                if (c < CB && cb + c < C) {
                    float run = orow[c];
#pragma unroll
                    for (int ww = 0; ww < GW_MERGED; ww++) run += part[ww][c];
                    orow[c] = run;
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            g.count[id] += found;
            if (g.views_hit) g.views_hit[id] += nviews;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One-view calls (B*V == 1): what the unchanged reference pipeline issues, once per image (debug_project_features.py:201-208),
// and what the aggregator's parity mode issues.  A one-view gather is SHORT -- R2: ~27 k touched voxels of ~20 pixels each,
// 1.09 GB, a quarter of a millisecond -- so what k_gather spends per voxel outside the row loads is most of it: a workgroup
// launched per 4 voxel IDs of the scene (50 000 for R2, 45 % of which find nothing to do), and in every wavefront a chain
// of dependent steps before the first row load goes out (work-list entry -> cell of the voxel -> view entry -> box on one
// lane -> ID tile -> ballot).  Here:
//   * the grid is a fixed number of wavefronts (a few per SIMD); wavefront w takes the entries w, w + NW, w + 2 NW, ... of
//     the size-ordered work list -- a static deal of a list that is sorted longest-first, no queue word to contend for;
//   * lane j of the wavefront takes the j-th of ITS entries: the IDs, cells and pixel counts of all the wavefront's voxels
//     arrive in one round of loads instead of one chain per voxel, and their pixel boxes are computed side by side, one
//     voxel per lane (in k_gather the box computation is laid out lane = view: for one view, ~350 VALU instructions on ONE
//     lane at the head of every voxel's chain);
//   * while voxel j's rows stream, the ID tile and the output row of voxel j + 1 are already in flight.
// Every voxel up to the split threshold is still summed by one wavefront in (y, x) order from the row already in `out`: the
// oracle's bits.  Voxels above it are cut into parts (round 6, gather_part_one): the parts lead the deal, k_combine_parts follows on
// the stream.  With VP_OPT_ONE_VIEW_SPLIT = 0 (round 5's path, the A/B arm) the voxels above 320 pixels go to workgroups of the same
// launch instead, four wavefronts per voxel, which join the deal afterwards.
// ------------------------------------------------------------------------------------------------
template <int K, int VEC, int U>
__global__ __launch_bounds__(256) void k_gather_one(GatherArgs g, Params p)
{
    __builtin_amdgcn_s_setprio(3);
    // rows in flight per wavefront of the heavy role: 8 (the deal's wavefronts: U) -- a close-up frame is ALL heavy voxels, a few
    // hundred workgroups of four wavefronts each: with 4 rows of 2 KiB per wavefront too few bytes are in flight to fill HBM
    constexpr int UH = (VEC == 4 && U < 8) ? 8 : U;
    __shared__ __attribute__((aligned(16))) float part[GW_MERGED][64 * K * VEC];
    __shared__ int part_found[GW_MERGED];
    if ((int)blockIdx.x < g.heavy_blocks) {
        const int n_heavy = *g.n_heavy;
        for (int h = blockIdx.x; h < n_heavy; h += g.heavy_blocks) {
            const int id = g.heavy_list[h];
            if (id < g.row_lo || id >= g.row_hi) continue;
            const int expected = g.cnt_call[id];
            if (!gather_voxel_block<K, VEC, UH, GW_MERGED>(g, p, id, expected, part, part_found, false)) {
                if (threadIdx.x == 0) atomicAdd(&g.status[ST_BOXMISS], 1);
                gather_voxel_block<K, VEC, UH, GW_MERGED>(g, p, id, expected, part, part_found, true);
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const long long NW = (long long)gridDim.x * 4;
    const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    // Parts of the view's split voxels (round 6) lead the deal -- they are its longest items: the items of the launch are the
    // n_parts parts, then the size-ordered list; wavefront w takes the items w, w + NW, w + 2 NW, ...  One wavefront per part, into
    // the part's slot; k_combine_parts follows on the stream.
    long long first = w;      // this wavefront's first entry of the list
    if (g.host_word && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(g.host_word, (int)(0x80000000u | ((unsigned)g.host_seq << 16) | (unsigned)min(g.work_n[WORK_CLASSES + 1], 0xffff)),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (g.parts_on) {
        const long long n_parts = min(g.work_n[WORK_CLASSES], g.slot_cap);
        long long s = w;
        for (; s < n_parts; s += NW) gather_part_one<K, VEC, U>(g, p, (int)s, lane);
        first = s - n_parts;
    }
    int n[WORK_CLASSES];
    long long total = 0;
#pragma unroll
    for (int k = 0; k < WORK_CLASSES; k++) { n[k] = g.work_n[k]; total += n[k]; }
    const int W = p.width, H = p.height, C = p.C;
    constexpr int CB = 64 * K * VEC;
    const float *fv = g.feats;
    const int *hv = g.hit;
    for (long long base = first; base < total; base += 64 * NW) {
        // lane j: the j-th entry of this wavefront in this batch -- ID, box, pixel count
        long long r = base + (long long)lane * NW;
        int id_l = 0;
        if (r < total) {
#pragma unroll
            for (int k = WORK_CLASSES - 1; k >= 0; k--) {
                if (id_l == 0 && r < n[k]) id_l = g.work[(long long)k * p.n_rows + r];
                r -= n[k];
            }
        }
        int bx0_l = 0, by0_l = 0, bx1_l = -1, by1_l = -1, cnt_l = 0;      // x1 < x0: no box, scan the whole image
        if (id_l != 0) {
            cnt_l = g.cnt_call[id_l];
            float cxw, cyw, czw;
            if (voxel_centre(g, p, 0, id_l, cxw, cyw, czw)) {
                int a0, a1, a2, a3;
                if (voxel_box(g.viewtab[0], g.intr[0], g.intr[1], g.intr[2], g.intr[3], cxw, cyw, czw, box_half_edge(p), near_plane(p),
                              W, H, a0, a1, a2, a3)) { bx0_l = a0; by0_l = a1; bx1_l = a2; by1_l = a3; }
            }
        }
        const int nv = __popcll(__ballot(id_l != 0));      // the entries of a wavefront fill its lanes from 0 upwards
        // the first voxel's ID tile and output row
        int id = 0, x0 = 0, y0 = 0, x1 = -1, y1 = -1, expected = 0, tile = 0;
        Acc<K, VEC> acc;
        auto fetch = [&](int j) {
            id = __builtin_amdgcn_readlane(id_l, j);
            expected = __builtin_amdgcn_readlane(cnt_l, j);
            x0 = __builtin_amdgcn_readlane(bx0_l, j); y0 = __builtin_amdgcn_readlane(by0_l, j);
            x1 = __builtin_amdgcn_readlane(bx1_l, j); y1 = __builtin_amdgcn_readlane(by1_l, j);
            if (x1 < x0 || y1 < y0) { x0 = 0; y0 = 0; x1 = W - 1; y1 = H - 1; }      // no box: the whole image
            tile = first_tile_id(hv, W, x0, y0, x1, y1, lane);
            acc_load<K, VEC>(acc, g.out + (long long)id * C, 0, C, lane);
        };
        if (nv > 0) fetch(0);
        for (int j = 0; j < nv; j++) {
            const int cid = id, cx0 = x0, cy0 = y0, cx1 = x1, cy1 = y1, cexp = expected, ctile = tile;
            Acc<K, VEC> cacc = acc;
            if (j + 1 < nv) fetch(j + 1);       // in flight while this voxel's rows stream
            for (int cb = 0; cb < C; cb += CB) {
                float *orow = g.out + (long long)cid * C + cb;
                if (cb > 0) acc_load<K, VEC>(cacc, orow, cb, C, lane);
                int found = 0;
                scan_box<K, VEC, U>(fv, hv, W, C, cid, cx0, cy0, cx1, cy1, cb, lane, cacc, found, cb == 0, ctile);
                if (found != cexp) {
                    // the box missed pixels (an ID labelling several cells, a degenerate pose ...): redo over the whole
                    // image from the row as it still is in memory.  Correctness never depends on the boxes.
                    if (lane == 0 && cb == 0) atomicAdd(&g.status[ST_BOXMISS], 1);
                    acc_load<K, VEC>(cacc, orow, cb, C, lane);
                    found = 0;
                    scan_box<K, VEC, U>(fv, hv, W, C, cid, 0, 0, W - 1, H - 1, cb, lane, cacc, found);
                }
                acc_store<K, VEC, false>(cacc, orow, cb, C, lane);
                if (cb == 0 && lane == 0) {
                    g.count[cid] += found;
                    if (g.views_hit) g.views_hit[cid] += found > 0;
                }
            }
        }
    }
}

}  // namespace
