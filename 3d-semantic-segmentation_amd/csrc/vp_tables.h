// vp_tables.h -- the arithmetic contract shared with oracle/projector_oracle.c (rounding, float->int), and the
// tables derived once per occupancy grid (ID -> cell, 4x4x4 block masks, block distance field, near-field cell
// distances) plus the per-call view table.  Included by voxproj.hip only.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// device helpers: the arithmetic contract of oracle/projector_oracle.c
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float round_half_away(float x)
{   // C roundf; x - trunc(x) is exact in binary32
    float t = truncf(x);
    float d = fabsf(x - t);
    return d >= 0.5f ? t + copysignf(1.0f, x) : t;
}

__device__ __forceinline__ int f2i_sat(float v)
{   // cvt.rzi.s32.f32 / v_cvt_i32_f32 semantics: saturate, NaN -> 0
    if (v != v) return 0;
    v = fminf(fmaxf(v, -2147483648.0f), 2147483520.0f);
    return (int)v;
}

// Closed-form advance of the ray parameter (used inside k_first_hit): J repetitions of t = fl(t + inc) without J
// dependent additions.  While t stays inside one binade [T, 2T), T = 2^e > inc, every addition rounds to the same
// grid of spacing u = ulp(T): fl(t + inc) = t + g with g = inc rounded to a multiple of u (unless inc lies exactly
// half-way between two multiples, where round-to-even depends on t; that binade is stepped one addition at a
// time).  g = fl(T + inc) - T.  For any m <= floor(((2T - u) - t) / g) each of the m exact sums t_i + inc stays
// below 2T, so every step adds exactly g, and t + m*g (a multiple of u below 2T) is exactly representable: one
// multiply and one add reproduce m additions.  m may be under-estimated (reciprocal scaled by 0.999999) -- the
// remaining steps are then taken by real additions; the addition that crosses the binade edge is always a real
// one.  All operations are IEEE binary32; tests compare 530k full-resolution rays with the oracle's plain loop.

// ------------------------------------------------------------------------------------------------
// occupancy-derived tables (built once per occupancy grid, see VP_FLAG_REUSE_ACCEL):
//   cell_of_id[b][id]   linear cell index of voxel `id` (largest cell wins if an ID labels several)
//   mask64[b][blk]      one bit per cell of each 4x4x4 block: (int)occ != 0   (bit = z%4*16+y%4*4+x%4)
//   dist[b][blk]        Chebyshev distance, in blocks, to the nearest non-empty block (0 = non-empty,
//                       capped at 255) -- a lower bound that lets the march leap over empty space
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_cells(const long long *__restrict__ occ, int *cell_of_id,
                                                     unsigned long long *mask64,
                                                     int dimz, int dimy, int dimx, int nby, int nbx,
                                                     long long nblk, int B, long long n_rows)
{
    const long long cells_per_batch = (long long)dimz * dimy * dimx;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long total = cells_per_batch * B;
    for (; i < total; i += stride) {
        const int id = (int)occ[i];   // K.cu:70 long -> int
        if (id == 0) continue;
        const int b = (int)(i / cells_per_batch);
        const int cell = (int)(i - (long long)b * cells_per_batch);
        const int z = cell / (dimy * dimx), r = cell - z * (dimy * dimx), y = r / dimx, x = r - y * dimx;
        const long long blk = ((long long)(z >> 2) * nby + (y >> 2)) * nbx + (x >> 2);
        const int bit = ((z & 3) << 4) | ((y & 3) << 2) | (x & 3);
        atomicOr(&mask64[(long long)b * nblk + blk], 1ull << bit);
        if (id > 0 && id < n_rows) atomicMax(&cell_of_id[(long long)b * n_rows + id], cell);
    }
}

// VP_FLAG_VERIFY_ACCEL: the tables depend on the grid only through (int)occ[i] (k_build_cells above).  Compare the
// caller's grid with the 32-bit copy taken when the tables were built, refresh the copy where it differs, and
// raise *differs if any cell changed.
__global__ __launch_bounds__(256) void k_occ_compare_copy(const long long *__restrict__ occ, int *__restrict__ copy,
                                                          long long n, int *differs)
{
    bool d = false;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int v = (int)occ[i];
        if (copy[i] != v) { copy[i] = v; d = true; }
    }
    if (d) *differs = 1;
}

// Separable Chebyshev distance transform on the block grid: D = min_q max(|dx|,|dy|,|dz|) factors into
// three 1-D passes because max distributes over min.  axis 0: along x from the masks; 1: y; 2: z.
__global__ __launch_bounds__(256) void k_block_dist(const unsigned long long *__restrict__ mask64,
                                                    const unsigned char *__restrict__ src,
                                                    unsigned char *__restrict__ dst,
                                                    int nbz, int nby, int nbx, long long nblk_padded, int B, int axis)
{
    const long long nreal = (long long)nbz * nby * nbx;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nreal * B) return;
    const long long base = (i / nreal) * nblk_padded;     // per-batch tables are padded to 16 entries
    const int blk = (int)(i % nreal);
    const int z = blk / (nby * nbx), r = blk - z * (nby * nbx), y = r / nbx, x = r - y * nbx;
    int best = 255;
    if (axis == 0) {
        for (int q = 0; q < nbx; q++)
            if (mask64[base + ((long long)z * nby + y) * nbx + q] != 0ull) best = min(best, abs(x - q));
    } else if (axis == 1) {
        for (int q = 0; q < nby; q++)
            best = min(best, max(abs(y - q), (int)src[base + ((long long)z * nby + q) * nbx + x]));
    } else {
        for (int q = 0; q < nbz; q++)
            best = min(best, max(abs(z - q), (int)src[base + ((long long)q * nby + y) * nbx + x]));
    }
    dst[base + blk] = (unsigned char)best;
}

// Near field: for every cell of every block within one block of an occupied block, the Chebyshev distance in
// CELLS to the nearest occupied cell, capped at NEAR_CAP ("7 or more"), stored as three bit planes per 4x4x4 block
// (NearRec, vp_common.h; nd == 0 <=> the cell is occupied).  One wavefront per block, one lane per cell.  An occupied cell
// within 6 cells of a cell of this block lies in one of the 5x5x5 blocks around it: their 125 occupancy masks are fetched
// in one round (two per lane), then walked in a wave-uniform loop -- empty blocks skipped, every set bit of the others
// compared with all 64 cells at once (nine VALU instructions per occupied cell in the neighbourhood; the first version
// searched shells of cells one dependent table read at a time and took 1.15 ms for R2 instead of 0.2).
constexpr int NEAR_CAP = 7;

__global__ __launch_bounds__(256) void k_build_near(const unsigned long long *__restrict__ mask64,
                                                    const unsigned char *__restrict__ dist, NearRec *near2,
                                                    int dimz, int dimy, int dimx, int nbz, int nby, int nbx,
                                                    long long nblk, int B)
{
    (void)dimz; (void)dimy; (void)dimx;      // cells beyond the grid are never set in the masks
    const long long wid = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long long nreal = (long long)nbz * nby * nbx;
    if (wid >= nreal * B) return;
    const int b = (int)(wid / nreal);
    const int blk = (int)(wid - (long long)b * nreal);
    const unsigned long long *mask_b = mask64 + (long long)b * nblk;
    const int bd = dist[(long long)b * nblk + blk];
    int nd = NEAR_CAP;
    if (bd <= 1) {
        const int bz = blk / (nby * nbx), r = blk - bz * (nby * nbx), by = r / nbx, bx = r - by * nbx;
        // lane l holds the masks of neighbours l and 64 + l of the 5x5x5 neighbourhood (index = (dz+2)*25 + (dy+2)*5 + dx+2)
        unsigned long long m_lo = 0ull, m_hi = 0ull;
        {
            const int i0 = lane, i1 = lane + 64;
            const int z0 = bz + i0 / 25 - 2, y0 = by + (i0 / 5) % 5 - 2, x0 = bx + i0 % 5 - 2;
            if ((unsigned)z0 < (unsigned)nbz && (unsigned)y0 < (unsigned)nby && (unsigned)x0 < (unsigned)nbx)
                m_lo = mask_b[((long long)z0 * nby + y0) * nbx + x0];
            if (i1 < 125) {
                const int z1 = bz + i1 / 25 - 2, y1 = by + (i1 / 5) % 5 - 2, x1 = bx + i1 % 5 - 2;
                if ((unsigned)z1 < (unsigned)nbz && (unsigned)y1 < (unsigned)nby && (unsigned)x1 < (unsigned)nbx)
                    m_hi = mask_b[((long long)z1 * nby + y1) * nbx + x1];
            }
        }
        // this lane's cell, in cells relative to the block's own corner
        const int cx = lane & 3, cy = (lane >> 2) & 3, cz = lane >> 4;
        for (int i = 0; i < 125; i++) {
            const unsigned lo32 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(i < 64 ? m_lo : m_hi), i & 63);
            const unsigned hi32 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((i < 64 ? m_lo : m_hi) >> 32), i & 63);
            unsigned long long m = ((unsigned long long)hi32 << 32) | lo32;
            if (m == 0ull) continue;
            const int ox = (i % 5 - 2) * 4, oy = ((i / 5) % 5 - 2) * 4, oz = (i / 25 - 2) * 4;      // the neighbour's corner
            while (m) {
                const int bit = __builtin_ctzll(m);
                m &= m - 1;
                const int dx = abs(ox + (bit & 3) - cx), dy = abs(oy + ((bit >> 2) & 3) - cy), dz = abs(oz + (bit >> 4) - cz);
                nd = min(nd, max(dx, max(dy, dz)));
            }
        }
    }
    const unsigned long long p0 = __ballot(nd & 1), p1 = __ballot(nd & 2), p2 = __ballot(nd & 4);
    if (lane == 0) {
        NearRec rec;
        rec.p0 = p0; rec.p1 = p1; rec.p2 = p2; rec.dist = (unsigned long long)bd;
        near2[(long long)b * nblk + blk] = rec;
    }
}

// Workspace header (vp_common.h, WsState): initialise the status blocks of memory that does not carry this record's
// generation (first call of a record, or memory recycled / overwritten since), leave them alone otherwise -- so pending
// sticky errors survive a table rebuild.  One workgroup of 64 threads, one status word each.
__global__ __launch_bounds__(64) void k_ws_open(int *status0, int *status1, unsigned magic, unsigned gen, int may_init)
{
    const bool ok = (unsigned)status0[ST_HDR_MAGIC] == magic && (unsigned)status0[ST_HDR_GEN] == gen;
    __syncthreads();
    if (ok || !may_init) return;
    const int t = threadIdx.x;
    status0[t] = t == ST_HDR_MAGIC ? (int)magic : t == ST_HDR_GEN ? (int)gen : 0;     // ST_HDR_TABLES = 0: no tables yet
    status1[t] = 0;
}

// after a table build: the header names the tables this memory now holds
__global__ void k_ws_seal(int *status0, unsigned tables) { status0[ST_HDR_TABLES] = (int)tables; }

// per-call reset: the status words of this buffer set and the per-call hit histogram, in one launch; one thread also
// compares the workspace header with what the host record expects (generation, tables key).  A mismatch marks the call
// stale: k_first_hit then does nothing (so neither does the gather: the histogram stays zero) and the sticky word makes
// vp_workspace_status report it.
__global__ __launch_bounds__(256) void k_zero_call(int *__restrict__ status, int *__restrict__ cnt_call, long long n_rows,
                                                   int *hdr, int *sticky, unsigned magic, unsigned gen, unsigned tables,
                                                   int *__restrict__ hit_waves, long long n_hit_waves)
{
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    // the march's per-wavefront hit counts (one-view calls): wavefronts whose tile lies outside the image never write theirs
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n_hit_waves; j += (long long)gridDim.x * blockDim.x) hit_waves[j] = 0;
    if (blockIdx.x == 0 && threadIdx.x < ST_CALL_WORDS) {
        int v = 0;
        if (threadIdx.x == ST_STALE) {
            const bool mine = (unsigned)hdr[ST_HDR_MAGIC] == magic && (unsigned)hdr[ST_HDR_GEN] == gen;
            if (!mine || (unsigned)hdr[ST_HDR_TABLES] != tables) {
                v = 1;
                if (!mine) {
                    // not this record's memory (any more): the block becomes this record's from here on, with "no tables" in
                    // the header, so every later call that trusts tables is stale too
                    hdr[ST_HDR_MAGIC] = (int)magic; hdr[ST_HDR_GEN] = (int)gen; hdr[ST_HDR_TABLES] = 0;
                }
                *(volatile int *)&sticky[ST_STICKY_STALE] = 1;
            }
        }
        status[threadIdx.x] = v;
    }
    if (i + 3 < n_rows) *reinterpret_cast<int4 *>(cnt_call + i) = make_int4(0, 0, 0, 0);
    else
        for (long long j = i; j < n_rows; j++) cnt_call[j] = 0;
}

// ------------------------------------------------------------------------------------------------
// view table: invert each view's 3x3 (double precision) for the phase-2 search boxes (computed by the trailing
// workgroups of k_worklist, vp_gather.h)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void view_entry(const float *__restrict__ vmi, ViewEntry *tab, int i)
{
    const float *m = vmi + (long long)i * 16;
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], k = m[10];
    double A = e * k - f * h, Bc = -(d * k - f * g), Cc = d * h - e * g;
    double det = a * A + b * Bc + c * Cc;
    ViewEntry ve;
    double scale = fabs(a) + fabs(b) + fabs(c) + fabs(d) + fabs(e) + fabs(f) + fabs(g) + fabs(h) + fabs(k);
    bool ok = (det == det) && fabs(det) > 1e-12 * scale * scale * scale && scale < 1e18;
    double r = ok ? 1.0 / det : 0.0;
    ve.inv[0] = (float)(A * r);  ve.inv[1] = (float)(-(b * k - c * h) * r); ve.inv[2] = (float)((b * f - c * e) * r);
    ve.inv[3] = (float)(Bc * r); ve.inv[4] = (float)((a * k - c * g) * r);  ve.inv[5] = (float)(-(a * f - c * d) * r);
    ve.inv[6] = (float)(Cc * r); ve.inv[7] = (float)(-(a * h - b * g) * r); ve.inv[8] = (float)((a * e - b * d) * r);
    ve.pos[0] = m[3]; ve.pos[1] = m[7]; ve.pos[2] = m[11];
    for (int j = 0; j < 9; j++) ok = ok && (fabsf(ve.inv[j]) < 1e18f);
    for (int j = 0; j < 3; j++) ok = ok && (fabsf(ve.pos[j]) < 1e18f);
    ve.ok = ok ? 1.0f : 0.0f;
    ve.pad[0] = ve.pad[1] = ve.pad[2] = 0.0f;
    tab[i] = ve;
}

}  // namespace
