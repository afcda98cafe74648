// vp_march.h -- phase 1: the first-hit ray-march (k_first_hit), exact replay of the reference's sample sequence with
// proven leaps over empty space.  Included by voxproj.hip only.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// phase 1: first-hit ray-march.  One lane per pixel, 8x8 pixel tile per wavefront (coherent rays),
// 16x16 per workgroup, blockIdx.z = b*V + v.
//
// ACCEL = false: the reference loop, one occupancy probe per step (K.cu:47-82), kept as the A/B arm.
// ACCEL = true : the same sample sequence t_k (repeated fp32 addition, never t0 + k*inc), but
//   * samples that provably cannot land in an occupied cell are not evaluated: from the block
//     distance field, a sample in cell c with D = (lower bound on the Chebyshev distance, in cells,
//     from c to the nearest occupied cell) allows skipping J steps with 1.5 + J*dcell <= D, where
//     dcell bounds the per-step motion in cells (1% + fp slack, see DESIGN.md for the proof);
//   * the cell index roundf((p - origin)/vs) is taken from the product with 1/vs when that product is
//     farther than 2^-21*|q| from a rounding boundary (then both roundings agree), and from the IEEE
//     division otherwise;
//   * near geometry the bound D is the cell's own Chebyshev distance (0..7, three 64-bit bit planes per 4x4x4 block,
//     held in registers while the ray stays in the block); D = 0 means "this cell is occupied" and only then is
//     the int64 grid read for the ID;
//   * the (u,v) bounds test of K.cu:53-61 is evaluated, with the reference's exact operations, only
//     for samples that found an occupied cell (it gates nothing else).
// Every evaluated sample uses the reference's exact fp32 operations, so the first-hit ID is identical.
//
// Round 5 tried a TAIL SPLIT (once at most 8 rays of a wavefront were still marching, its 64 lanes were dealt out again, several
// per ray, each taking a stretch of the ray's remaining samples from its exact repeated-sum starting parameter; bit-exact) for
// launches of ONE view, which last as long as their slowest wavefront: 10 % SLOWER (R1 march 33.3 -> 38.0 us, R2 54.8 -> 58.4).
// The slow rays come in whole tiles -- coherent rays creeping along one surface -- so the slow wavefronts have no idle lanes to
// deal out, and half of all wavefronts paid the exchange for two more iterations (profiles/r05_march_tail_split.log,
// tools/march_waves.py).  Removed; march_sample / march_advance below are what is left of the restructuring.
// ------------------------------------------------------------------------------------------------
// MODE 0: the reference loop (A/B arm, VP_FLAG_EXACT_MARCH); MODE 1: the leaping march.
struct FirstHitArgs {
    const long long *occ;
    const float *vmi;
    const float *intr;
    const NearRec *near2;
    const unsigned char *dist;
    int nby, nbx;
    long long nblk;
    int *hit;
    int *cnt_call;
    int *heavy_list;
    int heavy_t;
    int *hit_waves; // nullable: [wavefront of the grid] = pixels of its tile whose ray hit a voxel (one-view calls that size their
                    // parts from the total; a plain store per wavefront -- one atomic word for all of them cost the march 85 us)
    int *status;
    int *sticky;   // the workspace record's sticky error words (pinned host memory, device mapping; never cleared by a call)
};

// what a ray's samples are evaluated against: constant per (b, v), uniform over a wavefront
struct MarchView {
    const long long *occ_b;
    const NearRec *near_b;
    const unsigned char *dist_b;
    int nby, nbx;
    float cpx, cpy, cpz;        // camera position (K.cu:185)
    float fx, fy, mx, my, fw, fh;
    float rvs;                  // 1 / voxel size
};

// a ray of the leaping march
struct MarchRay {
    float cdx, cdy, cdz;        // camDir (K.cu:184)
    float wdx, wdy, wdz;        // worldDir (K.cu:186-187)
    float tEnd;
    float inv_dcell;            // 0.999 / (bound of the motion per step in cells); 0 where leaping is not allowed
    float thr;                  // |q - rint(q)| below which the product with 1/vs rounds like the IEEE quotient; -1: always divide
    bool leap_ok;
};

// the 4x4x4 block a ray is in: its distance record, held in registers while the ray stays there
struct MarchBlock {
    unsigned cur_blk = 0xffffffffu;
    int cur_d = 0;
    unsigned long long p0 = 0ull, p1 = 0ull, p2 = 0ull;
};

// binade cache of the closed-form t advance (derivation: "Closed-form advance" in vp_tables.h): valid while t < bT2
struct MarchBinade {
    float bT2 = 0.0f, bTu = 0.0f, bg = 0.0f, brg = 0.0f;
};

// One sample of the leaping march at ray parameter t (K.cu:49-81 for the samples that can matter): returns the ID of an
// occupied, in-image cell (non-zero = the first hit if no earlier sample had one), and in D a lower bound on the Chebyshev
// cell distance from the sample's cell to any occupied cell.
__device__ __forceinline__ int march_sample(const Params &p, const MarchView &mv, const MarchRay &r, MarchBlock &mb, float t, int &D)
{
    const float px = mv.cpx + t * r.wdx, py = mv.cpy + t * r.wdy, pz = mv.cpz + t * r.wdz;
    const float ax = px - p.ox, ay = py - p.oy, az = pz - p.oz;
    const float qx = ax * mv.rvs, qy = ay * mv.rvs, qz = az * mv.rvs;
    const float rx = rintf(qx), ry = rintf(qy), rz = rintf(qz);
    const bool safe = (fabsf(qx - rx) < r.thr) & (fabsf(qy - ry) < r.thr) & (fabsf(qz - rz) < r.thr);
    int ix = (int)rx, iy = (int)ry, iz = (int)rz;   // exact when safe (|q| < 2^17)
    if (__builtin_expect(!safe, 0)) {
        ix = f2i_sat(round_half_away(ax / p.vs));
        iy = f2i_sat(round_half_away(ay / p.vs));
        iz = f2i_sat(round_half_away(az / p.vs));
    }
    D = 0;
    const bool ing = ((unsigned)ix < (unsigned)p.dimx) & ((unsigned)iy < (unsigned)p.dimy) & ((unsigned)iz < (unsigned)p.dimz);
    if (__builtin_expect(ing, 1)) {
        const unsigned blk = ((unsigned)(iz >> 2) * (unsigned)mv.nby + (unsigned)(iy >> 2)) * (unsigned)mv.nbx + (unsigned)(ix >> 2);
        if (blk != mb.cur_blk) {
            mb.cur_blk = blk;
            // one 32-byte record: the block's distance in blocks and the three bit planes of its cells' distances
            // (the planes are only meaningful when cur_d <= 1)
            const NearRec n3 = mv.near_b[blk];
            mb.cur_d = (int)n3.dist;
            mb.p0 = n3.p0; mb.p1 = n3.p1; mb.p2 = n3.p2;
        }
        const int bit = ((iz & 3) << 4) | ((iy & 3) << 2) | (ix & 3);
        const int nd = (int)((mb.p0 >> bit) & 1ull) | ((int)((mb.p1 >> bit) & 1ull) << 1) | ((int)((mb.p2 >> bit) & 1ull) << 2);
        D = mb.cur_d <= 1 ? nd : (mb.cur_d - 1) * 4 + 1;
        if (__builtin_expect((mb.cur_d <= 1) & (nd == 0), 0)) {
            const float camx = r.cdx * t, camy = r.cdy * t, camz = r.cdz * t;
            const float u = mv.fx * (camx / camz) + mv.mx;
            const float v = mv.fy * (camy / camz) + mv.my;
            if ((u >= 0.0f) && (u < mv.fw) && (v >= 0.0f) && (v < mv.fh))
                return (int)mv.occ_b[((long long)iz * p.dimy + iy) * p.dimx + ix];
        }
    } else if (r.leap_ok) {
        const int lim = 1 << 29;
        const int jx = min(max(ix, -lim), lim), jy = min(max(iy, -lim), lim), jz = min(max(iz, -lim), lim);
        const int ex = jx < 0 ? -jx : (jx >= p.dimx ? jx - p.dimx + 1 : 0);
        const int ey = jy < 0 ? -jy : (jy >= p.dimy ? jy - p.dimy + 1 : 0);
        const int ez = jz < 0 ? -jz : (jz >= p.dimz ? jz - p.dimz + 1 : 0);
        const int dbox = max(ex, max(ey, ez));   // every occupied cell lies inside the grid box
        const int kx = min(max(jx, 0), p.dimx - 1), ky = min(max(jy, 0), p.dimy - 1), kz = min(max(jz, 0), p.dimz - 1);
        const int cb_ = ((kz >> 2) * mv.nby + (ky >> 2)) * mv.nbx + (kx >> 2);
        const int dd = mv.dist_b[cb_];
        const int din = dd > 0 ? (dd - 1) * 4 + 1 : 0;
        D = max(dbox, din - dbox);
    }
    return 0;
}

// samples a bound D allows to advance by: the evaluated one plus J = floor((D - 1.5) / dcell) provably unable to reach an
// occupied cell
__device__ __forceinline__ int march_steps(const MarchRay &r, int D)
{
    return 1 + (D >= 2 ? (int)fminf(((float)D - 1.5f) * r.inv_dcell, 16777216.0f) : 0);
}

// S repetitions of t = fl(t + inc) (stops early once t >= tEnd): the running sum is reproduced exactly by the closed form,
// with the binade constants cached across calls
__device__ __forceinline__ void march_advance(float inc, float tEnd, MarchBinade &bn, float &t, int S)
{
    for (;;) {
        if (t >= bn.bT2) {
            const unsigned eb = __float_as_uint(t) & 0x7f800000u;
            const float T = __uint_as_float(eb);
            const float u = __uint_as_float(eb - (23u << 23));
            bn.bT2 = __uint_as_float(eb + (1u << 23));
            bn.bTu = bn.bT2 - u;
            bn.bg = (T + inc) - T;
            const float r = inc - bn.bg;
            const bool fast = (t > 0.0f) & (eb >= (30u << 23)) & (eb < (0xfeu << 23)) & (inc < T) & (bn.bg > 0.0f) & (fabsf(r) * 2.0f != u);
            bn.brg = fast ? __builtin_amdgcn_rcpf(bn.bg) * 0.999999f : 0.0f;   // under-estimate: m <= floor(A/g)
        }
        const int m = (int)fminf(fmaxf((bn.bTu - t) * bn.brg, 0.0f), (float)S);
        t = t + (float)m * bn.bg;
        S -= m;
        if (S <= 0) break;
        t += inc;          // the addition that crosses the binade edge (or a binade stepped one by one)
        S -= 1;
        if (S <= 0 || !(t < tEnd)) break;
    }
}

template <int MODE>
__device__ __forceinline__ void first_hit_body(const FirstHitArgs &fa, const Params &p, int x, int y, int bv)
{
    constexpr bool ACCEL = MODE != 0;
    const long long *__restrict__ occ = fa.occ;
    const float *__restrict__ vmi = fa.vmi;
    const float *__restrict__ intr = fa.intr;
    const long long nblk = fa.nblk;
    int *__restrict__ hit = fa.hit;
    int *cnt_call = fa.cnt_call, *heavy_list = fa.heavy_list, *status = fa.status;
    const int heavy_t = fa.heavy_t;
    const int b = bv / p.V;
#ifdef VP_DIAG
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    int dbg_it = 0;
#endif
    if (x >= p.width || y >= p.height) return;
    if (status[ST_STALE]) return;      // the workspace does not hold the tables this call was told to trust (k_zero_call)

    const float *m = vmi + (long long)bv * 16;           // K.cu:178-179 (row-major float4x4)
    const float fx = intr[b * 4 + 0], fy = intr[b * 4 + 1], mx = intr[b * 4 + 2], my = intr[b * 4 + 3];

    // K.cu:182-184, cudaUtil.h:106-119
    const float depth = 1.0f * (p.dmax - p.dmin) + p.dmin;
    const float sx = ((float)(unsigned)x - mx) / fx;
    const float sy = ((float)(unsigned)y - my) / fy;
    float cx = depth * sx, cy = depth * sy, cz = depth;
    float inv = 1.0f / sqrtf(cx * cx + cy * cy + cz * cz);   // cutil_math.h:1207-1211, :81-84
    const float cdx = cx * inv, cdy = cy * inv, cdz = cz * inv;
    // K.cu:185 float4x4 * float3 (w = 1), cuda_SimpleMatrixUtil.h:900-908
    const float cpx = m[0] * 0.0f + m[1] * 0.0f + m[2] * 0.0f + m[3] * 1.0f;
    const float cpy = m[4] * 0.0f + m[5] * 0.0f + m[6] * 0.0f + m[7] * 1.0f;
    const float cpz = m[8] * 0.0f + m[9] * 0.0f + m[10] * 0.0f + m[11] * 1.0f;
    // K.cu:186-187 float4x4 * float4(camDir, 0), cuda_SimpleMatrixUtil.h:888-896
    float wx = m[0] * cdx + m[1] * cdy + m[2] * cdz + m[3] * 0.0f;
    float wy = m[4] * cdx + m[5] * cdy + m[6] * cdz + m[7] * 0.0f;
    float wz = m[8] * cdx + m[9] * cdy + m[10] * cdz + m[11] * 0.0f;
    inv = 1.0f / sqrtf(wx * wx + wy * wy + wz * wz);
    const float wdx = wx * inv, wdy = wy * inv, wdz = wz * inv;

    // K.cu:31-82
    const float d2r = 1.0f / cdz;
    float t = d2r * p.dmin;
    const float tEnd = d2r * p.dmax;
    const long long cells = (long long)p.dimz * p.dimy * p.dimx;
    const long long *occ_b = occ + (long long)b * cells;
    const float fw = (float)p.width, fh = (float)p.height;
    int id = 0;
    // t += inc must make progress all the way to tEnd, or the loop (the reference's too, K.cu:47,81) never ends:
    // ulp(t) <= ulp(tEnd), so it does iff adding inc changes tEnd.  Such a ray is reported, not marched.
    if ((t < tEnd) && !(tEnd + p.inc > tEnd)) {
        atomicOr(&status[ST_STUCK], 1);
        *(volatile int *)&fa.sticky[ST_STICKY_STUCK] = 1;      // a plain store of the constant: host memory, no atomic needed
        t = tEnd;
    }
    if constexpr (!ACCEL) {
        while (t < tEnd) {
            const float px = cpx + t * wdx, py = cpy + t * wdy, pz = cpz + t * wdz;
            const int ix = f2i_sat(round_half_away((px - p.ox) / p.vs));
            const int iy = f2i_sat(round_half_away((py - p.oy) / p.vs));
            const int iz = f2i_sat(round_half_away((pz - p.oz) / p.vs));
            const float camx = cdx * t, camy = cdy * t, camz = cdz * t;
            const float u = fx * (camx / camz) + mx;
            const float v = fy * (camy / camz) + my;
            const bool inb = (u >= 0.0f) && (u < fw) && (v >= 0.0f) && (v < fh);
            if (inb && ix >= 0 && iy >= 0 && iz >= 0 && ix < p.dimx && iy < p.dimy && iz < p.dimz) {
                id = (int)occ_b[((long long)iz * p.dimy + iy) * p.dimx + ix];
                if (id != 0) break;
            }
            t += p.inc;
        }
    } else {
        MarchView mv;
        mv.occ_b = occ_b;
        mv.near_b = fa.near2 + (long long)b * nblk;
        mv.dist_b = fa.dist + (long long)b * nblk;
        mv.nby = fa.nby; mv.nbx = fa.nbx;
        mv.cpx = cpx; mv.cpy = cpy; mv.cpz = cpz;
        mv.fx = fx; mv.fy = fy; mv.mx = mx; mv.my = my; mv.fw = fw; mv.fh = fh;
        mv.rvs = 1.0f / p.vs;
        MarchRay r;
        r.cdx = cdx; r.cdy = cdy; r.cdz = cdz; r.wdx = wdx; r.wdy = wdy; r.wdz = wdz; r.tEnd = tEnd;
        // upper bound of the per-step motion in cells (1% covers the rounding of t += inc and of rvs)
        const float dcell = fabsf(p.inc * mv.rvs) * fmaxf(fabsf(wdx), fmaxf(fabsf(wdy), fabsf(wdz))) * 1.01f + 1e-6f;
        // leaping is allowed only where fp32 position error stays far below one cell and the step count
        // is sane; otherwise every sample is evaluated (still exact, just slower)
        const float span = (fabsf(cpx) + fabsf(cpy) + fabsf(cpz) + fabsf(p.ox) + fabsf(p.oy) + fabsf(p.oz) + fabsf(tEnd)) * fabsf(mv.rvs);
        r.leap_ok = (span < 131072.0f) & (fabsf(tEnd) < 1.0e5f * fabsf(p.inc)) & (dcell == dcell) & (dcell < 1.0e6f);
        r.inv_dcell = r.leap_ok ? 0.999f / dcell : 0.0f;
        // cell index from the product q = (p-o)*(1/vs) when |q - rint(q)| < thr: |q| <= span along the whole ray, so
        // thr = 0.5 - 2^-21*span keeps q and the IEEE quotient on the same side of every rounding boundary
        r.thr = r.leap_ok ? 0.5f - span * 0x1p-21f : -1.0f;
        MarchBlock mb;
        MarchBinade bn;
#ifdef VP_DIAG
        int dbg_leap = 0, dbg_fine = 0;
#endif
        while (t < tEnd) {
            int D;
            id = march_sample(p, mv, r, mb, t, D);
            if (id != 0) break;
#ifdef VP_DIAG
            if (heavy_t < 0) { if (D >= 2) dbg_leap++; else dbg_fine++; }
            dbg_it++;
#endif
            // advance by 1 + J samples, J of them provably unable to reach an occupied cell
            march_advance(p.inc, tEnd, bn, t, march_steps(r, D));
        }
#ifdef VP_DIAG
        if (heavy_t == -2) {   // VP_FLAG_DIAG_WAVES: lanes 0-2 of the wavefront leave its start and end stamps (100 MHz clock) and its
            // iteration count (= its slowest lane's; full tiles only: every lane is here); the other lanes their own evaluation count
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            const int l = threadIdx.x & 63;
            const int mine = dbg_leap + dbg_fine;
            for (int off = 32; off > 0; off >>= 1) dbg_it = max(dbg_it, __shfl_xor(dbg_it, off));
            hit[((long long)bv * p.height + y) * p.width + x] = l == 0 ? (int)(unsigned)dbg_t0 : l == 1 ? (int)(unsigned)t1 : l == 2 ? dbg_it : mine;
            return;
        }
        if (heavy_t < 0) {   // diagnostic build only (make diag, VP_FLAG_DIAG_EVALS): per-ray evaluation counts instead of IDs
            hit[((long long)bv * p.height + y) * p.width + x] = (min(dbg_leap, 1023) << 10) | min(dbg_fine, 1023);
            return;
        }
#endif
    }
    if (id != 0 && (id < 0 || id >= p.n_rows)) {   // the reference would write out of bounds here
        atomicOr(&status[ST_BADID], 1);
        *(volatile int *)&fa.sticky[ST_STICKY_BADID] = 1;
        id = 0;
    }
    hit[((long long)bv * p.height + y) * p.width + x] = id;
    // Per-call hit histogram, aggregated per wavefront: the lanes of an 8x8 tile share a handful of voxel IDs, so
    // one lane per distinct ID adds the whole group (returning integer atomics on hot addresses were measured to
    // slow a concurrently running gather 3-4x; this issues ~8x fewer of them).  The add that lifts a voxel's
    // per-call count above heavy_t enlists it for the workgroup path.
    {
        // every lane leads at most one group (the group of its own ID), so the group sizes are collected first and
        // ALL groups are added by one wave-level atomic instruction: one memory round trip instead of one per group
        const int lane_ = threadIdx.x & 63;
        int my_n = 0;
        unsigned long long todo = __ballot(id != 0);
        if (fa.hit_waves && todo != 0ull && lane_ == __builtin_ctzll(todo))
            fa.hit_waves[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = __popcll(todo);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            const int cur = __builtin_amdgcn_readlane(id, l);
            const unsigned long long m = __ballot(id == cur);
            if (lane_ == l) my_n = __popcll(m);
            todo &= ~m;
        }
        if (my_n > 0) {
            const int old = atomicAdd(&cnt_call[id], my_n);
            if (old <= heavy_t && old + my_n > heavy_t) heavy_list[atomicAdd(&status[ST_NHEAVY], 1)] = id;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_first_hit(FirstHitArgs fa, Params p)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int y = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    first_hit_body<MODE>(fa, p, x, y, blockIdx.z);
}

}  // namespace
