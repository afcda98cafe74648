// voxproj.hip -- MI355X (gfx950) 2D -> sparse-voxel feature projector: C-ABI and launch plumbing (kernels in vp_*.h).
//
// Replaces the device path of the reference's project_features_cuda extension
// (cuda_project_image_to_sparse_voxel/project_image_cuda_kernel.cu:24-92,140-334,374-459) with a
// two-phase design written for CDNA4:
//
//   phase 1  k_first_hit   one lane per (pixel, view): the reference's ray-march replayed in the
//                          exact fp32 operation order of oracle/projector_oracle.c (no FMA
//                          contraction, IEEE divide/sqrt, t += inc) -> first-hit voxel ID image and a
//                          per-call integer hit histogram.  Pixel -> voxel assignment is bit-exact.
//   phase 2  k_gather      one 64-lane wavefront per voxel: project the voxel's cube into every
//                          view (lane = view, world->camera table from k_worklist's trailing workgroups), scan the small pixel box in the ID
//                          image for pixels that first-hit THIS voxel ("occlusion test"), and stream
//                          their C-wide feature rows from HBM with 16-byte-per-lane coalesced loads,
//                          accumulating in registers in (view, y, x) order; one non-atomic
//                          read-modify-write of the output row; hit count by ballot/popcount.
//                          The box is only a search hint: the per-call histogram of phase 1 is the
//                          ground truth, and a voxel whose box scan finds fewer pixels than phase 1
//                          counted is rescanned over the whole image.
//
// No float atomics (deterministic sums), no MFMA (the path is gather/accumulate, HBM-bound).
//
// One translation unit: this file holds the host side (launch plumbing, C-ABI entry points) and includes
//   vp_common.h  error text, timing spans, VP_FLAG_PIPELINE stream state, Params, workspace Layout
//   vp_tables.h  arithmetic contract helpers, occupancy-derived tables, view table
//   vp_march.h   phase 1 (k_first_hit)
//   vp_gather.h  work list and phase 2 (k_worklist, k_gather: one wavefront per voxel, one workgroup per heavy voxel)
//   vp_aux.h     RGB projection, nearest-voxel map, streaming-read probe
//   vp_prep.h    feature-map up-sampler (PTD:119-127), occupancy builder (BSO:30-53)
//   vp_aggregate.h  the aggregator's per-view fp16 accumulate over the hit rows (AGG:307-313)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "voxproj.h"

#include "vp_common.h"
#include "vp_tables.h"
#include "vp_march.h"
#include "vp_gather.h"
#include "vp_aux.h"
#include "vp_prep.h"
#include "vp_aggregate.h"

// ------------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------------
// VEC_OK: 0 = scalar fp32 path, 1 = 16-byte vector fp32 path, 2 = fp16 feature maps
#define VP_DISPATCH_KVU(KERNEL, VEC_OK, C, ...)                                   \
    do {                                                                          \
        if ((VEC_OK) == 2) hipLaunchKernelGGL((KERNEL<1, 8, VP_F16_U>), __VA_ARGS__);    \
        else if ((VEC_OK) && (C) > 256) hipLaunchKernelGGL((KERNEL<2, 4, 4>), __VA_ARGS__); \
        else if (VEC_OK) hipLaunchKernelGGL((KERNEL<1, 4, 4>), __VA_ARGS__);      \
        else hipLaunchKernelGGL((KERNEL<4, 1, 4>), __VA_ARGS__);                  \
    } while (0)
// the same for k_gather, whose fourth template argument is the number of views whose first ID tile is fetched together
// (G32: 1 or 4 for fp32 rows, vp_gather.h)
#define VP_DISPATCH_GATHER(G32, VEC_OK, C, ...)                                   \
    do {                                                                          \
        if ((VEC_OK) == 2) hipLaunchKernelGGL((k_gather<1, 8, VP_F16_U, GATHER_G16>), __VA_ARGS__);    \
        else if ((VEC_OK) && (C) > 256) hipLaunchKernelGGL((k_gather<2, 4, 4, G32>), __VA_ARGS__); \
        else if (VEC_OK) hipLaunchKernelGGL((k_gather<1, 4, 4, G32>), __VA_ARGS__);      \
        else hipLaunchKernelGGL((k_gather<4, 1, 4, 1>), __VA_ARGS__);               \
    } while (0)

// the one-view gather (vp_gather.h, k_gather_one).  SMALL: the view has few pixels (up to GATHER_G32_SMALL_IMAGE): a voxel gets
// a handful of rows and the launch is bounded by the round trips per voxel, not by bandwidth -- 8 rows in flight per
// wavefront instead of 4 (3 wavefronts per SIMD instead of 4): one R1 view 86.4 -> 75.5 us; one R2 view 225 -> 228 us, so
// large views keep 4 (profiles/r04_one_view_gather.log)
#define VP_DISPATCH_GATHER_ONE(SMALL, VEC_OK, C, ...)                             \
    do {                                                                          \
        if ((VEC_OK) == 2) hipLaunchKernelGGL((k_gather_one<1, 8, VP_F16_U>), __VA_ARGS__);    \
        else if ((VEC_OK) && (C) > 256 && (SMALL)) hipLaunchKernelGGL((k_gather_one<2, 4, 8>), __VA_ARGS__); \
        else if ((VEC_OK) && (C) > 256) hipLaunchKernelGGL((k_gather_one<2, 4, 4>), __VA_ARGS__); \
        else if (VEC_OK) hipLaunchKernelGGL((k_gather_one<1, 4, 4>), __VA_ARGS__);      \
        else hipLaunchKernelGGL((k_gather_one<4, 1, 4>), __VA_ARGS__);                  \
    } while (0)

// rows in flight per wavefront in the fp16 gather
#ifndef VP_F16_U
#define VP_F16_U 4
#endif

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

int vp_abi_version(void) { return VP_ABI_VERSION; }

const char *vp_last_error(void) { return g_err; }

size_t vp_workspace_bytes(int B, int V, int H, int W, int C, int dimz, int dimy, int dimx, int64_t n_rows)
{
    if (B <= 0 || V <= 0 || H <= 0 || W <= 0 || C <= 0 || n_rows <= 0 || dimz <= 0 || dimy <= 0 || dimx <= 0) return 0;
    return make_layout(B, V, H, W, C, n_rows, dimz, dimy, dimx).total;
}

static int workspace_status_impl(void *workspace, void *stream_, bool drain_all);

static int project_impl(const float *feats, bool feats_f16, const int64_t *occ, const float *vmi, const float *intr,
                        const float *opts_host, int32_t *count, float *out, int32_t *views_hit,
                        const float *grid_origin_host, float voxel_size,
                        int B, int V, int H, int W, int C, int dimz, int dimy, int dimx, int64_t n_rows,
                        void *workspace, size_t workspace_bytes, void *stream_, int flags)
{
    if (feats_f16 && (C % 8 != 0 || ((uintptr_t)feats & 15) != 0 || ((uintptr_t)out & 15) != 0))
        return fail(VP_EINVAL, "fp16 feature maps need C %% 8 == 0 and 16-byte aligned feats/out");
    if (!feats || !occ || !vmi || !intr || !opts_host || !count || !out || !grid_origin_host || !workspace)
        return fail(VP_EINVAL, "null pointer argument");
    // What VP_FLAG_GATHER_ONLY and vp_copy_hit_image rely on (first-hit images of the last call, its arguments) is valid only
    // once a call has queued all of its launches: withdrawn when a call fails -- refused arguments, refused flag, HIP error --
    // so that a later gather-only call cannot match the call before the failed one
    WsState *st = ws_state(workspace, true);
    struct HitGuard { WsState *st; bool keep; ~HitGuard() { if (!keep) st->has_hit = false; } } hit_guard{st, false};
    if (B <= 0 || V <= 0 || H <= 0 || W <= 0 || C <= 0 || dimz <= 0 || dimy <= 0 || dimx <= 0 || n_rows <= 0)
        return fail(VP_EINVAL, "non-positive dimension");
    if ((long long)B * V > 65535) return fail(VP_EINVAL, "B*V = %lld exceeds 65535", (long long)B * V);
    if ((long long)dimz * dimy * dimx >= (1ll << 31)) return fail(VP_EINVAL, "occupancy grid has >= 2^31 cells per batch");
    if ((long long)H * W >= (1ll << 31) || n_rows >= (1ll << 31)) return fail(VP_EINVAL, "image or row count >= 2^31");
    if ((flags & VP_FLAG_SYNC) && (flags & VP_FLAG_PIPELINE)) return fail(VP_EINVAL, "VP_FLAG_SYNC and VP_FLAG_PIPELINE exclude each other");
    Params p;
    p.width = (int)(opts_host[0] + 0.5f);    // K.cu:403
    p.height = (int)(opts_host[1] + 0.5f);   // K.cu:404
    if (p.width != W || p.height != H)
        return fail(VP_EINVAL, "opts width/height (%d,%d) must equal the feature map's (%d,%d)", p.width, p.height, W, H);
    p.dmin = opts_host[2]; p.dmax = opts_host[3]; p.inc = opts_host[4];
    if (!(p.inc > 0.0f)) return fail(VP_EINVAL, "rayIncrement must be > 0 (the reference would never terminate)");
    p.ox = grid_origin_host[0]; p.oy = grid_origin_host[1]; p.oz = grid_origin_host[2];
    p.vs = voxel_size;
    p.dimz = dimz; p.dimy = dimy; p.dimx = dimx;
    p.B = B; p.V = V; p.C = C; p.n_rows = n_rows;

    const Layout l = make_layout(B, V, H, W, C, n_rows, dimz, dimy, dimx, workspace_bytes);
    if (workspace_bytes < l.total) return fail(VP_EWORKSPACE, "workspace has %zu bytes, need %zu", workspace_bytes, l.total);
    if ((uintptr_t)workspace & 255) return fail(VP_EWORKSPACE, "workspace must be 256-byte aligned");
    // (every check above is host arithmetic; from here on the device is touched)
    if (!sticky_open(*st)) return fail(VP_EHIP, "could not allocate the workspace record's page of pinned host memory (sticky error words)");
    char *ws = (char *)workspace;
    hipStream_t s0 = (hipStream_t)stream_;

    // buffer set and streams: plain calls use set 0 on the caller's stream only; pipelined calls alternate sets
    // and run phase 1 on the side stream
    const bool pipe = (flags & VP_FLAG_PIPELINE) != 0;
    PipeState *ps = &st->pipe;
    if (pipe && !pipe_open(*ps)) return fail(VP_EHIP, "could not create the side stream / events for VP_FLAG_PIPELINE");
    // VP_FLAG_GATHER_ONLY: phase 2 once more, on another row range, from what the previous call's phase 1 left in ITS
    // buffer set; everything runs on the caller's stream, behind that call's gather
    const bool gather_only = (flags & VP_FLAG_GATHER_ONLY) != 0;
    if (gather_only) {
        if (!st->has_hit || st->last_feats != (const void *)feats || st->last_f16 != feats_f16 || st->last_B != B || st->last_V != V ||
            st->last_H != H || st->last_W != W || st->last_C != C || st->B != B || st->dimz != dimz || st->dimy != dimy ||
            st->dimx != dimx || st->n_rows != n_rows || st->last_out != (const void *)out || st->last_count != (const void *)count ||
            st->last_vmi != (const void *)vmi)
            return fail(VP_EINVAL, "VP_FLAG_GATHER_ONLY repeats phase 2 of the previous call on this workspace: there is none (or it "
                                   "failed), or its feature maps / poses / outputs / shapes differ from this call's");
        if (st->opt_row_begin < 0 && st->opt_row_end < 0)
            return fail(VP_EINVAL, "VP_FLAG_GATHER_ONLY without a row range (VP_OPT_ROW_BEGIN / VP_OPT_ROW_END) would gather every row twice");
        if (!st->last_ranged)
            return fail(VP_EINVAL, "VP_FLAG_GATHER_ONLY after a call that had no row range: that call gathered every row already");
    }

    int q = 0;
    hipStream_t s1 = s0;
    if (gather_only) {
        q = st->last_q;
    } else if (pipe) {
        q = (int)(ps->calls & 1);
        s1 = ps->side;
    } else if (ps->ok && (ps->used[0] || ps->used[1])) {
        // a plain call after pipelined ones on this workspace: drain the side streams first
        VP_HIP(hipStreamSynchronize(ps->side));
        ps->used[0] = ps->used[1] = false;
    }
    int *status = (int *)(ws + l.status[q]);
    int *cell_of_id = (int *)(ws + l.cell_of_id);
    unsigned long long *mask64 = (unsigned long long *)(ws + l.mask64);
    NearRec *near2 = (NearRec *)(ws + l.near2);
    unsigned char *dist = (unsigned char *)(ws + l.dist);
    unsigned char *dist_tmp = (unsigned char *)(ws + l.dist_tmp);
    int *cnt_call = (int *)(ws + l.cnt_call[q]);
    int *heavy_list = (int *)(ws + l.heavy[q]);
    int *work = (int *)(ws + l.work[q]);
    ViewEntry *viewtab = (ViewEntry *)(ws + l.viewtab[q]);
    int *hit = (int *)(ws + l.hit[q]);
    int *status0 = (int *)(ws + l.status[0]);      // header + sticky words live in the block of set 0

    // Occupancy-derived tables: rebuilt unless the caller vouches for them (VP_FLAG_REUSE_ACCEL) or asks for a
    // check (VP_FLAG_VERIFY_ACCEL, blocking calls only): then the grid is compared with the copy the tables were
    // built from, and they are rebuilt only if a cell changed.
    const long long cells = (long long)dimz * dimy * dimx;
    int *occ_copy = (int *)(ws + l.occ_copy);
    WsState &rec = *st;
    const bool rec_matches = rec.B == B && rec.dimz == dimz && rec.dimy == dimy && rec.dimx == dimx && rec.n_rows == n_rows;
    const bool verify = (flags & VP_FLAG_VERIFY_ACCEL) && !pipe && !gather_only && !(flags & VP_FLAG_REUSE_ACCEL);
    const int cmp_blocks = (int)((cells * B + 255) / 256 > 8192 ? 8192 : (cells * B + 255) / 256);
    bool rebuild = !gather_only && !(flags & VP_FLAG_REUSE_ACCEL);
    // VP_FLAG_REUSE_ACCEL is a promise about the tables in THIS workspace: refuse it when the library never built
    // them here (fresh or recycled memory) or built them for another grid shape / row count -- the march would leap on
    // garbage and silently miss hits
    if (!rebuild && !gather_only && (rec.builds == 0 || !rec_matches))
        return fail(VP_EINVAL, "VP_FLAG_REUSE_ACCEL, but this workspace holds no occupancy tables for a grid of this shape "
                               "(B, dims, n_rows): call once without the flag");
    if (rebuild || !rec.opened) {
        // The workspace header (magic, this record's generation) and the sticky error words: initialised by the first call
        // of a record and whenever the memory does not carry this record's generation (any more) -- decided on the
        // device, no read-back.  A call that trusts the tables never initialises: it must find the header intact.
        if (pipe && ps->ok) {
            VP_HIP(hipStreamSynchronize(ps->side));
        }
        hipLaunchKernelGGL(k_ws_open, dim3(1), dim3(64), 0, s0, status0, (int *)(ws + l.status[1]), WS_MAGIC, rec.gen, rebuild ? 1 : 0);
        rec.opened = true;
    }
    if (verify && rec_matches && rec.copy_valid) {
        // the verdict comes back through the record's page of pinned host memory (no memset, no device-to-host copy)
        volatile int *differs = rec.sticky_host + ST_OCCDIFF;
        *differs = 0;
        hipLaunchKernelGGL(k_occ_compare_copy, dim3(cmp_blocks), dim3(256), 0, s0, (const long long *)occ, occ_copy, cells * B, rec.sticky_dev + ST_OCCDIFF);
        VP_HIP(hipStreamSynchronize(s0));
        rebuild = *differs != 0;      // the copy is already up to date either way
    } else if (rebuild) {
        if (verify) {
            // first checked call on this workspace / new shape: take the copy now
            hipLaunchKernelGGL(k_occ_compare_copy, dim3(cmp_blocks), dim3(256), 0, s0, (const long long *)occ, occ_copy, cells * B, status + ST_OCCDIFF);
            rec.copy_valid = true;
        } else {
            rec.copy_valid = false;  // tables rebuilt without refreshing the copy
        }
    }
    if (rebuild) {
        // the tables are shared by both buffer sets: nothing of an earlier call may still be running
        if (pipe) {
            VP_HIP(hipStreamSynchronize(ps->side));
            VP_HIP(hipStreamSynchronize(s0));
        }
        ProfSpan sp; sp.begin(0, s0);
        VP_HIP(hipMemsetAsync(cell_of_id, 0xFF, size_t(B) * n_rows * sizeof(int), s0));
        VP_HIP(hipMemsetAsync(mask64, 0, size_t(B) * l.nblk * sizeof(unsigned long long), s0));
        const int blocks = (int)((cells * B + 255) / 256 > 16384 ? 16384 : (cells * B + 255) / 256);
        hipLaunchKernelGGL(k_build_cells, dim3(blocks), dim3(256), 0, s0, (const long long *)occ, cell_of_id,
                           mask64, dimz, dimy, dimx, l.nby, l.nbx, l.nblk, B, (long long)n_rows);
        const int db = (int)(((long long)l.nbz * l.nby * l.nbx * B + 255) / 256);
        hipLaunchKernelGGL(k_block_dist, dim3(db), dim3(256), 0, s0, mask64, (const unsigned char *)nullptr, dist, l.nbz, l.nby, l.nbx, l.nblk, B, 0);
        hipLaunchKernelGGL(k_block_dist, dim3(db), dim3(256), 0, s0, mask64, (const unsigned char *)dist, dist_tmp, l.nbz, l.nby, l.nbx, l.nblk, B, 1);
        hipLaunchKernelGGL(k_block_dist, dim3(db), dim3(256), 0, s0, mask64, (const unsigned char *)dist_tmp, dist, l.nbz, l.nby, l.nbx, l.nblk, B, 2);
        const long long near_waves = (long long)l.nbz * l.nby * l.nbx * B;
        hipLaunchKernelGGL(k_build_near, dim3((unsigned)((near_waves + 3) / 4)), dim3(256), 0, s0, mask64, (const unsigned char *)dist,
                           near2, dimz, dimy, dimx, l.nbz, l.nby, l.nbx, l.nblk, B);
        rec.B = B; rec.dimz = dimz; rec.dimy = dimy; rec.dimx = dimx; rec.n_rows = n_rows;
        rec.builds++;
        // seal: the header now names the tables this memory holds
        hipLaunchKernelGGL(k_ws_seal, dim3(1), dim3(1), 0, s0, status0, tables_key(B, dimz, dimy, dimx, n_rows, rec.builds));
        sp.end();
        if (pipe) VP_HIP(hipStreamSynchronize(s0));   // rare: the side stream must see the finished tables
    }
    const unsigned expect_tables = tables_key(rec.B, rec.dimz, rec.dimy, rec.dimx, rec.n_rows, rec.builds);

    if (pipe && !gather_only) {
        // set q was last used two calls ago: its gather must be over before phase 1 overwrites hit/cnt
        if (ps->used[q]) VP_HIP(hipStreamWaitEvent(s1, ps->call_done[q], 0));
    }
    // Row range of phase 2 (VP_OPT_ROW_BEGIN / _END).  The heavy list is the march's, i.e. the whole call's: the workgroup
    // role of a ranged gather skips the listed IDs outside its range, so the gathers of a split call share the list without
    // summing a voxel twice -- and every voxel is summed by the same role (and so to the same bits) as in the unsplit call.
    const bool ranged = rec.opt_row_begin >= 0 || rec.opt_row_end >= 0;
    const long long row_lo = std::max<long long>(1, rec.opt_row_begin);
    const long long row_hi = rec.opt_row_end < 0 ? (long long)n_rows : std::min<long long>(rec.opt_row_end, (long long)n_rows);

    // ---- phase 1 (on s1) ----
    // More pixels than heavy_t in one call -> the voxel is not one wavefront's job: it is cut into parts of part_px pixels
    // (vp_gather.h, "Split voxels").  Calls of more than one view: both numbers default to min(256 + 64*B*V, 2048) -- the longest
    // item a wavefront can be handed bounds the tail of the launch; the last items run on an emptying machine at ~4 GB/s per
    // wavefront, 2048 rows of 2 KiB in a millisecond (sweep of 512 ... 4096: profiles/r05_ab_split_voxels.log, fp16 calls:
    // r06_f16_part_slots.log).  One-view calls: the device sizes both from the view's hit total (PlanArgs); 256 + 64 is the
    // threshold of round 5's workgroup role, kept as the A/B arm (VP_OPT_ONE_VIEW_SPLIT = 0).
    int heavy_t = (int)std::min<long long>(256 + 64ll * B * V, 2048);
    if ((long long)B * V == 1) heavy_t = 256 + 64;
    if (rec.opt_heavy_t > 0) heavy_t = (int)std::min<long long>(rec.opt_heavy_t, 2147483647ll);   // VP_OPT_HEAVY_THRESHOLD
    if (flags & VP_FLAG_SERIAL_SUMS) heavy_t = 2147483647;
    // One-view calls (the drop-in module's, the parity aggregator's) take the one-view gather: a fixed grid of wavefronts
    // dealt the size-ordered list, a wavefront's boxes computed one voxel per lane, the next voxel's tile and row fetched
    // under the current voxel's rows (vp_gather.h, k_gather_one).  VP_OPT_ONE_VIEW_GATHER = 0 keeps k_gather as the A/B arm.
    const bool one_view = (long long)B * V == 1 && rec.opt_one_view != 0;
    // The parts' partial rows live in the buffer set's part slots, and a call's parts must never outnumber them: a split voxel
    // has c > heavy_t >= part_px pixels and P = ceil(c / part_px) <= 2c / part_px parts, the c of a call add up to at most
    // B*V*H*W, so part_px >= 2*B*V*H*W / slots is enough -- both values are raised to that bound (only calls larger than the
    // bench's are: 65536 slots allow parts of 2048 pixels up to 67 M pixels per call).
    PlanArgs plan;
    plan.heavy_t = heavy_t; plan.part_t = 2147483647; plan.part_px = 0; plan.count_heavy = 1; plan.dyn_px_min = 0; plan.dyn_t_ratio = 0;
    plan.dyn_t_floor = 0; plan.cell_in_item = 0;
    const long long px2 = 2ll * B * V * (long long)H * W;
    // One-view calls (round 6) cut their large voxels into parts too -- one wavefront of k_gather_one per part, k_combine_parts
    // behind it -- and size the parts on the device from the view's hit total (PlanArgs, vp_gather.h).  VP_OPT_ONE_VIEW_SPLIT = 0
    // keeps round 5's path (a workgroup per voxel above 320 pixels) as the A/B arm.
    const bool one_split = one_view && heavy_t != 2147483647 && rec.opt_one_view_split != 0;
    if (!one_view && heavy_t != 2147483647) {
        long long ppx = rec.opt_part_px > 0 ? rec.opt_part_px : std::max<long long>(1, heavy_t);     // VP_OPT_PART_PIXELS
        ppx = std::max(ppx, (px2 + l.slot_cap - 1) / l.slot_cap);
        plan.part_px = (int)std::min<long long>(ppx, 2147483647ll);
        heavy_t = std::max(heavy_t, plan.part_px);
        plan.heavy_t = plan.part_t = heavy_t;
    } else if (one_split) {
        // fixed numbers where the options give them (VP_OPT_ONE_VIEW_SPLIT, else VP_OPT_HEAVY_THRESHOLD; VP_OPT_PART_PIXELS), raised
        // to the slot bound like those of multi-view calls; otherwise the device's
        const long long T = rec.opt_one_view_split > 0 ? rec.opt_one_view_split : rec.opt_heavy_t > 0 ? rec.opt_heavy_t : 0;
        long long ppx = rec.opt_part_px > 0 ? std::max(rec.opt_part_px, (px2 + l.slot_cap - 1) / l.slot_cap) : 0;
        if (ppx == 0 && T > 0) ppx = std::max<long long>((T + ONE_VIEW_T_RATIO - 1) / ONE_VIEW_T_RATIO, (px2 + l.slot_cap - 1) / l.slot_cap);
        plan.part_px = (int)std::min<long long>(ppx, 2147483646ll);
        plan.dyn_px_min = ppx > 0 ? 0 : ONE_VIEW_PART_MIN;
        plan.heavy_t = plan.part_t = T > 0 ? (int)std::min<long long>(std::max(T, ppx), 2147483646ll) : 0;
        plan.dyn_t_ratio = T > 0 ? 0 : ONE_VIEW_T_RATIO;
        plan.dyn_t_floor = (long long)H * W <= GATHER_G32_SMALL_IMAGE ? ONE_VIEW_T_FLOOR_SMALL : 0;
        plan.cell_in_item = 1;
        heavy_t = plan.heavy_t;
    } else if (one_view) {
        plan.count_heavy = 0;      // the march enlists and counts the voxels above heavy_t
    }
    static_assert(sizeof(PlanArgs) == sizeof(int) * 8, "PlanArgs is kept as ints in the workspace record");
    if (gather_only) { memcpy(&plan, st->last_plan, sizeof(plan)); heavy_t = plan.heavy_t; }      // the thresholds of the call whose march is reused
    const int part_px = plan.part_px;
    const bool plans_parts = part_px > 0 || plan.dyn_px_min > 0;
#ifdef VP_DIAG
    if (flags & VP_FLAG_DIAG_EVALS) heavy_t = -1;      // diagnostic build only: the hit image then holds evaluation counts
    if (flags & VP_FLAG_DIAG_WAVES) heavy_t = -2;      // ... per-wavefront clock stamps
#endif
    const int wl_blocks = (int)((n_rows + 256 * WL_PER_THREAD - 1) / (256 * WL_PER_THREAD));
    int4 *parts = (int4 *)(ws + l.parts[q]), *split = (int4 *)(ws + l.split[q]), *pmeta = (int4 *)(ws + l.pmeta[q]);
    float *prow = (float *)(ws + l.prow[q]);
    int *hit_waves = (int *)(ws + l.hitcnt[q]);
#define VP_LAUNCH_WORKLIST(STREAM)                                                                                              \
    hipLaunchKernelGGL(k_worklist, dim3((unsigned)(wl_blocks + (B * V + 255) / 256)), dim3(256), 0, STREAM, (const int *)cnt_call,   \
                       plan, (long long)n_rows, work, status, wl_blocks, vmi, viewtab, B * V, row_lo, row_hi, parts, split, \
                       (int)l.slot_cap, rec.sticky_dev, (const int *)cell_of_id, (const int *)hit_waves, (int)l.n_hitcnt)
    if (gather_only) {
        // the work list of the new row range, from the histogram the previous call's march left
        VP_HIP(hipMemsetAsync(status + ST_WORK0, 0, ST_PLAN_WORDS * sizeof(int), s0));
        VP_LAUNCH_WORKLIST(s0);
    }
    if (!gather_only) {
        ProfSpan sp; sp.begin(0, s1);
        // one launch clears the per-call status words and the per-call histogram, and checks the workspace header
        hipLaunchKernelGGL(k_zero_call, dim3((unsigned)((n_rows + 1023) / 1024)), dim3(256), 0, s1, status, cnt_call, (long long)n_rows,
                           status0, rec.sticky_dev, WS_MAGIC, rec.gen, expect_tables, hit_waves, plan.dyn_px_min > 0 ? l.n_hitcnt : 0ll);
        sp.end();
    }
    if (!gather_only) {
        FirstHitArgs fa;
        fa.occ = (const long long *)occ; fa.vmi = vmi; fa.intr = intr; fa.near2 = near2; fa.dist = dist;
        fa.nby = l.nby; fa.nbx = l.nbx; fa.nblk = l.nblk; fa.hit = hit; fa.cnt_call = cnt_call;
        fa.heavy_list = heavy_list; fa.heavy_t = ((one_view && !one_split) || heavy_t < 0) ? heavy_t : 2147483647;      // (the march enlists heavy voxels for one-view calls without parts only)
        fa.hit_waves = (plan.dyn_px_min > 0 && l.n_hitcnt > 0) ? hit_waves : nullptr;
        fa.status = status; fa.sticky = rec.sticky_dev;
        const dim3 grid((W + 15) / 16, (H + 15) / 16, B * V);
        ProfSpan sp; sp.begin(1, s1);
        if (flags & VP_FLAG_EXACT_MARCH) {
            hipLaunchKernelGGL(k_first_hit<0>, grid, dim3(256), 0, s1, fa, p);
        } else {
            // Occupancy shaping for the pipelined mode: a 41-KiB dynamic-LDS reservation (the kernel does not touch
            // it) admits at most 3 march workgroups = 12 wavefronts per CU.  Spread that thin the march still
            // finishes under the gather of the previous call (40 ms vs 50 ms per R2 pass) and costs the gather
            // ~1 % instead of ~8 % (measured: mean 55.3 -> 53.5 ms per pass); alone it runs unrestricted.
            // Only while the previous call's gather is still queued or running: behind an idle GPU (first call of
            // a job, or after the caller synchronised) the march has nothing to spare and runs unrestricted.
            bool beside_gather = false;
            if (pipe && ps->used[q ^ 1]) {
                beside_gather = hipEventQuery(ps->call_done[q ^ 1]) == hipErrorNotReady;
                (void)hipGetLastError();   // hipErrorNotReady is an answer, not a failure
            }
            // Rows of up to 1 KiB (fp16 maps of 512 channels, fp32 maps of 256): the gather moves half the bytes per view, so a
            // march held to 3 workgroups per CU takes longer than the gather it hides under and becomes the critical path
            // (30.1 vs 29.2 ms per fp16 pass); 5 workgroups per CU (30 KiB) bring the pass from 33.1-33.7 to 29.5-31.6 ms,
            // 6 and 4 are worse (profiles/r03_march_occupancy_cap_sweep.log).
            const size_t row_bytes = size_t(C) * (feats_f16 ? 2 : 4);
            size_t lds_req = beside_gather ? (row_bytes <= 1024 ? 30 : 41) * 1024 : 0;
            if (rec.opt_march_lds_kb >= 0) lds_req = size_t(std::min<long long>(rec.opt_march_lds_kb, 64)) * 1024;   // VP_OPT_MARCH_LDS_KB
            hipLaunchKernelGGL(k_first_hit<1>, grid, dim3(256), lds_req, s1, fa, p);
        }
        // the gather's work list: touched voxels by size class, largest first (needs the finished histogram); its trailing
        // workgroups compute the view table, which is phase 2's too -- behind the march, not in front of it (in pipelined
        // mode a kernel with that many registers waits for a wavefront of the previous call's gather to retire)
        VP_LAUNCH_WORKLIST(s1);
        sp.end();
    }
    if (pipe && !gather_only) VP_HIP(hipEventRecord(ps->fh_done[q], s1));

    // ---- phase 2 ----
    GatherArgs g;
    g.feats = feats; g.hit = hit; g.viewtab = viewtab; g.intr = intr; g.cell_of_id = cell_of_id;
    g.cnt_call = cnt_call; g.heavy_list = heavy_list; g.n_heavy = status + ST_NHEAVY;
    g.row_lo = (int)row_lo; g.row_hi = (int)row_hi;
    g.work = work; g.work_n = status + ST_WORK0;
    g.parts = parts; g.split = split; g.pmeta = pmeta; g.prow = prow;
    g.host_word = nullptr; g.host_seq = 0;
    g.parts_on = (one_view && plans_parts) ? 1 : 0; g.slot_cap = (int)l.slot_cap; g.count = count; g.views_hit = views_hit; g.out = out; g.status = status;
    const int vec_ok = feats_f16 ? 2 : ((C % 4 == 0) && (((uintptr_t)feats & 15) == 0) && (((uintptr_t)out & 15) == 0)) ? 1 : 0;
    // grouped ID-tile fetch on small images needs views to group
    const bool small_image = (long long)B * V >= 8 && (long long)H * W <= GATHER_G32_SMALL_IMAGE;
    g.heavy_blocks = 0;
    if (pipe && !gather_only) VP_HIP(hipStreamWaitEvent(s0, ps->fh_done[q], 0));
    if (one_view) {
        // a fixed number of workgroups per CU.  The kernel's registers admit 4 at a time; 16 are launched, so that the
        // dispatcher evens out what the static deal leaves uneven (one R2 view: 2 / 4 / 8 / 16 / 32 / 64 per CU -> 229 / 219 /
        // 210-226 / 217 / 220 / 222 us, R1: 99 / 95 / 87 / 83 / 83.5 / 82.6 us, profiles/r04_one_view_gather.log) -- a
        // quarter of the workgroups k_gather launches for the same call, none of them without work.
        // VP_OPT_ONE_VIEW_GATHER = n > 0 overrides it.
        ProfSpan sp; sp.begin(2, s0);
        const int per_cu = rec.opt_one_view > 0 ? (int)std::min<long long>(rec.opt_one_view, 256) : 16;
        // (values from 1000 on: a grid of exactly n - 1000 workgroups -- tests walk the batches of 64 entries per wavefront)
        const long long want = rec.opt_one_view >= 1000 ? rec.opt_one_view - 1000 : (long long)device_cus() * per_cu;
        const long long cap = (n_rows - 1 + 3) / 4;           // never more wavefronts than voxel IDs
        const unsigned nblk = (unsigned)std::max<long long>(1, std::min(want, cap));
        // every workgroup of the grid takes heavy voxels first (round 5; rounds 1-4: the first 128): on a close-up frame EVERY
        // voxel of the view is heavy -- 300-400 of them -- and 128 workgroups summed them three apiece while the rest of the grid
        // had nothing to deal (0.70 ms per call instead of 0.3, profiles/r05_dropin_trajectory.log)
        g.heavy_blocks = (heavy_t != 2147483647 && !plans_parts) ? (int)nblk : 0;
        // A BLOCKING call (the drop-in module's) launches k_combine_parts only if the view has split voxels: most frames of a
        // walk through a room have none, and the empty launch is 6-7 us of a 0.14-0.3 ms call.  The gather's first wavefront
        // writes the count (final since k_worklist) into the record's pinned page, tagged with this call's sequence number; the
        // host, which would otherwise sleep in the stream synchronise, reads it a few microseconds into the gather -- long before
        // the gather ends.  Nothing depends on the note arriving: without it (2 ms) the launch goes out as for any other call.
        volatile int *note = nullptr;
        if ((flags & VP_FLAG_SYNC) && plans_parts && n_rows > 1) {
            rec.split_seq = (rec.split_seq + 1) & 0x7fffu;
            note = rec.sticky_host + ST_HOST_NSPLIT;
            *note = 0;
            g.host_word = rec.sticky_dev + ST_HOST_NSPLIT; g.host_seq = (int)rec.split_seq;
        }
        if (n_rows > 1) VP_DISPATCH_GATHER_ONE((long long)H * W <= GATHER_G32_SMALL_IMAGE, vec_ok, C, dim3(nblk), dim3(256), 0, s0, g, p);
        sp.end();
        bool combine = n_rows > 1 && plans_parts;
        if (combine && note) {
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spin = 0;; spin++) {
                const unsigned v = (unsigned)*note;
                if ((v >> 31) && ((v >> 16) & 0x7fffu) == rec.split_seq) { combine = (v & 0xffffu) != 0; break; }
                if ((spin & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
                __builtin_ia32_pause();
            }
        }
        if (combine) {
            // the split voxels' partial rows -> their rows in `out`
            ProfSpan sc; sc.begin(3, s0);
            const dim3 cgrid((unsigned)std::max<long long>(1, std::min<long long>(COMBINE_BLOCKS, l.slot_cap / 2)));
            VP_DISPATCH_KVU(k_combine_parts, vec_ok, C, cgrid, dim3(256), 0, s0, g, p);
            sc.end();
        }
    } else if (n_rows > 1) {
        // one wavefront per item of the work list: at most one item per voxel row that is not split, plus the parts
        // (the parts of a call: at most 2 * pixels / part_px, and never more than the slots)
        const long long items = (n_rows - 1) + (part_px > 0 ? std::min<long long>(l.slot_cap, px2 / part_px + 1) : 0);
        {
            ProfSpan sp; sp.begin(2, s0);
            const dim3 ggrid((unsigned)((items + 3) / 4));
            if (small_image) VP_DISPATCH_GATHER(4, vec_ok, C, ggrid, dim3(256), 0, s0, g, p);
            else VP_DISPATCH_GATHER(1, vec_ok, C, ggrid, dim3(256), 0, s0, g, p);
            sp.end();
        }
        if (part_px > 0) {
            // the split voxels' partial rows -> their rows in `out`, in slot order (with VP_FLAG_SERIAL_SUMS nothing is split:
            // the launch is left out)
            ProfSpan sp; sp.begin(3, s0);
            const dim3 cgrid((unsigned)std::min<long long>(COMBINE_BLOCKS, std::max<long long>(1, l.slot_cap / 2)));
            VP_DISPATCH_KVU(k_combine_parts, vec_ok, C, cgrid, dim3(256), 0, s0, g, p);
            sp.end();
        }
    }
    if (pipe) {
        // (a gather-only call re-records the event of the set it shares with its predecessor and does not advance the sets)
        VP_HIP(hipEventRecord(ps->call_done[q], s0));
        ps->used[q] = true;
        ps->last_q = q;
        if (!gather_only) ps->calls++;
    } else if (!gather_only) {
        ps->last_q = 0;
    }
    VP_HIP(hipGetLastError());
    st->last_B = B; st->last_V = V; st->last_H = H; st->last_W = W; st->last_C = C; st->last_q = q;
    st->last_f16 = feats_f16; st->last_feats = (const void *)feats; st->last_out = (const void *)out; st->last_count = (const void *)count;
    st->last_vmi = (const void *)vmi; st->last_ranged = ranged; memcpy(st->last_plan, &plan, sizeof(plan));
    st->has_hit = true; st->hit_off = l.hit[q];
    hit_guard.keep = true;
    if (flags & VP_FLAG_SYNC) return workspace_status_impl(workspace, stream_, true);
    return VP_OK;
}

int vp_project_features(const float *feats, const int64_t *occ, const float *vmi, const float *intr,
                        const float *opts_host, int32_t *count, float *out, int32_t *views_hit,
                        const float *grid_origin_host, float voxel_size,
                        int B, int V, int H, int W, int C, int dimz, int dimy, int dimx, int64_t n_rows,
                        void *workspace, size_t workspace_bytes, void *stream_, int flags)
{
    return project_impl(feats, false, occ, vmi, intr, opts_host, count, out, views_hit, grid_origin_host, voxel_size,
                        B, V, H, W, C, dimz, dimy, dimx, n_rows, workspace, workspace_bytes, stream_, flags);
}

int vp_project_features_f16(const void *feats_f16, const int64_t *occ, const float *vmi, const float *intr,
                            const float *opts_host, int32_t *count, float *out, int32_t *views_hit,
                            const float *grid_origin_host, float voxel_size,
                            int B, int V, int H, int W, int C, int dimz, int dimy, int dimx, int64_t n_rows,
                            void *workspace, size_t workspace_bytes, void *stream_, int flags)
{
    return project_impl((const float *)feats_f16, true, occ, vmi, intr, opts_host, count, out, views_hit,
                        grid_origin_host, voxel_size, B, V, H, W, C, dimz, dimy, dimx, n_rows, workspace,
                        workspace_bytes, stream_, flags);
}

static int read_status(void *workspace, hipStream_t stream, int *st /* [2][ST_WORDS] */)
{
    if (WsState *rec = ws_state(workspace, false)) {
        if (rec->pipe.ok) VP_HIP(hipStreamSynchronize(rec->pipe.side));
    }
    VP_HIP(hipMemcpyAsync(st, workspace, 2 * align256(ST_WORDS * sizeof(int)), hipMemcpyDeviceToHost, stream));
    VP_HIP(hipStreamSynchronize(stream));
    return VP_OK;
}

int vp_nearest_voxel(const float *pts_sorted, const int32_t *perm, const int32_t *cell_start, const double *grid_origin3,
                     double cell_size, int nx, int ny, int nz, const float *queries, int64_t M, int64_t *out,
                     void *stream_)
{
    if (!pts_sorted || !perm || !cell_start || !grid_origin3 || !queries || !out) return fail(VP_EINVAL, "null pointer argument");
    if (!(cell_size > 0.0) || nx <= 0 || ny <= 0 || nz <= 0 || M < 0) return fail(VP_EINVAL, "bad grid or query count");
    if (M == 0) return VP_OK;
    hipLaunchKernelGGL(k_nearest_voxel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, pts_sorted,
                       (const int *)perm, (const int *)cell_start, grid_origin3[0], grid_origin3[1], grid_origin3[2],
                       cell_size, nx, ny, nz, queries, (long long)M, (long long *)out);
    VP_HIP(hipGetLastError());
    return VP_OK;
}

int vp_stream_read(const float *src, int64_t n_floats, float *sink, void *stream_)
{
    if (!src || !sink || n_floats < 4) return fail(VP_EINVAL, "bad argument");
    hipLaunchKernelGGL(k_stream_read, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream_, src, (long long)(n_floats / 4), sink);
    VP_HIP(hipGetLastError());
    return VP_OK;
}

// drain_all: report the highest-priority pending condition, name the others in the message and clear ALL of them (the end of
// a VP_FLAG_SYNC call: every pending error belongs to this call or to asynchronous ones before it, and a word left behind
// would fail the next, valid blocking call); otherwise ONE condition per read, only the reported word is cleared.
static int workspace_status_impl(void *workspace, void *stream_, bool drain_all)
{
    static_assert(ST_WORDS * sizeof(int) == 256, "a status block -- and the record's sticky page -- is one 256-byte slot");
    if (!workspace) return fail(VP_EINVAL, "null workspace");
    WsState *rec = ws_state(workspace, false);
    // everything queued on the workspace's streams has run when this returns: the side stream first (its work feeds the caller's)
    if (rec && rec->pipe.ok) VP_HIP(hipStreamSynchronize(rec->pipe.side));
    VP_HIP(hipStreamSynchronize((hipStream_t)stream_));
    if (!rec || !rec->opened || !rec->sticky_host) return VP_OK;   // no call has run on this workspace yet
    // the sticky words collect the errors of EVERY call since the last read (the per-call words of a buffer set are cleared
    // when the set is reused two pipelined calls later).  They live in the record's own page of pinned host memory, written
    // by the kernels through its device mapping: nothing to copy, and nobody else's to overwrite
    volatile int *sw = rec->sticky_host;
    const int stale = sw[ST_STICKY_STALE], stuck = sw[ST_STICKY_STUCK], badid = sw[ST_STICKY_BADID];
    const int word = stale ? ST_STICKY_STALE : stuck ? ST_STICKY_STUCK : badid ? ST_STICKY_BADID : -1;
    if (word >= 0) {
        if (drain_all) sw[ST_STICKY_STALE] = sw[ST_STICKY_STUCK] = sw[ST_STICKY_BADID] = 0;
        else sw[word] = 0;
    }
    const char *also = !drain_all ? "" : (word == ST_STICKY_STALE && (stuck || badid)) ? " (also pending, now cleared: stuck rays and/or out-of-range IDs)"
                                        : (word == ST_STICKY_STUCK && badid) ? " (also: a ray hit an occupancy ID outside [1, n_rows))" : "";
    if (stale) {
        rec->builds = 0;        // whatever tables the memory held are gone: VP_FLAG_REUSE_ACCEL is refused until a rebuild
        rec->copy_valid = false;
        return fail(VP_EINVAL, "VP_FLAG_REUSE_ACCEL, but the workspace memory no longer holds the tables this library built in it "
                               "(freed and handed out again without vp_workspace_release, or overwritten): those calls did no "
                               "work; call once without the flag%s", also);
    }
    if (stuck)
        return fail(VP_EINVAL, "rayIncrement is too small to advance a float32 ray parameter near depthMax: the reference "
                               "loop would never terminate (those rays were skipped, outputs are incomplete)%s", also);
    if (badid)
        return fail(VP_EBADID, "a ray hit an occupancy ID outside [1, n_rows): outputs are too small for the grid's IDs");
    return VP_OK;
}

int vp_workspace_status(void *workspace, void *stream_) { return workspace_status_impl(workspace, stream_, false); }

int vp_workspace_counters(void *workspace, int32_t *host_words, int n, void *stream_)
{
    if (!workspace || !host_words || n <= 0 || n > ST_WORDS) return fail(VP_EINVAL, "bad argument");
    int st[2 * ST_WORDS];
    int rc = read_status(workspace, (hipStream_t)stream_, st);
    if (rc != VP_OK) return rc;
    WsState *rec = ws_state(workspace, false);
    const int q = rec ? rec->pipe.last_q : 0;
    memcpy(host_words, st + q * ST_WORDS, size_t(n) * sizeof(int));
    return VP_OK;
}

int vp_profile_enable(int on)
{
    std::lock_guard<std::mutex> g(g_prof.mu);
    g_prof.on = on != 0;
    g_prof.used = 0;
    return VP_OK;
}

int vp_profile_read(double *ms4, int64_t *launches4)
{
    if (!ms4 || !launches4) return fail(VP_EINVAL, "null pointer argument");
    std::lock_guard<std::mutex> g(g_prof.mu);
    for (int k = 0; k < 4; k++) { ms4[k] = 0.0; launches4[k] = 0; }
    for (size_t i = 0; i < g_prof.used; i++) {
        VP_HIP(hipEventSynchronize(g_prof.pool[i * 2 + 1]));
        float ms = 0.f;
        VP_HIP(hipEventElapsedTime(&ms, g_prof.pool[i * 2], g_prof.pool[i * 2 + 1]));
        ms4[g_prof.kind[i]] += ms;
        launches4[g_prof.kind[i]] += 1;
    }
    g_prof.used = 0;
    return VP_OK;
}

int vp_copy_hit_image(const void *workspace, int32_t *dst, int B, int V, int H, int W, int C,
                      int dimz, int dimy, int dimx, int64_t n_rows, void *stream_)
{
    (void)C; (void)dimz; (void)dimy; (void)dimx; (void)n_rows;
    if (!workspace || !dst) return fail(VP_EINVAL, "null pointer argument");
    WsState *rec = ws_state(workspace, false);
    if (!rec || !rec->has_hit) return fail(VP_EINVAL, "no vp_project_features call has used this workspace");
    const size_t off = rec->hit_off;
    if (rec->pipe.ok) {
        VP_HIP(hipStreamSynchronize(rec->pipe.side));
    }
    VP_HIP(hipMemcpyAsync(dst, (const char *)workspace + off, size_t(B) * V * H * W * sizeof(int),
                          hipMemcpyDeviceToDevice, (hipStream_t)stream_));
    return VP_OK;
}

size_t vp_colors_workspace_bytes(int64_t n_rows)
{
    if (n_rows <= 0) return 0;
    // status words | cell of every ID | the voxel list {ID, cell} in curve order | one count per walking wavefront
    return 256 + align256(size_t(n_rows) * sizeof(int)) + align256(size_t(n_rows) * sizeof(int2)) + align256(size_t(COLOR_WALKERS) * sizeof(int));
}

int vp_project_colors(const int32_t *occ, int dimz, int dimy, int dimx, const float *c2w, const float *intr,
                      int V, const float *grid_origin_host, double voxel_size, const uint8_t *images,
                      int img_h, int img_w, float *color_sum, int32_t *hit_count, int32_t *first_view,
                      int32_t *pixel_uv, int64_t n_rows, int view_base, void *workspace, size_t workspace_bytes,
                      void *stream_)
{
    if (!occ || !c2w || !intr || !grid_origin_host || !images || !color_sum || !hit_count || !workspace)
        return fail(VP_EINVAL, "null pointer argument");
    if (dimz <= 0 || dimy <= 0 || dimx <= 0 || V <= 0 || img_h <= 0 || img_w <= 0 || n_rows <= 0)
        return fail(VP_EINVAL, "non-positive dimension");
    const long long cells = (long long)dimz * dimy * dimx;
    if (cells >= (1ll << 31) || n_rows >= (1ll << 31)) return fail(VP_EINVAL, "occupancy grid or row count >= 2^31");
    if (workspace_bytes < vp_colors_workspace_bytes(n_rows))
        return fail(VP_EWORKSPACE, "workspace has %zu bytes, need %zu", workspace_bytes, vp_colors_workspace_bytes(n_rows));
    if ((uintptr_t)workspace & 255) return fail(VP_EWORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t stream = (hipStream_t)stream_;
    int *status = (int *)workspace;
    int *cell_of_id = (int *)((char *)workspace + 256);
    int2 *list = (int2 *)((char *)cell_of_id + align256(size_t(n_rows) * sizeof(int)));
    int *counts = (int *)((char *)list + align256(size_t(n_rows) * sizeof(int2)));
    VP_HIP(hipMemsetAsync(status, 0, 256, stream));
    VP_HIP(hipMemsetAsync(cell_of_id, 0xFF, size_t(n_rows) * sizeof(int), stream));
    // the Morton curve over the grid's 4x4x4 blocks, shared by wavefronts that take 64 or more positions each
    ColorCurve curve;
    curve.nbx = (dimx + 3) / 4; curve.nby = (dimy + 3) / 4; curve.nbz = (dimz + 3) / 4;
    const auto bits = [](int n) { int b = 0; while ((1 << b) < n) b++; return b; };
    curve.bx = bits(curve.nbx); curve.by = bits(curve.nby); curve.bz = bits(curve.nbz);
    curve.slots = 1ll << (curve.bx + curve.by + curve.bz);       // < 2^32: < 8 x the blocks of a grid of < 2^31 cells
    const long long want = (curve.slots + 63) / 64;
    const int walkers = (int)(want > COLOR_WALKERS ? COLOR_WALKERS : (want + 3) / 4 * 4);
    hipLaunchKernelGGL(k_color_cells<1>, dim3(walkers / 4), dim3(256), 0, stream, (const int *)occ, dimz, dimy, dimx, curve,
                       cell_of_id, (long long)n_rows, status, counts, list);
    hipLaunchKernelGGL(k_color_scan, dim3(1), dim3(1024), 0, stream, counts, walkers, status);
    hipLaunchKernelGGL(k_color_cells<2>, dim3(walkers / 4), dim3(256), 0, stream, (const int *)occ, dimz, dimy, dimx, curve,
                       cell_of_id, (long long)n_rows, status, counts, list);
    int st[2];
    VP_HIP(hipMemcpyAsync(st, status, sizeof(st), hipMemcpyDeviceToHost, stream));
    VP_HIP(hipStreamSynchronize(stream));
    if (st[CST_BADID]) return fail(VP_EBADID, "an occupancy ID is outside [1, n_rows): outputs are too small for the grid's IDs");
    if (st[CST_DUP]) return fail(VP_EINVAL, "an occupancy ID labels more than one cell: the colour path needs unique IDs "
                                            "(build_sparse_occupancy.py:44-46 produces them)");
    const bool tiny = (long long)V * img_h * img_w * 3 < 4;
#define VP_LAUNCH_COLORS(UV, TINY)                                                                                           \
    hipLaunchKernelGGL((k_project_colors<UV, TINY>), dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, stream,           \
                       (const int *)cell_of_id, (const int2 *)list, (const int *)status, dimy, dimx, c2w, intr, V,           \
                       grid_origin_host[0], grid_origin_host[1], grid_origin_host[2], voxel_size,                            \
                       (const unsigned char *)images, img_h, img_w, color_sum, (int *)hit_count, (int *)first_view,          \
                       (int *)pixel_uv, (long long)n_rows, view_base)
    if (pixel_uv) { if (tiny) VP_LAUNCH_COLORS(true, true); else VP_LAUNCH_COLORS(true, false); }
    else          { if (tiny) VP_LAUNCH_COLORS(false, true); else VP_LAUNCH_COLORS(false, false); }
#undef VP_LAUNCH_COLORS
    VP_HIP(hipGetLastError());
    VP_HIP(hipStreamSynchronize(stream));
    return VP_OK;
}

size_t vp_upsample_workspace_bytes(int C, int h, int w, int src_is_f16)
{
    if (C <= 0 || h <= 0 || w <= 0) return 0;
    return align256(size_t(C) * h * w * (src_is_f16 ? 2 : 4));
}

int vp_upsample_features(const void *src_chw, int src_is_f16, int C, int h, int w, void *dst_hwc, int dst_is_f16,
                         int H, int W, void *workspace, size_t workspace_bytes, void *stream_)
{
    if (!src_chw || !dst_hwc || !workspace) return fail(VP_EINVAL, "null pointer argument");
    if (C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return fail(VP_EINVAL, "non-positive dimension");
    if (dst_is_f16 && !src_is_f16) return fail(VP_EINVAL, "a float16 destination needs a float16 source (PTD:126 casts back to the file's dtype)");
    if ((long long)H * W >= (1ll << 31) || (long long)h * w >= (1ll << 31)) return fail(VP_EINVAL, "image has >= 2^31 pixels");
    if (workspace_bytes < vp_upsample_workspace_bytes(C, h, w, src_is_f16))
        return fail(VP_EWORKSPACE, "workspace has %zu bytes, need %zu", workspace_bytes, vp_upsample_workspace_bytes(C, h, w, src_is_f16));
    hipStream_t stream = (hipStream_t)stream_;
    const long long P = (long long)h * w;
    const dim3 tgrid((unsigned)((P + 63) / 64), (unsigned)((C + 63) / 64));
    // cv::resize derives the scale from the destination size: inv_scale = dsize / ssize, scale = 1 / inv_scale
    const double scale_x = 1.0 / ((double)W / (double)w), scale_y = 1.0 / ((double)H / (double)h);
    const unsigned ublocks = (unsigned)(((long long)H * W + 4 * UPS_PIX - 1) / (4 * UPS_PIX));
#define VP_UPS1(TS_, TD_, VEC_, NV_) hipLaunchKernelGGL((k_upsample_hwc<TS_, TD_, VEC_, NV_>), dim3(ublocks), dim3(256), 0, stream, \
        (const TS_ *)workspace, (TD_ *)dst_hwc, C, h, w, H, W, scale_x, scale_y)
    // channel groups a lane owns: the register window of the kernel is instantiated for 1, 2 or 4 of them
#define VP_UPS(TS_, TD_, VEC_)                                           \
    do {                                                                 \
        const int groups_ = (C + 64 * (VEC_) - 1) / (64 * (VEC_));      \
        if (groups_ <= 1) VP_UPS1(TS_, TD_, VEC_, 1);                    \
        else if (groups_ <= 2) VP_UPS1(TS_, TD_, VEC_, 2);               \
        else if (groups_ <= 4) VP_UPS1(TS_, TD_, VEC_, 4);               \
        else VP_UPS1(TS_, TD_, VEC_, 0);                                 \
    } while (0)
    const bool al16 = (((uintptr_t)workspace | (uintptr_t)dst_hwc) & 15) == 0;
    if (src_is_f16) {
        if ((((uintptr_t)src_chw | (uintptr_t)workspace) & 15) == 0 && P % 8 == 0 && C % 8 == 0)
            hipLaunchKernelGGL((k_chw_to_hwc_v16<_Float16>), tgrid, dim3(256), 0, stream, (const _Float16 *)src_chw, (_Float16 *)workspace, C, P);
        else
            hipLaunchKernelGGL((k_chw_to_hwc<_Float16>), tgrid, dim3(256), 0, stream, (const _Float16 *)src_chw, (_Float16 *)workspace, C, P);
        const bool v8 = al16 && C % 8 == 0;
        if (dst_is_f16) { if (v8) VP_UPS(_Float16, _Float16, 8); else VP_UPS(_Float16, _Float16, 1); }
        else { if (v8) VP_UPS(_Float16, float, 8); else VP_UPS(_Float16, float, 1); }
    } else {
        if ((((uintptr_t)src_chw | (uintptr_t)workspace) & 15) == 0 && P % 4 == 0 && C % 4 == 0)
            hipLaunchKernelGGL((k_chw_to_hwc_v16<float>), tgrid, dim3(256), 0, stream, (const float *)src_chw, (float *)workspace, C, P);
        else
            hipLaunchKernelGGL((k_chw_to_hwc<float>), tgrid, dim3(256), 0, stream, (const float *)src_chw, (float *)workspace, C, P);
        if (al16 && C % 4 == 0) VP_UPS(float, float, 4); else VP_UPS(float, float, 1);
    }
#undef VP_UPS1
#undef VP_UPS
    VP_HIP(hipGetLastError());
    return VP_OK;
}

int vp_voxel_coords(const float *points_xyz, int64_t N, const float *grid_origin_host, float voxel_size,
                    int32_t *coords, int32_t *scratch8_dev, int32_t *minmax_host, void *stream_)
{
    if (!points_xyz || !grid_origin_host || !coords || !scratch8_dev || !minmax_host) return fail(VP_EINVAL, "null pointer argument");
    if (N <= 0 || N >= (1ll << 31) - 1) return fail(VP_EINVAL, "point count must be in [1, 2^31 - 2]");
    hipStream_t stream = (hipStream_t)stream_;
    const int init[8] = {2147483647, 2147483647, 2147483647, -2147483647 - 1, -2147483647 - 1, -2147483647 - 1, 0, 0};
    VP_HIP(hipMemcpyAsync(scratch8_dev, init, sizeof(init), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_voxel_coords, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, points_xyz, (long long)N,
                       grid_origin_host[0], grid_origin_host[1], grid_origin_host[2], voxel_size, (int *)coords,
                       (int *)scratch8_dev, (int *)scratch8_dev + 6);
    VP_HIP(hipGetLastError());
    int back[8];
    VP_HIP(hipMemcpyAsync(back, scratch8_dev, sizeof(back), hipMemcpyDeviceToHost, stream));
    VP_HIP(hipStreamSynchronize(stream));
    if (back[6]) return fail(VP_EINVAL, "a point's voxel coordinate is not finite or beyond 2^30 cells from the grid origin");
    memcpy(minmax_host, back, 6 * sizeof(int));
    return VP_OK;
}

int vp_scatter_occupancy(const int32_t *coords, int64_t N, const int32_t *shift3_host, int dimz, int dimy, int dimx,
                         int32_t *occ, int32_t *scratch8_dev, void *stream_)
{
    if (!coords || !shift3_host || !occ || !scratch8_dev) return fail(VP_EINVAL, "null pointer argument");
    if (N <= 0 || N >= (1ll << 31) - 1 || dimz <= 0 || dimy <= 0 || dimx <= 0) return fail(VP_EINVAL, "bad size");
    if ((long long)dimz * dimy * dimx >= (1ll << 31)) return fail(VP_EINVAL, "occupancy grid has >= 2^31 cells");
    hipStream_t stream = (hipStream_t)stream_;
    VP_HIP(hipMemsetAsync(occ, 0, size_t(dimz) * dimy * dimx * sizeof(int), stream));
    VP_HIP(hipMemsetAsync(scratch8_dev, 0, 8 * sizeof(int), stream));
    hipLaunchKernelGGL(k_scatter_occupancy, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, stream, (const int *)coords, (long long)N,
                       shift3_host[0], shift3_host[1], shift3_host[2], dimz, dimy, dimx, (int *)occ, (int *)scratch8_dev);
    VP_HIP(hipGetLastError());
    int bad = 0;
    VP_HIP(hipMemcpyAsync(&bad, scratch8_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    VP_HIP(hipStreamSynchronize(stream));
    if (bad) return fail(VP_EINVAL, "a shifted voxel coordinate falls outside the [dimz,dimy,dimx] grid");
    return VP_OK;
}

int vp_aggregate_view_f16(float *view_sum, int32_t *view_count, void *run16, int32_t *views, int32_t *first_view,
                          int view_index, int32_t *nonfinite_dev, int64_t n_rows, int C, void *stream_)
{
    if (!view_sum || !view_count || !run16 || !views || !first_view || !nonfinite_dev) return fail(VP_EINVAL, "null pointer argument");
    if (n_rows <= 0 || C <= 0) return fail(VP_EINVAL, "non-positive dimension");
    if (((uintptr_t)view_sum & 15) || ((uintptr_t)run16 & 7)) return fail(VP_EINVAL, "view_sum must be 16-byte, run16 8-byte aligned");
    if (n_rows == 1) return VP_OK;
    hipLaunchKernelGGL(k_aggregate_view_f16, dim3((unsigned)((n_rows - 1 + 3) / 4)), dim3(256), 0, (hipStream_t)stream_, view_sum,
                       (int *)view_count, (_Float16 *)run16, (int *)views, (int *)first_view, view_index, (int *)nonfinite_dev,
                       (long long)n_rows, C);
    VP_HIP(hipGetLastError());
    return VP_OK;
}

long long vp_workspace_table_builds(const void *workspace)
{
    WsState *rec = ws_state(workspace, false);
    return rec ? rec->builds : 0;
}

int vp_workspace_create(void *workspace, size_t workspace_bytes)
{
    if (!workspace) return fail(VP_EINVAL, "null workspace");
    if ((uintptr_t)workspace & 255) return fail(VP_EWORKSPACE, "workspace must be 256-byte aligned");
    if (workspace_bytes < 2 * align256(ST_WORDS * sizeof(int))) return fail(VP_EWORKSPACE, "workspace has %zu bytes, need at least %zu", workspace_bytes, 2 * align256(ST_WORDS * sizeof(int)));
    ws_forget(workspace);            // whatever this address was before
    (void)ws_state(workspace, true); // a new record, a new generation
    return VP_OK;
}

int vp_workspace_set_option(void *workspace, int option, long long value)
{
    if (!workspace) return fail(VP_EINVAL, "null workspace");
    WsState *rec = ws_state(workspace, true);
    switch (option) {
    case VP_OPT_HEAVY_THRESHOLD: rec->opt_heavy_t = value > 0 ? value : -1; return VP_OK;
    case VP_OPT_MARCH_LDS_KB:
        // the reservation is dynamic LDS of k_first_hit, whose limit without a function attribute is 64 KiB: a larger value
        // would fail every launch on this workspace with a generic HIP error, surfacing (pipelined) only calls later
        if (value > 64) return fail(VP_EINVAL, "VP_OPT_MARCH_LDS_KB = %lld: the march's dynamic-LDS reservation is limited to 64 KiB (0 .. 64; < 0 = default)", value);
        rec->opt_march_lds_kb = value >= 0 ? value : -1; return VP_OK;
    case VP_OPT_ROW_BEGIN:       rec->opt_row_begin = value >= 0 ? value : -1; return VP_OK;
    case VP_OPT_ROW_END:         rec->opt_row_end = value >= 0 ? value : -1; return VP_OK;
    case VP_OPT_ONE_VIEW_GATHER: rec->opt_one_view = value >= 0 ? value : -1; return VP_OK;
    case VP_OPT_PART_PIXELS:     rec->opt_part_px = value > 0 ? value : -1; return VP_OK;
    case VP_OPT_ONE_VIEW_SPLIT:  rec->opt_one_view_split = value >= 0 ? value : -1; return VP_OK;
    default: return fail(VP_EINVAL, "unknown workspace option %d", option);
    }
}

int vp_workspace_release(void *workspace)
{
    ws_forget(workspace);
    return VP_OK;
}

}  // extern "C"
