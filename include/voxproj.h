/*
 * voxproj.h -- C-ABI of the MI355X-native 2D -> sparse-voxel feature projector.
 *
 * Plain pointers and sizes only (no torch types).  Every entry point names the reference
 * interface it replaces; paths are relative to
 * /root/reference/cuda_project_image_to_sparse_voxel/ :
 *
 *   K.cu  = project_image_cuda_kernel.cu      W.cpp = project_image_cuda.cpp
 *   DPF   = debug_project_features.py         DPC   = debug_project_colors.py
 *   BSO   = build_sparse_occupancy.py         AGG   = aggregate_voxel_features_onthefly.py
 *
 * All device pointers must belong to the HIP device that is current on the calling thread.
 * Functions return VP_OK (0) or a negative VP_E* code; vp_last_error() gives the message for
 * the calling thread.  Nothing here falls back to a CPU path.
 */
#ifndef VOXPROJ_H
#define VOXPROJ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VP_ABI_VERSION 4   /* 4: the workspace carries part slots for split voxels (vp_workspace_bytes grew), vp_profile_read slot [3]
                              is k_combine_parts, counters [7] / [24] / [25], VP_OPT_PART_PIXELS, VP_OPT_ONE_VIEW_SPLIT; one-view calls
                              split large voxels too.  A caller built against 3 must re-query vp_workspace_bytes */

enum {
    VP_OK = 0,
    VP_EINVAL = -1,      /* bad argument (shape, null pointer, size)                             */
    VP_EWORKSPACE = -2,  /* workspace too small / misaligned                                     */
    VP_EHIP = -3,        /* a HIP runtime call failed                                            */
    VP_EBADID = -4,      /* an occupancy ID hit by a ray is outside [1, n_rows)  (K.cu:71,77:     */
                         /* the reference would write out of bounds)                 */
    VP_EUNSUPPORTED = -5 /* pred_mode = true (K.cu:444-450 is unreachable in the reference too)  */
};

/* flags for vp_project_features */
enum {
    VP_FLAG_SYNC = 1,        /* block until the device work is done and report device-side errors
                                (the reference always does: K.cu:454-457).  EVERY error pending on the
                                workspace is drained by this report: the highest-ranking one is returned, the
                                others are named in vp_last_error, none is left to fail the next call       */
    VP_FLAG_REUSE_ACCEL = 2, /* the occupancy-derived tables in the workspace are still valid for
                                this occupancy grid (same pointer, contents and n_rows): skip
                                rebuilding them.  VP_EINVAL if the library has not built tables on
                                this workspace for the same (B, dims, n_rows)                       */
    VP_FLAG_EXACT_MARCH = 4, /* A/B arm: evaluate every ray sample like K.cu:47-82 does instead of
                                leaping over provably empty space (same results, slower)           */
    VP_FLAG_PIPELINE = 8,    /* asynchronous job mode (excludes VP_FLAG_SYNC): phase 1 (ray-march) runs on a
                                library-owned side stream -- held to a few wavefronts per CU -- so that the
                                march of this call overlaps the gather of the previous call on the same
                                workspace (two buffer sets alternate).  The
                                gather and every write to count/out/views_hit stay on `stream`, in order.
                                The caller promises that occ, vmi and intr are not being written by work
                                still pending on `stream`, keeps them alive and unchanged until the stream
                                has been synchronised (vp_workspace_status does), and uses one stream per
                                workspace.                                                            */
    VP_FLAG_SERIAL_SUMS = 32,  /* sum EVERY voxel with one wavefront in (b, v, y, x) order, however many pixels it got in
                                the call: no voxel is split into parts (one view: shared by a workgroup), so all sums are
                                bit-identical to the serial order of oracle/projector_oracle.c (split voxels are within 1e-4,
                                not bit-equal).
                                For callers that round the sums afterwards and promise the reference's bits -- the
                                aggregator's parity mode (DPF:252 rounds to float16).  Slower only when a voxel is large. */
    VP_FLAG_GATHER_ONLY = 64,  /* phase 2 only: no ray-march; the first-hit images, the per-call histogram and the view table of
                                  the PREVIOUS call on this workspace are used again (same arguments, checked) for the row range
                                  now set with VP_OPT_ROW_BEGIN / VP_OPT_ROW_END.  Two calls -- rows [0, h), then rows [h, n_rows)
                                  with this flag -- leave exactly what one call leaves, and the rows below h are final while
                                  the second gather still runs: a multi-GPU job starts their all-reduce under it.  Runs on the
                                  caller's stream; VP_EINVAL when no SUCCESSFUL call precedes it on the workspace, when that
                                  call had no row range (it gathered every row already), or when its feature maps, poses,
                                  outputs or shapes differ from this call's */
    VP_FLAG_VERIFY_ACCEL = 16  /* blocking calls only (ignored with VP_FLAG_PIPELINE or VP_FLAG_REUSE_ACCEL): the
                                workspace has not been written by anyone else since the previous call on it;
                                compare the occupancy grid with the 32-bit copy kept from the call that built
                                the tables (one pass over the grid + a 4-byte read-back) and rebuild them only
                                if a cell, the shape or n_rows changed.  For callers that pass a NEW tensor with
                                the SAME contents on every call, as debug_project_features.py:143 does.       */
};

int vp_abi_version(void);
const char *vp_last_error(void);

/*
 * Bytes of device scratch memory vp_project_features needs for a call of this shape
 * (first-hit ID image, per-call hit histogram, ID -> cell table, occupancy block masks and block
 * distance field, view table, part slots of C floats for the split voxels' partial rows).
 */
size_t vp_workspace_bytes(int B, int V, int H, int W, int C,
                          int dimz, int dimy, int dimx, int64_t n_rows);

/*
 * Replaces project_features_cuda_forward_impl (K.cu:374-459), i.e. what the extension function
 * project_features_cuda.project_features_cuda(...) (W.cpp:23-79) does after validation, for
 * pred_mode = false.
 *
 *   feats        f32 [B,V,H,W,C] channels-last, device            (K.cu:375)
 *   occ          i64 [B,dimz,dimy,dimx], 0 = empty, else voxel ID  (K.cu:376)
 *   vmi          f32 [B*V*16] row-major camera->world matrices     (K.cu:377, :178-179)
 *   intr         f32 [B,4] = fx, fy, mx, my per batch              (K.cu:378, cudaUtil.h:86-93)
 *   opts_host    f32 [5] HOST = width, height, depthMin, depthMax, rayIncrement (K.cu:400-407)
 *   count        i32 [n_rows]   in/out, count[id] += #pixels        (K.cu:77)
 *   out          f32 [n_rows,C] in/out, out[id,:] += feature rows   (K.cu:85-91)
 *   views_hit    i32 [n_rows]   in/out or NULL, += number of views of this call in which the voxel
 *                received at least one pixel -- the aggregator's "hit_count" (AGG:313 counts VIEWS,
 *                one per debug_project_features run), so that multi-view calls can feed it
 *   grid_origin_host f32 [3] HOST                                  (K.cu:412-413)
 *   voxel_size                                                     (K.cu:414)
 *   workspace    device scratch of >= vp_workspace_bytes(...) bytes, 256-byte aligned
 *   stream       hipStream_t (NULL = default stream)
 *
 * Results: first-hit voxel assignment and counts bit-exact with the arithmetic contract in
 * oracle/projector_oracle.c; feature sums accumulated in fp32 in (b, v, y, x) order per voxel.
 * width/height in opts must equal W/H (the reference indexes features with the opts values,
 * K.cu:74-76; a mismatch reads garbage there and is rejected here).
 */
int vp_project_features(const float *feats, const int64_t *occ, const float *vmi,
                        const float *intr, const float *opts_host,
                        int32_t *count, float *out, int32_t *views_hit,
                        const float *grid_origin_host, float voxel_size,
                        int B, int V, int H, int W, int C,
                        int dimz, int dimy, int dimx, int64_t n_rows,
                        void *workspace, size_t workspace_bytes,
                        void *stream, int flags);

/*
 * Stage-5 front end: index of the nearest voxel for each of M Gaussian centres -- replaces
 * map_gaussians_to_voxels (voxel_to_gaussian/voxeltoGaussian_logits.py:87-105; identical code at
 * voxel_to_gaussian/voxeltoGaussian.py:84-93: sklearn KDTree(leaf_size=16).query(k=1) in float64).
 *   pts_sorted  f32 [N,3] voxel positions sorted by grid cell (device)
 *   perm        i32 [N]   original index of each sorted position
 *   cell_start  i32 [nx*ny*nz + 1] first sorted position of each cell ((z*ny + y)*nx + x order)
 *   grid_origin3 f64 [3] HOST, cell_size, nx, ny, nz: the bucketing grid
 *   queries     f32 [M,3] device;  out i64 [M] device: nearest original index (lowest index on exact ties)
 * Asynchronous on `stream`.
 */
int vp_nearest_voxel(const float *pts_sorted, const int32_t *perm, const int32_t *cell_start,
                     const double *grid_origin3, double cell_size, int nx, int ny, int nz,
                     const float *queries, int64_t M, int64_t *out, void *stream);

/*
 * Measurement aid: streams n_floats (a multiple of 4, 16-byte aligned) from `src` with non-temporal 16-byte
 * loads and discards them -- bench.py times it to quote the gather against the box's own streaming-read
 * ceiling as well as the nominal HBM peak.  No reference counterpart.
 */
int vp_stream_read(const float *src, int64_t n_floats, float *sink, void *stream);

/*
 * Same as vp_project_features with the feature maps stored as IEEE binary16 [B,V,H,W,C] (C % 8 == 0).
 * LSeg features are fp16 at rest (script/extract_lseg_features.py:97) and
 * prepare_tensor_data.py:126 casts the resized maps back to fp16 before widening them, so every value the
 * reference kernel reads is fp16-representable; this entry point reads half the bytes, widens exactly and
 * accumulates in fp32 in the same order -- outputs are bit-identical to the fp32 path on the same data.
 * (No reference counterpart: its wrapper insists on float32, W.cpp:46.)
 */
int vp_project_features_f16(const void *feats_f16, const int64_t *occ, const float *vmi,
                            const float *intr, const float *opts_host,
                            int32_t *count, float *out, int32_t *views_hit,
                            const float *grid_origin_host, float voxel_size,
                            int B, int V, int H, int W, int C,
                            int dimz, int dimy, int dimx, int64_t n_rows,
                            void *workspace, size_t workspace_bytes,
                            void *stream, int flags);

/*
 * Drains the workspace's streams (the library's side stream, then `stream`), then reads the sticky device-side error words of
 * the workspace -- a page of pinned host memory owned by the library's record of the workspace, which the kernels write
 * through its device mapping: no device-to-host copy, and nothing a stranger could have overwritten.  They collect the
 * errors (tables gone from recycled memory: VP_EINVAL; a ray parameter that cannot advance: VP_EINVAL; out-of-range ID:
 * VP_EBADID) of EVERY call made on the workspace since they were last reported, pipelined or not -- no later call erases
 * them.  One condition is reported per call, in that order, and only the reported one is cleared: call again (until
 * VP_OK) to see the others.
 * The reference only prints device errors (K.cu:454-457, cutilCheckMsg); here they surface as a return code.
 * Returns VP_OK when no call has reported anything.
 */
int vp_workspace_status(void *workspace, void *stream);

/*
 * Diagnostic counters of the last call on this workspace, copied to host_words[0..n) after a
 * stream synchronise: [0] = rays that hit an out-of-range ID, [1] = voxels whose search box
 * missed pixels and were rescanned over whole images (performance hint only; results are exact
 * either way), [2] = voxels that collected more pixels than the heavy threshold in this call (summed in
 * parts), [7] = the heavy threshold in force, [8] = pixels of a one-view call whose ray hit a voxel (when the device sizes
 * the parts from it), [9] / [10] = pixels above which a voxel was cut into parts / pixels per part in force, [24] = the parts
 * planned in this call, [25] = the voxels they belong to.  No reference counterpart.
 */
int vp_workspace_counters(void *workspace, int32_t *host_words, int n, void *stream);

/*
 * Per-kernel device timing with HIP events recorded on the stream the kernels are launched on
 * (measurement harness; no reference counterpart -- the reference has no timers).
 * vp_profile_enable(1) starts recording for subsequent vp_project_features calls of this process;
 * vp_profile_read synchronises the recorded events and returns, per kernel group, the summed
 * milliseconds and the number of launches: [0] = table preparation (memsets, occupancy tables, view
 * table), [1] = k_first_hit + work list + view table (phase 1), [2] = k_gather / k_gather_one (phase 2, the parts of the split
 * voxels included), [3] = k_combine_parts (the split voxels' partial rows added to their rows);
 * then clears the record.
 */
int vp_profile_enable(int on);
int vp_profile_read(double *ms4, int64_t *launches4);

/*
 * RGB path: replaces the per-voxel Python loop of DPC:54-81 plus the per-view accumulation of
 * aggregate_voxel_colors_onthefly.py:134-140 for a batch of V views (voxel-driven, nearest pixel, no
 * occlusion test, float64 arithmetic exactly as numpy promotes it there).  One lane per occupied voxel; the
 * lanes take the voxels along the Morton curve of the grid's cells (a list the call builds on the device), which
 * changes no output bit and halves the image lines a launch fetches.
 *
 *   occ        i32 [dimz,dimy,dimx] device, > 0 = voxel ID (BSO:44-46; DPC:50 tests occ > 0).  Every ID must label
 *              exactly ONE cell (build_sparse_occupancy.py guarantees it); a duplicate returns VP_EINVAL
 *   c2w        f32 [V,16] device, row-major camera->world (DPC:61-62 reads R and t from it)
 *   intr       f32 [V,4] device, fx fy cx cy of each view (DPC:64)
 *   images     u8  [V,img_h,img_w,3] device (DPC:52,70)
 *   color_sum  f32 [n_rows,3] in/out: += img[v,u]/255 for every view that sees the voxel, in view order (AGGC:139)
 *   hit_count  i32 [n_rows]   in/out: += number of such views (AGGC:140)
 *   first_view i32 [n_rows]   in/out or NULL: min(view_base + v) over those views -- reproduces the
 *                             dict insertion order of AGGC:136-137 on the host
 *   pixel_uv   i32 [V,n_rows,2] out or NULL: the pixel (u, v) sampled for voxel `id` in view v (DPC:76
 *                             `pixel_indices`), (-1,-1) where the voxel is not seen
 *   workspace  device scratch of >= vp_colors_workspace_bytes(n_rows) bytes (12 per row + 33 KiB), 256-byte aligned
 * Synchronous (returns after the stream has drained).
 */
size_t vp_colors_workspace_bytes(int64_t n_rows);
int vp_project_colors(const int32_t *occ, int dimz, int dimy, int dimx,
                      const float *c2w, const float *intr, int V,
                      const float *grid_origin_host, double voxel_size,
                      const uint8_t *images, int img_h, int img_w,
                      float *color_sum, int32_t *hit_count, int32_t *first_view, int32_t *pixel_uv,
                      int64_t n_rows, int view_base, void *workspace, size_t workspace_bytes, void *stream);

/*
 * Feature-map up-sampler: replaces the 512 x cv2.resize(channel, (W,H), INTER_LINEAR) calls, the cast back to the file's
 * dtype and the permute to channels-last of prepare_tensor_data.py:119-127,152,183-185.
 *   src_chw   f16 or f32 [C,h,w] device (the LSeg .npy layout, script/extract_lseg_features.py:97)
 *   dst_hwc   f32 [H,W,C] (what project_features_cuda reads) or f16 [H,W,C] (for vp_project_features_f16; f16 source only)
 *   workspace >= vp_upsample_workspace_bytes(C,h,w,src_is_f16) bytes: the [h,w,C] transpose of the source
 * Arithmetic = OpenCV's published INTER_LINEAR rule for CV_32F images (half-pixel centres, float32 coefficients,
 * horizontal then vertical pass, no FMA), cast to the source dtype, widened: spelled out in csrc/vp_prep.h and restated in
 * oracle/resize_oracle.py.  Also produces a plain copy when (H,W) == (h,w).  Asynchronous on `stream`.
 */
size_t vp_upsample_workspace_bytes(int C, int h, int w, int src_is_f16);
int vp_upsample_features(const void *src_chw, int src_is_f16, int C, int h, int w,
                         void *dst_hwc, int dst_is_f16, int H, int W,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * Occupancy builder: replaces build_sparse_occupancy.py:30-53 in two steps around the one host decision (grid size):
 *   vp_voxel_coords      coords[i] = np.round((pts[i] - origin) / voxel_size) in float32, half to even (BSO:32), written as
 *                        i32 [N,3] (x,y,z); minmax_host[0..2] = per-axis minimum, [3..5] = maximum (BSO:35,40).  Blocking.
 *   vp_scatter_occupancy occ[z,y,x] = i + 1 for coords[i] - shift (BSO:36-39 shifts by the minimum when any is negative),
 *                        the LAST vertex wins on duplicates (BSO:45-46); occ i32 [dimz,dimy,dimx] is zeroed first.  Blocking.
 *   scratch8_dev         8 ints of device scratch
 */
int vp_voxel_coords(const float *points_xyz, int64_t N, const float *grid_origin_host, float voxel_size,
                    int32_t *coords, int32_t *scratch8_dev, int32_t *minmax_host, void *stream);
int vp_scatter_occupancy(const int32_t *coords, int64_t N, const int32_t *shift3_host,
                         int dimz, int dimy, int dimx, int32_t *occ, int32_t *scratch8_dev, void *stream);

/*
 * The aggregator's per-view accumulate in the reference's arithmetic (aggregate_voxel_features_onthefly.py:307-313 on
 * the float16 rows of debug_project_features.py:252), over the rows hit in this view only:
 *   view_sum   f32 [n_rows,C] in: the view's pixel sums (vp_project_features into a zeroed buffer); out: zero again
 *   view_count i32 [n_rows]   in: the view's pixel counts; out: zero again
 *   run16      f16 [n_rows,C] running per-voxel sum: first hit = clone of the fp16-rounded row, later hits fp16 +=
 *   views      i32 [n_rows]   += 1 for every voxel hit in this view (AGG:313 counts VIEWS)
 *   first_view i32 [n_rows]   = view_index where the voxel is hit for the first time (dict insertion order)
 *   nonfinite_dev i32 [1]     |= 1 if a float16 row of this view holds NaN or Inf (AGG:303-304 prints an error)
 * Asynchronous on `stream`.
 */
int vp_aggregate_view_f16(float *view_sum, int32_t *view_count, void *run16, int32_t *views, int32_t *first_view,
                          int view_index, int32_t *nonfinite_dev, int64_t n_rows, int C, void *stream);

/*
 * Workspace lifetime.  The library keeps ONE record per workspace (side stream and events of VP_FLAG_PIPELINE, the shape
 * the occupancy tables in it were built for, its options).  The record is looked up by the workspace's address but is
 * not trusted on the address alone: its generation number is also written into the workspace memory (a header in the
 * first 256 bytes) and compared, on the device, by every call -- so memory that was freed and handed out again, or
 * overwritten, is recognised: calls that trust the tables (VP_FLAG_REUSE_ACCEL) then do no work and the next
 * vp_workspace_status returns VP_EINVAL; calls that rebuild them re-initialise the header.
 *
 *   vp_workspace_create   "this memory is a new workspace": drops any record the address had (streams, tables, options)
 *                         and starts a new generation.  Optional for memory the library has never seen (the first call
 *                         creates the record), REQUIRED manners for memory that is being reused as a workspace.
 *   vp_workspace_release  drains and destroys the record; call before freeing or recycling the memory.
 * No reference counterpart (the reference allocates nothing between calls).
 */
int vp_workspace_create(void *workspace, size_t workspace_bytes);
int vp_workspace_release(void *workspace);

/*
 * Options of a workspace, read by the calls made on it (they replace the environment variables of ABI v2; the
 * library reads no environment variable).  value < 0 (or 0 for the threshold) restores the default.
 *   VP_OPT_HEAVY_THRESHOLD  voxels that collect more than this many pixels in ONE call are not summed by a single wavefront: they
 *                           are cut into parts of VP_OPT_PART_PIXELS pixels, each part summed by a wavefront of the same gather
 *                           launch, the partial rows added to the voxel's row in a fixed order by a follow-up kernel.  Default
 *                           min(256 + 64*B*V, 2048) -- the longest job a wavefront can get bounds the tail of the launch --;
 *                           calls of one view: see VP_OPT_ONE_VIEW_SPLIT;
 *                           VP_FLAG_SERIAL_SUMS overrides it with "never".  Never below the part size
 *   VP_OPT_PART_PIXELS      pixels per part of a split voxel (default: the threshold; 256 for calls of one view); raised to
 *                           2*B*V*H*W / slots when the call is so large that its parts could outnumber the workspace's part
 *                           slots (65536, fewer for rows wider than 2 KiB; 8192 for calls of one view)
 *   VP_OPT_ONE_VIEW_SPLIT   calls of ONE view: voxels that collect more than this many pixels are cut into parts like those of
 *                           multi-view calls (one wavefront of the one-view gather per part, k_combine_parts behind it), the
 *                           others are summed by one wavefront in the oracle's order.  Default (< 0): decided on the device from
 *                           the number of pixels of the view whose ray hit a voxel -- parts of max(32, 2 * hits / 8192) pixels
 *                           (VP_OPT_PART_PIXELS fixes the size), threshold twice the part, at least 256 pixels for views of up
 *                           to 262144 pixels; VP_OPT_HEAVY_THRESHOLD, when set, is taken as this threshold.  0 = never split:
 *                           round 5's behaviour (the four wavefronts of a workgroup share a voxel above VP_OPT_HEAVY_THRESHOLD,
 *                           default 320), the A/B arm.  Never below the part size
 *   VP_OPT_MARCH_LDS_KB     dynamic-LDS reservation of the march kernel in KiB = its occupancy cap (default beside a
 *                           running gather in VP_FLAG_PIPELINE mode: 41 KiB = 3 workgroups per CU, 30 KiB = 5 when a
 *                           feature row is at most 1 KiB -- fp16 maps of 512 channels --; 0 otherwise).  Valid: 0 .. 64
 *                           (a kernel's dynamic-LDS limit); larger values are refused with VP_EINVAL
 *   VP_OPT_ROW_BEGIN / _END phase 2 of the following calls gathers only the voxel IDs in [begin, end) (default: all rows;
 *                           value < 0 restores it).  Phase 1 is not restricted: the histogram it leaves covers every row, so
 *                           that a VP_FLAG_GATHER_ONLY call can gather the other rows from it.  A voxel's parts depend on
 *                           its pixel count and boxes alone, so each ranged gather sums every voxel of its range to the
 *                           same bits as the unsplit call
 *   VP_OPT_ONE_VIEW_GATHER  0 = calls of ONE view (B*V == 1) go through the general gather kernel instead of the one-view
 *                           kernel (A/B arm; same results bit for bit); n > 0 = the one-view kernel with n workgroups per CU
 *                           (default 16; 1000 + g: exactly g workgroups, a test hook); < 0 restores the default
 */
enum { VP_OPT_HEAVY_THRESHOLD = 1, VP_OPT_MARCH_LDS_KB = 2, VP_OPT_ROW_BEGIN = 3, VP_OPT_ROW_END = 4, VP_OPT_ONE_VIEW_GATHER = 5,
       VP_OPT_PART_PIXELS = 6, VP_OPT_ONE_VIEW_SPLIT = 7 };
int vp_workspace_set_option(void *workspace, int option, long long value);

/* How many times the occupancy-derived tables of this workspace have been (re)built so far (0 if never);
 * diagnostic for VP_FLAG_REUSE_ACCEL / VP_FLAG_VERIFY_ACCEL. */
long long vp_workspace_table_builds(const void *workspace);

/*
 * Copies the first-hit ID image i32 [B,V,H,W] of the LAST vp_project_features call on this
 * workspace into dst (device pointer).  Test/diagnostic hook: the reference has no such output,
 * the parity tests use it to compare the pixel -> voxel assignment of K.cu:47-82 directly.
 */
int vp_copy_hit_image(const void *workspace, int32_t *dst, int B, int V, int H, int W,
                      int C, int dimz, int dimy, int dimx, int64_t n_rows, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VOXPROJ_H */
