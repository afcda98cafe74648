// vp_aggregate.h -- the aggregator's per-view accumulate in the reference's own (fp16) arithmetic, over the hit rows
// only.  Included by voxproj.hip only.
#pragma once

namespace {

// ------------------------------------------------------------------------------------------------
// The reference passes each view's result through two files and a Python dict loop:
//   debug_project_features.py:237-252   rows with count > 0 -> (z,y,x), per-view pixel SUMS rounded to float16
//   aggregate_voxel_features_onthefly.py:307-313
//        first time a voxel is seen:  voxel_feature_sum[k] = feat.clone()              (float16)
//        afterwards:                  voxel_feature_sum[k] += feat                     (float16 += float16: the add is
//                                                                                       done in float and rounded to half)
//        voxel_hit_count[k] += 1                                                       (counts VIEWS)
//   aggregate_voxel_features_onthefly.py:303-304  a NaN / Inf in the float16 rows is reported per view
// k_aggregate_view_f16 does exactly that for ONE view whose float32 pixel sums / pixel counts sit in view_sum /
// view_count: one wavefront per voxel ID, IDs without a pixel in this view return after one 4-byte read.  A hit row is
// read once, rounded to binary16 (round to nearest even, overflow to inf -- torch's .to(float16)), folded into the
// running float16 row, and then ZEROED together with its count: the scratch pair is all-zero again when the kernel
// ends, so the next view needs no 2 x (n_rows x C) fill.  first_view[id] records the view at which the voxel entered the
// dict (insertion order of the reference's output rows).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_aggregate_view_f16(float *__restrict__ view_sum, int *__restrict__ view_count,
                                                            _Float16 *__restrict__ run16, int *__restrict__ views,
                                                            int *__restrict__ first_view, int view_index,
                                                            int *nonfinite, long long n_rows, int C)
{
    const long long id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6) + 1;
    if (id >= n_rows) return;
    if (view_count[id] <= 0) return;                                  // DPF:237 (mapping2dto3d_num > 0)
    const int lane = threadIdx.x & 63;
    const bool first = views[id] == 0;                                // AGG:309 (voxel_feature_sum[k] is None)
    float *srow = view_sum + id * C;
    _Float16 *rrow = run16 + id * C;
    bool bad = false;
    if ((C & 3) == 0) {
        typedef _Float16 v4h_ __attribute__((ext_vector_type(4)));
        for (int c = lane * 4; c < C; c += 256) {
            const float4 s = *reinterpret_cast<const float4 *>(srow + c);
            v4h_ f = {(_Float16)s.x, (_Float16)s.y, (_Float16)s.z, (_Float16)s.w};      // DPF:252 .to(torch.float16)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float fe = (float)f[e];
                bad |= !(fabsf(fe) <= 65504.0f);                                        // AGG:303 isnan | isinf
            }
            if (!first) {
                const v4h_ r = *reinterpret_cast<const v4h_ *>(rrow + c);
#pragma unroll
                for (int e = 0; e < 4; e++) f[e] = (_Float16)((float)r[e] + (float)f[e]);   // AGG:312 fp16 +=
            }
            *reinterpret_cast<v4h_ *>(rrow + c) = f;                                    // AGG:310 clone / AGG:312
            *reinterpret_cast<float4 *>(srow + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
        for (int c = lane; c < C; c += 64) {
            _Float16 f = (_Float16)srow[c];
            bad |= !(fabsf((float)f) <= 65504.0f);
            if (!first) f = (_Float16)((float)rrow[c] + (float)f);
            rrow[c] = f;
            srow[c] = 0.f;
        }
    }
    if (__ballot(bad) != 0ull && lane == 0) atomicOr(nonfinite, 1);
    if (lane == 0) {
        if (first) first_view[id] = view_index;
        views[id] += 1;                                               // AGG:313
        view_count[id] = 0;
    }
}

}  // namespace
