"""View sharding across the GPUs of one node (no counterpart in the reference, which is
single-process / single-GPU).

Every (pixel, view) ray is independent given the replicated, read-only occupancy grid, and per-voxel results
combine by ``+`` (project_image_cuda_kernel.cu:77,88), so views are the natural unit: rank r of G owns views
r, r+G, r+2G, ... together with their feature maps (never moved between GPUs), and ONE sum-reduction of
{feature-sum f32 [N+1,C], pixel-count i32 [N+1], view-count i32 [N+1]} finishes the scene.  Counts are
reduced as integers (bit-exact); fp32 sums differ from the single-GPU order only in rounding (<= 1e-6 rel).
"""


def views_of_rank(n_views, rank, world):
    """Indices of the views rank ``rank`` projects."""
    return list(range(rank, n_views, world))


def reduce_partials(dist, tensors):
    """In-place SUM all-reduce of each tensor (RCCL on GPUs, gloo in the CPU tests)."""
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tensors
