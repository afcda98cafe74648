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


def reduce_partials(dist, tensors, dst=None, async_op=False):
    """In-place SUM reduction of each tensor over all ranks (RCCL on GPUs, gloo in the CPU tests).

    ``dst=None``: all-reduce -- every rank ends up with the scene totals (what BASELINE.json's north_star words).
    ``dst=r``:    reduce to rank r only -- half the xGMI traffic of an all-reduce; the right collective when one rank
                  writes the scene's files; the other ranks' buffers are left in an unspecified state.
    Integer tensors (pixel and view counts) are reduced as integers, so they stay bit-exact.  Returns the list of work
    handles when ``async_op`` (wait on all of them before touching the tensors), else the tensors."""
    works = []
    for t in tensors:
        if dst is None:
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)
        else:
            w = dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, async_op=async_op)
        works.append(w)
    return works if async_op else tensors
