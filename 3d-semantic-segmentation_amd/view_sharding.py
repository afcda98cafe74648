"""View sharding across the GPUs of one node (no counterpart in the reference, which is
single-process / single-GPU).

Every (pixel, view) ray is independent given the replicated, read-only occupancy grid, and per-voxel results
combine by ``+`` (project_image_cuda_kernel.cu:77,88), so views are the natural unit: rank r of G owns views
r, r+G, r+2G, ... together with their feature maps (never moved between GPUs), and ONE sum-reduction of
{feature-sum f32 [N+1,C], pixel-count i32 [N+1], view-count i32 [N+1]} finishes the scene.  Counts are
reduced as integers (bit-exact); fp32 sums differ from the single-GPU order only in rounding (<= 1e-6 rel).

The collective is the same order of time as a rank's projection (410 MB over xGMI against 7-8 ms of gather at G = 8), and
a scene is ONE pass -- there is no next pass to hide it under.  ``project_final_call_and_reduce`` therefore cuts the
rank's LAST projector call by voxel ID: the rows below the cut are final when the first gather is over and their
reduction runs on the collective's stream under the second gather.  One implementation, used by the entry point
(VoxelFeatureAggregator.add_final_views) and, through it, by bench.py's multi-rank step.
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


def split_point(n_rows):
    """Voxel ID at which a scene's rows are cut in two for ``project_final_call_and_reduce`` (a multiple of 64 rows near the
    middle: both halves of the big tensor stay 16-byte aligned whatever C is); 0 = too few rows to cut."""
    h = (int(n_rows) // 2 + 63) & ~63
    return h if 0 < h < int(n_rows) else 0


def project_final_call_and_reduce(dist, project, set_row_range, row_tensors, whole_tensors, n_rows, dst=None, split=True,
                                  on_projected=None):
    """A rank's LAST projector call of the scene together with the scene's collective.

    ``project(gather_only)``  queues the call on the current stream: ``False`` = the whole call (ray-march + gather of the row
                              range now set), ``True`` = phase 2 once more for the range now set, from the first-hit images
                              the previous call left (VP_FLAG_GATHER_ONLY; no second march).
    ``set_row_range(b, e)``   VP_OPT_ROW_BEGIN / _END of the workspace the call runs on; ``(None, None)`` = every row.
    ``row_tensors``           tensors indexed by voxel ID in their first dimension whose rows [0, h) are final after the first
                              gather: the feature sums [n_rows, C] (410 MB at R2 -- the one that matters).
    ``whole_tensors``         everything else that is reduced (hit counts, view counts, the number of views seen): small,
                              reduced once after the second gather.
    ``on_projected()``        optional, called when the last gather has been queued (bench.py records an event there).

    split: rows [0, h) are gathered first and their reduction is issued at once -- a collective queued on RCCL's stream
    waits for what the current stream holds at that moment, i.e. for the first gather only -- then rows [h, n_rows) are
    gathered while the first half is on the links; the second half and the small tensors follow.  Same bytes moved, about
    half of them under the gather.  Not split: one call, then one collective per tensor.  Either way every reduction has
    completed (and the row range is reset) on return; the sums equal the unsplit ones bit for bit on every rank's own part
    (every voxel is summed by the same kernel role in both forms)."""
    h = split_point(n_rows) if split else 0
    if not h:
        project(False)
        if on_projected is not None:
            on_projected()
        works = reduce_partials(dist, list(row_tensors) + list(whole_tensors), dst=dst, async_op=True)
    else:
        try:
            set_row_range(0, h)
            project(False)
            works = reduce_partials(dist, [t[:h] for t in row_tensors], dst=dst, async_op=True)
            set_row_range(h, int(n_rows))
            project(True)
        finally:
            set_row_range(None, None)
        if on_projected is not None:
            on_projected()
        works += reduce_partials(dist, [t[h:] for t in row_tensors] + list(whole_tensors), dst=dst, async_op=True)
    for w in works:
        if w is not None:
            w.wait()
    return h
