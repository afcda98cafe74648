"""Scene-level feature aggregation -- the named entry point, in one process.

Counterpart of the reference's aggregate_voxel_features_onthefly.py
(cuda_project_image_to_sparse_voxel/aggregate_voxel_features_onthefly.py:101-453).  The reference runs,
PER VIEW, three Python sub-processes glued by ~1 GB ``.pt`` files (build_sparse_occupancy /
prepare_tensor_data / debug_project_features) and then adds the per-view result into Python dicts keyed by
(z,y,x).  Here the same steps run in-process on the GPU; the output files keep the reference's names, keys,
dtypes and row order, so stage 5 (voxel_to_gaussian/voxeltoGaussian_logits.py:40-46) reads them unchanged:

  ALL_nonzero_voxel_features_{n_views}_vox{N}.pt   xyz f32 [n,3], avg_feats f16 [n,C], voxel_coords i32 [n,3]   (AGG:443-451)
  ALL_nonzero_voxels_with_features_{n_views}_vox{N}.ply   ASCII x y z + first three channels as uchar rgb        (AGG:425-440)
  checkpoint_features_{k}.pt  (every 20 views)     xyz f64, avg_feats f16, hit_count i32, voxel_coords i32      (AGG:318-352)

Two modes:

  --mode parity  (default) one projector call per view; the per-view pixel sums are rounded to fp16
                 (DPF:252), the running per-voxel sum is an fp16 tensor updated with fp16 adds (AGG:309-312),
                 "hit_count" counts VIEWS (AGG:313) and avg = sum / hit_count in fp16 (AGG:385).
                 Rows appear in dict-insertion order: by first view that hit the voxel, then by voxel ID.
                 This reproduces the reference's file bit for bit (tests/test_gpu_pipeline_rows.py checks it against
                 oracle.aggregate_views).
  --mode fast    multi-view pipelined calls (VP_FLAG_PIPELINE), fp32 sums, pixel counts and view counts kept
                 exactly; avg = fp32 sum / views (the same definition without the per-view fp16 round trips),
                 rows in voxel-ID order; additionally saves sum/count/views in ``*_fp32.pt``.  With torchrun,
                 rank r projects views r::G and one SUM collective (RCCL) per tensor combines {sum, count, views};
                 the rank's last call is cut by voxel ID so that half of the 410 MB of sums travels under the
                 gather of the other half (add_final_views; ``--no_split_collective`` = one call, then reduce).

Constants below keep the reference's names (AGG:18-29,106,209,318); every one can be overridden on the
command line, and ``--first_only`` keeps its meaning (AGG:14,112-113).
"""
import argparse
import glob
import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # RCCL over dmabuf IPC on this pool's hosts (multi-GPU --mode fast)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import build_sparse_occupancy as bso  # noqa: E402
import prepare_tensor_data as ptd  # noqa: E402
import voxproj_host  # noqa: E402
from debug_project_features import build_id_to_zyx  # noqa: E402

CHECKPOINT_DIR = "voxel_feature_checkpoints"
LSEG_DIR = "lseg_embed_features/features"
CAM_PARAMS_ORIG = "camera_params/camera_params.json"
VOXEL_PLY = "minkowski_grid.ply"
MAX_IMAGES = 216                 # AGG:106
DOWNSAMPLE_FACTOR = 0.5          # AGG:209
CHECKPOINT_EVERY = 20            # AGG:318
_NEVER = 2 ** 30


class VoxelFeatureAggregator:
    """Running per-voxel aggregate over views, resident on one GPU."""

    def __init__(self, occ_zyx, grid_origin, voxel_size, channels, mode="parity", device="cuda", parity_pipeline=True):
        self.dev = torch.device(device)
        self.mode = mode
        # VP_FLAG_PIPELINE for the one-view calls of the parity mode too: the march of view k+1 under the gather and the
        # fp16 epilogue of view k -- 0.124 vs 0.159 ms per R1 view (profiles/r03_bench_entry_parity*.json; round 2 measured
        # the opposite with a host synchronisation per call in the way)
        self.parity_pipeline = bool(parity_pipeline)
        self._intr_dev = {}                              # intrinsics already on the device, by value
        self._copy_stream = None
        self.occ3 = occ_zyx.to(self.dev).contiguous()
        self.occ = self.occ3.unsqueeze(0).long().contiguous()            # DPF:143
        self.grid_origin = [float(v) for v in grid_origin]
        self.voxel_size = float(voxel_size)
        self.C = int(channels)
        self.n_rows = int(self.occ3.max().item()) + 1                     # DPF:158-159
        self.id_to_zyx = build_id_to_zyx(self.occ3)                       # DPF:35-45
        self.valid_id = self.id_to_zyx[:, 0] != -1
        self.ws = voxproj_host.Workspace()       # device scratch is allocated by the first projector call
        self.n_seen = 0
        n, C = self.n_rows, self.C
        self.views = torch.zeros(n, dtype=torch.int32, device=self.dev)   # AGG voxel_hit_count
        if mode == "parity":
            self.run16 = torch.zeros(n, C, dtype=torch.float16, device=self.dev)   # AGG voxel_feature_sum (fp16)
            self.first_view = torch.full((n,), _NEVER, dtype=torch.int32, device=self.dev)
            # per-view scratch: zero here, and zero again after every view (the epilogue clears the rows it consumed)
            self._cnt = torch.zeros(n, dtype=torch.int32, device=self.dev)
            self._sum = torch.zeros(n, C, dtype=torch.float32, device=self.dev)
            self._nonfinite = torch.zeros(max(1, MAX_IMAGES), dtype=torch.int32, device=self.dev)   # AGG:303-304, per view
            self._reported = 0
        else:
            self.sum32 = torch.zeros(n, C, dtype=torch.float32, device=self.dev)
            # every integer the scene's collective moves lives in ONE tensor (round 5): pixel counts, view counts and, in a
            # spare slot behind them, the number of views seen -- a multi-rank pass issues the feature sums (one piece, or two
            # when the last call is cut) plus ONE integer collective instead of three
            pad = (n + 63) & ~63
            self._ints = torch.zeros(2 * pad + 64, dtype=torch.int32, device=self.dev)
            self.count = self._ints[:n]
            self.views = self._ints[pad:pad + n]
            self._n_seen_slot = self._ints[2 * pad:2 * pad + 1]

    def reset(self):
        """Forget every view seen so far (the occupancy grid, its derived tables and the workspace stay)."""
        self.flush()
        self.n_seen = 0
        if self.mode != "parity":
            self.sum32.zero_()
            self._ints.zero_()
            return
        self.views.zero_()
        self.run16.zero_()
        self.first_view.fill_(_NEVER)
        self._nonfinite.zero_()
        self._reported = 0

    def _opts(self, W, H):
        return [float(W), float(H), 0.01, 10.0, float(np.float32(self.voxel_size * 0.5))]   # DPF:167-169

    def _to_device_ready(self, t):
        """A small host tensor (poses, intrinsics) as a float32 device tensor that is READY -- for the caller's stream
        and for the library's side stream, which does not wait for the caller's (VP_FLAG_PIPELINE contract) -- without
        waiting for the projector: pinned staging, a copy stream of its own, and a host wait on that copy alone.  (A plain
        ``.to(device)`` from pageable memory synchronises torch's current stream, i.e. waits for the previous call's gather:
        ADVICE r2.)  A tensor that is already on the device is the caller's promise that it is ready."""
        if t.is_cuda:
            return t.to(self.dev, torch.float32).contiguous()
        if self._copy_stream is None:
            # high priority = a hardware queue from another pool than the projector's stream: in a process that holds many
            # streams (torch.distributed) a normal-priority stream may be handed the very queue the gather sits on, and this
            # copy -- waited for on the host -- would queue up behind it
            self._copy_stream = torch.cuda.Stream(self.dev, priority=-1)
        host = torch.empty(t.shape, dtype=torch.float32, pin_memory=True)
        host.copy_(t)
        with torch.cuda.stream(self._copy_stream):
            d = host.to(self.dev, non_blocking=True)
        self._copy_stream.synchronize()                  # the copy stream carries nothing else
        return d

    def _stage(self, feats, c2w, intr4):
        """(feats contiguous, c2w on the device and ready, intrinsics on the device, intrinsics key) of one add_views call."""
        assert feats.shape[-1] == self.C
        if self.dev.type != "cuda":
            raise RuntimeError("VoxelFeatureAggregator.add_views projects on the GPU: there is no CPU path")
        feats = feats.contiguous()
        ikey = tuple(float(v) for v in intr4.reshape(-1).tolist()) if not intr4.is_cuda else None
        intr = self._intr_dev.get(ikey) if ikey is not None else None
        if intr is None:
            intr = self._to_device_ready(intr4.reshape(1, 4))
            if ikey is not None:
                self._intr_dev[ikey] = intr              # uploaded once per distinct intrinsics
        return feats, self._to_device_ready(c2w), intr, ikey

    def _project_fast(self, feats, vmi, intr, gather_only=False):
        """One pipelined multi-view projector call of the fast mode into {sum32, count, views} (the row range set on the
        workspace, if any, applies).  What the library's side stream reads (poses, intrinsics) is kept alive until the next
        flush: torch's allocator knows nothing about that stream."""
        V, H, W, C = feats.shape
        self._keep = getattr(self, "_keep", [])
        self._keep.append((vmi, intr))
        voxproj_host.project_features_raw(feats.unsqueeze(0), self.occ, vmi, intr, self._opts(W, H), self.count,
                                          self.sum32, self.grid_origin, self.voxel_size, workspace=self.ws,
                                          sync=False, reuse_accel=None, pipeline=True, views_hit=self.views,
                                          gather_only=gather_only)

    def add_views(self, feats, c2w, intr4):
        """feats f32 [V,H,W,C] on the GPU, c2w f32 [V,4,4], intr4 f32 [4] (shared by the call's views).  Host tensors are
        staged without blocking on the projector; device tensors must be ready (not pending on a stream)."""
        V, H, W, C = feats.shape
        feats, c2w, intr, ikey = self._stage(feats, c2w, intr4)
        if self.mode == "parity":
            # One projector call per view (DPF runs once per image), then ONE hand-written epilogue over the rows that
            # view hit (vp_aggregate_view_f16): fp32 pixel sums -> fp16 (DPF:252) -> first-time clone / fp16 "+="
            # (AGG:309-312), views += 1 (AGG:313), first_view for the dict order; the epilogue leaves the scratch pair
            # zeroed, so nothing of size [n_rows, C] is filled, cast or blended per view.  Everything is queued on the
            # current stream; nothing blocks until flush().
            vmis = c2w.reshape(V, 16)
            if self.n_seen + V > self._nonfinite.numel():
                self.flush()
                grown = torch.zeros(2 * (self.n_seen + V), dtype=torch.int32, device=self.dev)
                grown[:self._nonfinite.numel()] = self._nonfinite
                self._nonfinite = grown
            # asynchronous calls through a call object that binds everything constant once -- the host side of a view is two
            # foreign calls
            key = (H, W, ikey if ikey is not None else intr.data_ptr(), torch.cuda.current_stream(self.dev).cuda_stream)
            if getattr(self, "_prep_key", None) != key:
                self.flush()
                self._prep_intr = intr
                self._prep = voxproj_host.PreparedViewCalls(self.occ, intr, self._opts(W, H), self._cnt, self._sum,
                                                            self.grid_origin, self.voxel_size, self.ws, (1, 1, H, W, C),
                                                            flags=voxproj_host.VP_FLAG_SERIAL_SUMS |   # the reference's bits
                                                            (voxproj_host.VP_FLAG_PIPELINE if self.parity_pipeline else 0))
                self._prep_key = key
            # poses and intrinsics are read by the library's side stream, which torch's allocator knows nothing about: they
            # are kept alive until the next flush -- entered AFTER the flushes above, which empty the list (entered
            # before them, the poses of this very call were dropped again and their block could be handed to the next call's
            # pose copy while the side stream still read it).  The feature maps are read on the caller's stream only, where
            # torch's stream-ordered reuse is safe, so they are not held.
            self._keep = getattr(self, "_keep", [])
            self._keep.append((vmis, intr))
            for v in range(V):
                self._prep(feats[v], vmis[v])
                voxproj_host.aggregate_view_f16(self._sum, self._cnt, self.run16, self.views, self.first_view,
                                                self.n_seen, self._nonfinite, self.n_seen, stream=self._prep.stream)
                self.n_seen += 1
            if len(self._keep) > 256:
                self.flush()
        else:
            self._project_fast(feats, c2w.reshape(-1), intr)
            self.n_seen += V
            if len(self._keep) > 64:
                self.flush()

    def add_final_views(self, feats, c2w, intr4, dst=None, split=True, on_projected=None):
        """The rank's LAST call of a scene shared by several ranks (fast mode, torch.distributed initialised) together with
        the scene's collective -- view_sharding.project_final_call_and_reduce: the call is cut by voxel ID, the feature sums of
        the rows below the cut are reduced on the collective's stream while the rows above it are still being gathered, the
        second half and the counts follow; ``split=False`` = the call as it is, then one collective per tensor (the same as
        add_views + all_reduce).  ``dst`` as in all_reduce.  On return every reduction has completed and ``n_seen`` counts the
        views of all ranks.  A rank that has no view left for this call passes ``feats=None`` and only joins the collectives."""
        import torch.distributed as dist
        from view_sharding import project_final_call_and_reduce, reduce_partials
        assert self.mode != "parity", "the parity mode is order-dependent (fp16 running sums) and stays on one GPU"
        staged = None
        if feats is not None:
            V = feats.shape[0]
            f, c, intr, _ = self._stage(feats, c2w, intr4)
            staged = (f, c.reshape(-1), intr)
            self.n_seen += V

        def project(gather_only):
            if staged is not None:
                self._project_fast(staged[0], staged[1], staged[2], gather_only=gather_only)

        self._n_seen_slot.fill_(self.n_seen)             # rides in the integer tensor's spare slot
        h = project_final_call_and_reduce(dist, project, self.ws.set_row_range, [self.sum32], [self._ints], self.n_rows,
                                          dst=dst, split=split, on_projected=on_projected)
        self.flush()
        self._finish_n_seen(dist, dst)
        return h

    def _finish_n_seen(self, dist, dst):
        """The number of views all ranks have seen: from the integer tensor's spare slot where the collective left the scene
        (all-reduce: everywhere; reduce: on rank ``dst``), broadcast from there otherwise."""
        from view_sharding import reduce_partials
        if dst is None:
            self.n_seen = int(self._n_seen_slot.item())
        else:
            n = self._n_seen_slot.clone() if dist.get_rank() == dst else torch.zeros(1, dtype=torch.int32, device=self.dev)
            reduce_partials(dist, [n])                   # only the root's slot holds the total; the others contribute zero
            self.n_seen = int(n.item())

    def flush(self):
        """Drain the stream, surface device-side errors, and (parity mode) report the views whose float16 rows held a
        NaN or Inf, as the reference does after every image (AGG:303-304)."""
        if self.ws.buf is not None:
            voxproj_host.workspace_status(self.ws, self.dev)
        self._keep = []
        if self.mode == "parity" and self.n_seen > self._reported:
            bad = torch.nonzero(self._nonfinite[self._reported:self.n_seen]).reshape(-1).tolist()
            for v in bad:
                print(f"[STEP 3][ERROR] NaN or Inf detected in projected features for view {self._reported + v}")
            self._reported = self.n_seen

    def all_reduce(self, dst=None):
        """Combine the ranks' partial {sum, count, views} (fast mode): one SUM collective for the feature sums and one for the
        integers (pixel counts, view counts, views seen -- one tensor) over RCCL.
        ``dst=None`` all-reduces (every rank holds the scene), ``dst=0`` reduces to rank 0 only -- half the traffic, and
        enough when rank 0 alone writes the files (main() does that).  The number of views seen is always all-reduced."""
        import torch.distributed as dist
        assert self.mode != "parity", "the parity mode is order-dependent (fp16 running sums) and stays on one GPU"
        from view_sharding import reduce_partials
        self.flush()
        self._n_seen_slot.fill_(self.n_seen)
        reduce_partials(dist, [self.sum32, self._ints], dst=dst)      # the sums and ONE integer tensor {count, views, n_seen}
        self._finish_n_seen(dist, dst)

    def result(self, xyz_dtype=np.float32):
        """Consolidated tensors in the reference's schema (AGG:381-451)."""
        self.flush()
        if self.mode == "parity":
            ids = torch.nonzero(self.views > 0).reshape(-1)
            order = torch.argsort(self.first_view[ids].long() * self.n_rows + ids)      # dict insertion order
            ids = ids[order]
            avg = (self.run16[ids].float() / self.views[ids].float()[:, None]).to(torch.float16)   # AGG:385
        else:
            ids = torch.nonzero((self.views > 0) & self.valid_id).reshape(-1)
            avg = (self.sum32[ids] / self.views[ids].float()[:, None]).to(torch.float16)
        zyx = self.id_to_zyx[ids].cpu().numpy()
        # AGG:404-407: np.array([x,y,z]) * VOXEL_SIZE + np.array(GRID_ORIGIN) in float64, stored as float32
        xyz64 = zyx[:, [2, 1, 0]].astype(np.int64) * self.voxel_size + np.array(self.grid_origin, dtype=np.float64)
        out = dict(xyz=torch.from_numpy(xyz64.astype(xyz_dtype)), avg_feats=avg.cpu(),
                   voxel_coords=torch.from_numpy(zyx.astype(np.int32)),
                   hit_count=self.views[ids].to(torch.int32).cpu())
        if self.mode != "parity":
            out.update(sum=self.sum32[ids].cpu(), count=self.count[ids].cpu(), voxel_ids=ids.to(torch.int32).cpu())
        return out


def write_feature_ply(path, xyz_f32, avg_feats_f16):
    """ALL_nonzero_voxels_with_features_*.ply (AGG:425-440): ASCII, float x y z, and -- when there are at least three
    feature channels -- the first three, clipped to [0,1] and scaled to uchar, as red green blue (the arithmetic stays in
    the features' own dtype, float16, as numpy does it there: np.clip(avg_feats[i,:3], 0, 1) * 255 -> astype(uint8))."""
    xyz = np.asarray(xyz_f32, dtype=np.float32)
    feats = np.asarray(avg_feats_f16)
    with_rgb = feats.ndim == 2 and feats.shape[1] >= 3
    if with_rgb:
        with np.errstate(invalid="ignore"):
            rgb = (np.clip(feats[:, :3], 0, 1) * 255).astype(np.uint8)
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\n")
        f.write(f"element vertex {xyz.shape[0]}\n")
        f.write("property float x\nproperty float y\nproperty float z\n")
        if with_rgb:
            f.write("property uchar red\nproperty uchar green\nproperty uchar blue\n")
        f.write("end_header\n")
        for i, pt in enumerate(xyz):
            line = f"{pt[0]} {pt[1]} {pt[2]}"
            if with_rgb:
                line += f" {rgb[i, 0]} {rgb[i, 1]} {rgb[i, 2]}"
            f.write(line + "\n")


def _granted_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup quota (a one-GPU box of the pool shows 256
    logical CPUs and grants 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def _npy_layout(path):
    """(data offset, shape, dtype) of a C-ordered .npy file, or None when it needs numpy's general loader (Fortran order,
    object arrays, pickles): numpy.lib.format's own header parser, no data read."""
    from numpy.lib import format as npf
    with open(path, "rb") as f:
        major, minor = npf.read_magic(f)
        read = {(1, 0): npf.read_array_header_1_0, (2, 0): npf.read_array_header_2_0}.get((major, minor))
        if read is None:
            return None
        shape, fortran, dtype = read(f)
        # plain native-endian numbers only: the bytes go to the device as they are and are viewed as the torch dtype of the same name
        if fortran or dtype.hasobject or dtype.fields is not None or not dtype.isnative or not isinstance(getattr(torch, dtype.name, None), torch.dtype):
            return None
        return f.tell(), tuple(int(v) for v in shape), dtype


def _load_native(path):
    """numpy's loader, C-ordered and in native byte order (what torch.from_numpy accepts)."""
    a = np.load(path)
    if not a.dtype.isnative:
        a = a.astype(a.dtype.newbyteorder("="))
    return np.ascontiguousarray(a)


class FeatureFeeder:
    """Feeds the .npy feature maps to the GPU ahead of the projector (the reference re-reads each map from disk three times
    per view through its sub-processes, AGG:248-294).

    A 199 MB map costs 3.7 ms on PCIe (the floor, profiles/r02_bench_prep.json) but 20-40 ms when ONE thread copies it out
    of the page cache -- the host copy, not the bus, set the pace of the round-2 feeder.  So every file is cut into
    ``CHUNK``-byte pieces that a pool of I/O threads ``preadv`` STRAIGHT into a pinned staging buffer (no intermediate
    array; the read releases the GIL), up to ``depth`` views ahead, and the consumer's thread issues the host-to-device copy
    PER PIECE, in order, on a copy stream of its own as soon as that piece has landed: the bus starts on a map while its
    tail is still being read, and reading view k+2, copying view k+1 and projecting view k overlap.

    Iterating yields ``(index, path, device_tensor)`` in file order, the tensor shaped and typed like the file's array and
    ready on torch's current stream.  ``depth = 0`` reads and copies synchronously (same results); files that are not
    plain C-ordered arrays go through ``numpy.load`` as a whole."""

    CHUNK = 16 << 20

    def __init__(self, paths, device, depth=3, io_threads=None):
        self.paths, self.dev, self.depth = list(paths), torch.device(device), int(depth)
        if io_threads is None:
            io_threads = max(2, min(16, _granted_cpus() - 2))     # a page-cache read is a memcpy: ~3 GB/s per thread
        self.io_threads = int(io_threads)

    def __iter__(self):
        if self.depth <= 0:
            for i, p in enumerate(self.paths):
                yield i, p, torch.from_numpy(_load_native(p)).to(self.dev)
            return
        from concurrent.futures import ThreadPoolExecutor
        nbuf = self.depth + 2
        pinned = [None] * nbuf        # uint8 staging buffers, grown on demand
        copied = [None] * nbuf        # event of the last H2D copy that read each buffer

        def read_piece(path, fd, view, off, file_off, n):
            done = 0
            while done < n:           # preadv may return short
                got = os.preadv(fd, [view[off + done:off + n]], file_off + done)
                if got <= 0:
                    raise IOError(f"{path}: unexpected end of file at byte {file_off + done} (its header promises {file_off - off + len(view)} bytes or more)")
                done += got
            return n

        def close_file(fd, futs):
            """Close a file only when none of its piece reads can still be inside preadv on it (the descriptor's number may be
            handed to the next os.open at once: a straggler would read another file's bytes into the staging buffer)."""
            if fd is None:
                return
            for f in futs:
                f.cancel()
            from concurrent.futures import wait
            wait(futs)
            os.close(fd)

        def submit(pool, i):
            """Queue the reads of file i: (layout, buffer slot, fd, [future per piece]) -- or a whole-array fallback."""
            lay = _npy_layout(self.paths[i])
            b = i % nbuf
            if copied[b] is not None:
                copied[b].synchronize()                          # view i - nbuf has left this buffer
                copied[b] = None
            if lay is None:
                return None, b, None, [pool.submit(_load_native, self.paths[i])]
            off0, shape, dtype = lay
            nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
            if pinned[b] is None or pinned[b].numel() < nbytes:
                pinned[b] = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True)
            view = memoryview(pinned[b].numpy())
            fd = os.open(self.paths[i], os.O_RDONLY)
            futs = [pool.submit(read_piece, self.paths[i], fd, view, o, off0 + o, min(self.CHUNK, nbytes - o)) for o in range(0, nbytes, self.CHUNK)]
            return (shape, dtype, nbytes), b, fd, futs

        copy_stream = torch.cuda.Stream(self.dev, priority=-1)     # a queue of its own (see _to_device_ready)
        with ThreadPoolExecutor(max_workers=self.io_threads) as pool:
            pending = []
            try:
                pending = [submit(pool, i) for i in range(min(self.depth, len(self.paths)))]
                for i, p in enumerate(self.paths):
                    lay, b, fd, futs = pending.pop(0)
                    with torch.cuda.stream(copy_stream):
                        if lay is None:
                            host = torch.from_numpy(futs[0].result())
                            d = host.to(self.dev)                    # pageable: blocking, rare path
                        else:
                            shape, dtype, nbytes = lay
                            raw = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=self.dev)
                            o = 0
                            try:
                                for f in futs:                       # in order: the bus follows the readers piece by piece
                                    n = f.result()
                                    raw[o:o + n].copy_(pinned[b][o:o + n], non_blocking=True)
                                    o += n
                            finally:
                                close_file(fd, futs)
                            d = raw[:nbytes].view(getattr(torch, np.dtype(dtype).name)).reshape(shape)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    copied[b] = ev
                    if i + self.depth < len(self.paths):             # submitted only now: its buffer's previous copy is recorded
                        pending.append(submit(pool, i + self.depth))
                    cur = torch.cuda.current_stream(self.dev)
                    cur.wait_event(ev)
                    d.record_stream(cur)                             # allocated on the copy stream, consumed on this one
                    yield i, p, d
            finally:
                # the consumer stopped early (an exception on its side, a break) or a file failed: the files read ahead are
                # still open, their reads may still be running
                for _, _, fd, futs in pending:
                    close_file(fd, futs)
                copy_stream.synchronize()                            # the staging buffers outlive the copies that read them


def _image_size(entry, cams, images_dir, name):
    """(H_orig, W_orig) of a view: the reference reads the image file (AGG:210-214); the camera JSON's
    width/height are used when the image is not there."""
    if images_dir:
        for ext in (".jpg", ".jpeg", ".png", ".JPG", ".JPEG", ".PNG", ""):
            p = os.path.join(images_dir, name + ext)
            if os.path.exists(p):
                from PIL import Image
                with Image.open(p) as im:
                    return im.height, im.width
    cam = cams[str(entry["camera_id"])]
    return int(cam["height"]), int(cam["width"])


def main(argv=None, timing=None):
    """``timing``: optional dict that receives ``loop_s`` (wall time of the per-view loop: first file requested to last
    projector call drained) and ``views`` -- what tools/bench_entry_files.py reports per view, free of the PLY parse and of
    writing the result files."""
    import time
    ap = argparse.ArgumentParser(description="Aggregate voxel features pipeline")
    ap.add_argument("--first_only", action="store_true", help="Only process the first input image for debug")
    ap.add_argument("--mode", choices=("parity", "fast"), default="parity")
    ap.add_argument("--lseg_dir", default=os.environ.get("LSEG_DIR", LSEG_DIR))
    ap.add_argument("--cam_params", default=os.environ.get("CAM_PARAMS_ORIG", CAM_PARAMS_ORIG))
    ap.add_argument("--voxel_ply", default=os.environ.get("VOXEL_PLY", VOXEL_PLY))
    ap.add_argument("--images_dir", default=os.environ.get("IMAGES_DIR", ""))
    ap.add_argument("--checkpoint_dir", default=os.environ.get("CHECKPOINT_DIR", CHECKPOINT_DIR))
    ap.add_argument("--max_images", type=int, default=MAX_IMAGES)
    ap.add_argument("--downsample_factor", type=float, default=DOWNSAMPLE_FACTOR)
    ap.add_argument("--views_per_call", type=int, default=8, help="fast mode: views per projector call")
    ap.add_argument("--half_features", action="store_true", help="fast mode: keep the (fp16-valued) feature maps in "
                    "fp16 on the GPU (vp_project_features_f16): half the HBM traffic, identical results")
    ap.add_argument("--no_split_collective", action="store_true", help="several ranks: project the rank's last call whole and "
                    "reduce afterwards (A/B arm of add_final_views' row-range split)")
    ap.add_argument("--prefetch", type=int, default=3, help="feature files read ahead of the GPU by worker threads and "
                    "copied over PCIe on a copy stream (0 = read and copy synchronously)")
    args = ap.parse_args(argv)

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    # VOXPROJ_SINGLE_DEVICE=1 + VOXPROJ_DIST_BACKEND=gloo: rehearse the multi-rank path on a one-GPU box
    single = os.environ.get("VOXPROJ_SINGLE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if single else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        assert args.mode == "fast", "multi-GPU aggregation needs --mode fast"
        backend = os.environ.get("VOXPROJ_DIST_BACKEND", "nccl")          # nccl IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    os.makedirs(args.checkpoint_dir, exist_ok=True)

    voxel_size, grid_origin, grid_shape, n_from_name = bso.extract_voxel_params(args.voxel_ply)   # AGG:92
    num_voxels = n_from_name if n_from_name is not None else (int(np.prod(grid_shape)) if grid_shape else "unknown")
    feature_files = sorted(glob.glob(os.path.join(args.lseg_dir, "*.npy")))                       # AGG:101
    if not feature_files:
        raise RuntimeError(f"No .npy feature files found in {args.lseg_dir}")
    feature_files = feature_files[:args.max_images]
    if args.first_only:
        feature_files = feature_files[:1]
    occ = bso.build_occupancy(bso.read_voxel_ply(args.voxel_ply), grid_origin, voxel_size, device=dev)   # AGG:118-127
    by_name, cams = ptd.load_camera_params(args.cam_params)

    agg, idx = None, 0
    # Views are up-sampled STRAIGHT into a slot of a small ring of call-sized device buffers [per_call, H, W, C] (round 2
    # stacked the per-view tensors: an extra pass over 1 GB per view and a fresh 8 GB allocation per call).  Stream order
    # keeps a slot safe: the up-sample that refills it is queued on the same stream behind the gather that read it.
    per_call = 1 if args.mode == "parity" else max(1, args.views_per_call)
    ring, ring_key, slot, fill = None, None, 0, 0
    batch_c, batch_intr = [], None
    keep_dtype = args.half_features and args.mode == "fast"

    def submit():
        nonlocal slot, fill, batch_c
        if fill:
            agg.add_views(ring[slot][:fill], torch.stack(batch_c), batch_intr)
            slot, fill, batch_c = (slot + 1) % len(ring), 0, []

    from view_sharding import views_of_rank
    mine = [feature_files[i] for i in views_of_rank(len(feature_files), rank, world)]
    usable = [f for f in mine if by_name.get(os.path.basename(f)[:-4]) is not None]
    for f in mine:
        if by_name.get(os.path.basename(f)[:-4]) is None:
            print(f"[ERROR] No camera entry for {os.path.basename(f)[:-4]}")
    position = {f: k for k, f in enumerate(mine)}                  # idx counts every file, as AGG:316 does
    t_loop = time.perf_counter()
    for _, fpath, raw in FeatureFeeder(usable, dev, args.prefetch):
        k = position[fpath]
        name = os.path.basename(fpath)[:-4]
        entry = by_name[name]
        H0, W0 = _image_size(entry, cams, args.images_dir, name)
        H_new, W_new = int(H0 * args.downsample_factor), int(W0 * args.downsample_factor)          # AGG:215
        intr, c2w = ptd.camera_for(entry, cams, args.downsample_factor)                            # PTD:132-172
        C_in = int(raw.shape[0])
        dt = torch.float16 if (keep_dtype and raw.dtype == torch.float16) else torch.float32
        key = (H_new, W_new, C_in, dt)
        if agg is None:
            agg = VoxelFeatureAggregator(occ, grid_origin, voxel_size, C_in, args.mode, dev)
        if key != ring_key or (fill and not torch.equal(batch_intr, intr)) or fill >= per_call:
            submit()
        if key != ring_key:
            if ring is not None:
                agg.flush()                                        # the old ring's calls are done before it goes away
            ring = [torch.empty((per_call, H_new, W_new, C_in), dtype=dt, device=dev) for _ in range(2 if per_call > 1 else 3)]
            ring_key, slot = key, 0
        ptd.upsample_features(raw, (H_new, W_new), device=dev, keep_dtype=keep_dtype, out=ring[slot][fill])   # PTD:115-127
        fill += 1
        batch_c.append(c2w)
        batch_intr = intr
        idx = k + 1
        # AGG:318-352: a consolidated checkpoint every 20 views (single process: with several ranks no rank holds the
        # scene before the final reduction, and the reference has no resume path that would read these files)
        if world == 1 and idx % CHECKPOINT_EVERY == 0:
            submit()
            r = agg.result(xyz_dtype=np.float64)
            torch.save({k2: r[k2] for k2 in ("xyz", "avg_feats", "hit_count", "voxel_coords")},
                       os.path.join(args.checkpoint_dir, f"checkpoint_features_{idx}.pt"))
            print(f"[CHECKPOINT] Saved consolidated checkpoint data after {idx} images")
    if world > 1:
        # every rank must reach the collectives: agree first that each of them had something to project
        ok = torch.tensor([int(agg is not None)], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            raise RuntimeError("a rank had no usable view (fewer views with camera entries than ranks)")
        # the rank's last call and the scene's collective together: rows below the cut are reduced under the gather of the
        # rows above it.  Rank 0 alone writes the files: a reduce, not an all-reduce
        last = (ring[slot][:fill], torch.stack(batch_c), batch_intr) if fill else (None, None, None)
        agg.add_final_views(*last, dst=0, split=not args.no_split_collective)
        fill, batch_c = 0, []
    else:
        submit()
    if timing is not None:
        if agg is not None:
            agg.flush()
        timing.update(loop_s=time.perf_counter() - t_loop, views=len(usable))
    if world == 1 and agg is None:
        raise RuntimeError("no view could be processed")
    if rank == 0:
        r = agg.result()
        n_done = agg.n_seen
        if r["xyz"].shape[0] == 0:
            print("[DONE] No occupied voxels in final aggregation, skipping export.")
        else:
            if not torch.isfinite(r["avg_feats"].float()).all():                                          # AGG:303-304
                print("[STEP 3][ERROR] NaN or Inf detected in the aggregated features")
            ply_path = os.path.join(args.checkpoint_dir, f"ALL_nonzero_voxels_with_features_{n_done}_vox{num_voxels}.ply")
            write_feature_ply(ply_path, r["xyz"].numpy(), r["avg_feats"].numpy())                        # AGG:425-440
            print(f"[PLY] Saved nonzero voxels with features as: {ply_path}")
            save_path = os.path.join(args.checkpoint_dir, f"ALL_nonzero_voxel_features_{n_done}_vox{num_voxels}.pt")
            torch.save({"xyz": r["xyz"], "avg_feats": r["avg_feats"], "voxel_coords": r["voxel_coords"]}, save_path)   # AGG:447-451
            print(f"[PT] Saved filtered and compressed voxel data (xyz, features, coords) as: {save_path}")
            if args.mode == "fast":
                torch.save(r, save_path[:-3] + "_fp32.pt")
        print(f"[DONE] PROJECTION PIPELINE COMPLETED. Checkpoint directory: {args.checkpoint_dir}")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
