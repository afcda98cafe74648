#!/usr/bin/env python3
"""bench.py -- Mvoxel-views/s + achieved HBM GB/s of the 2D -> sparse-voxel feature projector.

Workload (BASELINE.json metric config, "R2"): 200 000 occupied voxels x 300 views x 968x548x512 fp32
synthetic feature maps (SURVEY.md section 8d generator, seed 0).  One STEP = one pass of the hot path
over the whole scene: every view's feature map is read from HBM exactly once, in calls of --chunk views
through the C-ABI (vp_project_features), features already resident in HBM.  326 GB of feature maps do
not fit one GPU, so a pool of --pool distinct maps (default 32 = 34.8 GB, far beyond the 256 MiB Infinity
Cache) is cycled; the rays, the voxel assignment and the bytes moved are those of 300 distinct views.

Set-up, untimed: the pool and the output rows are allocated --pool-tries times and the placement with the fastest
pass is kept (their physical placement moves the gather by several per cent; every try is in the JSON line under
"pool_placement", --pool-tries 1 takes the first allocation as it comes).

Multi-GPU (torchrun, one rank per GPU): rank r projects views r::G of the same 300-view scene, then one
RCCL all-reduce of the per-voxel {feature-sum f32 [N+1,512], hit-count i32 [N+1]} -- total work fixed,
"scaling": "strong".

Prints ONE JSON line on rank 0 (contract in the task statement) with "roofline" for the dominant kernel
(k_gather, HBM-bound; HIP events on its launch stream, recorded live during the timed steps) and
"cpu_baseline" (the CPU oracle -- a port of the reference kernel -- timed on this box's host cores on a
bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "3d-semantic-segmentation_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (n_vox, n_views, W, H, C)
    "R2": (200000, 300, 968, 548, 512),      # BASELINE config 3 (metric config)
    "R1": (80000, 100, 484, 274, 512),       # BASELINE config 2
    "S0": (10000, 8, 64, 64, 32),            # BASELINE config 1 (plumbing)
}
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="R2", choices=sorted(WORKLOADS))
    ap.add_argument("--chunk", type=int, default=32, help="views per vp_project_features call")
    ap.add_argument("--pool", type=int, default=32, help="distinct resident feature maps")
    ap.add_argument("--views", type=int, default=0, help="override the number of views (0 = workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="plain calls: phase 1 and 2 serial on one stream")
    ap.add_argument("--cpu-views", type=int, default=8, help="views in the cpu_baseline sample")
    ap.add_argument("--dtype", default="f32", choices=("f32", "f16"), help="feature-map storage type; f32 is the "
                    "BASELINE metric config, f16 the lossless half-bandwidth mode of SURVEY 8f/n4 (extra, not the headline)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on a multi-GPU node; gloo only to rehearse "
                    "the multi-rank code path on a single-GPU box (together with --single-device)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--pool-tries", type=int, default=5,
                    help="allocate the resident feature pool and the output rows this many times and keep the placement "
                         "with the fastest pass (untimed set-up; every try is reported in the JSON line); 1 = take the "
                         "first allocation as it comes")
    ap.add_argument("--no-overlap-reduce", action="store_true",
                    help="multi-GPU: wait for each pass's all-reduce before starting the next pass (default: the "
                         "all-reduce of pass k runs on RCCL's stream while pass k+1 is projected into a second buffer)")
    return ap.parse_args()


def pmc_traffic(workload, chunk):
    """HBM bytes per k_gather launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json),
    corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x2 for 16-B-per-lane streaming reads, KB units).
    Returned per VIEW of a full launch (the caller scales by its average views per launch); None unless the
    profile was taken on this workload / views-per-call."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            prof = json.load(f)
    except OSError:
        return None
    if prof.get("workload") != workload or prof.get("views_per_call") != chunk:
        return None
    g = prof["k_gather"]
    return (2.0 * g["FETCH_SIZE_KB_per_launch"] + g["WRITE_SIZE_KB_per_launch"]) * 1024 / prof["views_per_call"]


def host_cores():
    """CPU threads this process may really use: the affinity mask capped by the cgroup CPU quota (a one-GPU box of
    the pool shows 256 logical CPUs but grants 16; 256 OpenMP threads on that share run the oracle 7x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(scene, C, n_views, n_threads):
    """Time the CPU oracle (port of project_image_cuda_kernel.cu:24-92,157-187) on n_views views."""
    from oracle import oracle
    from synthetic_scene import make_features_np
    feats = make_features_np(n_views, scene.height, scene.width, C, seed=0)[None]
    n_rows = scene.n_vox + 1
    count = np.zeros(n_rows, np.int32)
    out = np.zeros((n_rows, C), np.float32)
    occ = scene.occ[None].astype(np.int64)
    t0 = time.perf_counter()
    oracle.project_features(feats, occ, scene.c2w[:n_views].reshape(-1), scene.intr[None], scene.opts(),
                            scene.grid_origin, scene.voxel_size, count, out, want_hits=False, nthreads=n_threads)
    dt = time.perf_counter() - t0
    return dict(value=round(scene.n_vox * n_views / dt / 1e6, 4), unit="Mvoxel-views/s", cores=n_threads, kind="port",
                sample=f"{n_views} of the workload's views at full resolution, all {scene.n_vox} voxels, "
                       f"{dt:.1f} s wall (OpenMP over pixel rows + channel slices; {os.cpu_count()} logical CPUs "
                       f"visible, {n_threads} granted to this process)")


def cpu_torch_loop(scene, n_views, n_threads):
    """The reference's only CPU projection loop (debug_project_features.py:59-84: every occupied voxel centre
    through the pinhole model, in-front and in-image tests) as the vectorised torch-CPU expression of
    debug_project_features.voxel_centre_diagnostics -- no occlusion test, no feature gather, so it is NOT the
    same work as the projector; reported beside cpu_baseline because north_star names it."""
    from debug_project_features import voxel_centre_diagnostics
    torch.set_num_threads(n_threads)
    occ = torch.from_numpy(scene.occ)
    c2w = torch.from_numpy(scene.c2w)
    intr = torch.from_numpy(scene.intr)
    origin = torch.from_numpy(np.asarray(scene.grid_origin, dtype=np.float32))
    voxel_centre_diagnostics(occ, c2w[0], intr, origin, scene.voxel_size, scene.width, scene.height)   # warm
    t0 = time.perf_counter()
    inb = 0
    for v in range(n_views):
        inb += voxel_centre_diagnostics(occ, c2w[v], intr, origin, scene.voxel_size, scene.width, scene.height)["n_in_bounds"]
    dt = time.perf_counter() - t0
    return dict(value=round(scene.n_vox * n_views / dt / 1e6, 3), unit="Mvoxel-views/s", cores=n_threads,
                what="voxel-centre projection + bounds test only (DPF:59-84), float64 torch-CPU, vectorised",
                sample=f"{n_views} views x {scene.n_vox} voxels, {dt:.2f} s wall, {inb} centres in bounds")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the projector has no CPU fallback")
    dev = torch.device("cuda", 0 if a.single_device else local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.dist_backend)

    import voxproj_host
    from synthetic_scene import make_features_torch, make_scene

    n_vox, n_views, W, H, C = WORKLOADS[a.workload]
    C = int(os.environ.get("VOXPROJ_BENCH_C", C))      # diagnostic only (row-size experiments)
    if a.views:
        n_views = a.views
    scene = make_scene(n_vox, n_views, W, H, seed=0)
    from view_sharding import reduce_partials, views_of_rank
    my_views = views_of_rank(n_views, rank, world)
    # at least four calls per rank so that the pipelined mode can hide phase 1 of all but the first call
    chunk = max(1, min(a.chunk, len(my_views), max(4, -(-len(my_views) // 4))))
    pool = max(chunk, (min(a.pool, len(my_views)) // chunk) * chunk)

    esize = 4 if a.dtype == "f32" else 2

    def alloc_pool():
        if a.dtype == "f32":
            f = torch.empty((1, pool, H, W, C), dtype=torch.float32, device=dev)
            make_features_torch(pool, H, W, C, dev, seed=0, out=f[0])
        else:
            f = torch.empty((1, pool, H, W, C), dtype=torch.float16, device=dev)
            for v in range(pool):
                f[0, v] = make_features_torch(1, H, W, C, dev, seed=v)[0].half()
        return f

    occ = torch.from_numpy(scene.occ[None].astype(np.int64)).to(dev)
    c2w = torch.from_numpy(scene.c2w).to(dev)
    intr = torch.from_numpy(scene.intr[None]).to(dev)
    n_rows = n_vox + 1
    count = out = feats = None       # allocated by the placement loop below
    opts = [float(v) for v in scene.opts()]
    origin = [float(v) for v in scene.grid_origin]
    ws = voxproj_host.Workspace()

    # calls of one step: (pool slot of the first view, view indices)
    calls = []
    for i in range(0, len(my_views), chunk):
        vs = my_views[i:i + chunk]
        slot = (i % pool) if (i % pool) + len(vs) <= pool else 0
        calls.append((slot, vs))
    vmis = [c2w[vs].reshape(-1).contiguous() for _, vs in calls]

    pipeline = not a.no_pipeline

    def one_call(ci, sync=False, o=None, c=None):
        slot, vs = calls[ci]
        voxproj_host.project_features_raw(feats[:, slot:slot + len(vs)], occ, vmis[ci], intr, opts,
                                          count if c is None else c, out if o is None else o,
                                          origin, scene.voxel_size, workspace=ws, sync=sync,
                                          reuse_accel=(ci > 0 or None), pipeline=(pipeline and not sync))

    # Placement of the resident buffers (untimed set-up).  Where the driver puts the physical pages of the feature
    # pool and of the output rows moves the gather's speed by several per cent from one allocation to the next
    # (tools/probe_placement*.py), stable for the life of the allocation.  A job that will read
    # the pool for minutes can afford to look: allocate, time one whole pass, park the allocation (so that the next
    # one lands elsewhere) and try again; the fastest placement is kept, the others are freed.  Every try is reported.
    placement = {"tries": [], "picked": 0}
    parked, best = [], None
    for t in range(max(1, a.pool_tries)):
        f_try = alloc_pool()
        c_try = torch.zeros(n_rows, dtype=torch.int32, device=dev)
        o_try = torch.zeros(n_rows, C, dtype=torch.float32, device=dev)
        feats = f_try
        ms = 0.0
        for rep in range(2):
            c_try.zero_(); o_try.zero_()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for ci in range(len(calls)):
                one_call(ci, o=o_try, c=c_try)
            voxproj_host.workspace_status(ws, dev)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) * 1e3
        placement["tries"].append({"ms_per_pass": round(ms, 3)})
        if best is None or ms < best[0]:
            best = (ms, t, f_try, c_try, o_try)
        parked.append((f_try, c_try, o_try))
    placement["picked"] = best[1]
    feats, count, out = best[2], best[3], best[4]
    del parked, best, f_try, c_try, o_try
    torch.cuda.empty_cache()

    # multi-GPU: two output buffers, so that the all-reduce of pass k (RCCL's own stream, over xGMI) overlaps the
    # projection of pass k+1; every reduction is waited for before its buffer is reused and before the timed
    # region ends
    bufs = [(out, count)]
    if dist is not None and not a.no_overlap_reduce:
        bufs.append((torch.zeros_like(out), torch.zeros_like(count)))
    inflight = [None] * len(bufs)
    state = {"k": 0}

    def drain(i=None):
        for j in (range(len(bufs)) if i is None else [i]):
            if inflight[j] is not None:
                for w in inflight[j]:
                    w.wait()
                inflight[j] = None

    def step():
        i = state["k"] % len(bufs)
        state["k"] += 1
        o, c = bufs[i]
        drain(i)
        c.zero_()
        o.zero_()
        for ci in range(len(calls)):
            one_call(ci, o=o, c=c)
        if dist is not None:
            if a.no_overlap_reduce:
                reduce_partials(dist, [o, c])
            else:
                inflight[i] = [dist.all_reduce(o, async_op=True), dist.all_reduce(c, async_op=True)]

    # untimed pre-pass: algorithmic bytes of the dominant kernel per launch (deterministic across steps)
    hit_px, touched, gather_bytes, cnt, max_px, heavy_px = 0, 0, 0, {}, 0, 0
    out.zero_()
    for ci in range(len(calls)):
        count.zero_()
        one_call(ci, sync=True)
        ph, nt = int(count.sum().item()), int((count > 0).sum().item())
        hit_px += ph
        touched += nt
        # k_gather proper: voxels above the library's per-call threshold (256 + 64*B*V pixels) are summed by
        # k_gather_heavy, so their rows and output RMW do not count for this kernel
        heavy = count > (int(os.environ.get("VOXPROJ_HEAVY_T", "0")) or (256 + 64 * len(calls[ci][1])))
        ph_heavy, nt_heavy = int(count[heavy].sum().item()), int(heavy.sum().item())
        heavy_px += ph_heavy
        gather_bytes += ((ph - ph_heavy) * C * esize + (nt - nt_heavy) * C * 4 * 2 + len(calls[ci][1]) * H * W * 4
                         + n_rows * 4 * 2)
        c1 = voxproj_host.counters(ws, dev)
        for k in c1:
            cnt[k] = cnt.get(k, 0) + c1[k]
        max_px = max(max_px, int(count.max().item()))

    ref_checksum = out.double().sum(0)       # plain (unpipelined) result of one full pass on this rank
    ref_abs = out.double().abs().sum(0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    drain()
    barrier()
    voxproj_host.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()
    barrier()
    dt = time.perf_counter() - t0
    voxproj_host.workspace_status(ws, dev)
    prof = voxproj_host.profile_read()
    if dist is None and not os.environ.get("VOXPROJ_BENCH_NOVERIFY"):
        # the timed passes must have produced the same result as the plain pre-pass
        assert int(count.sum().item()) == hit_px, "hit-count total changed between the pre-pass and the timed steps"
        assert ((out.double().sum(0) - ref_checksum).abs() <= 1e-6 * ref_abs + 1e-9).all(), "feature sums changed"
        last = voxproj_host.counters(ws, dev)
        assert last["bad_id"] == 0 and last["box_miss"] == 0, last
    voxproj_host.profile_enable(False)
    if dist is None:
        # one more pass timed exactly like the placement tries (after the timed region; diagnostic only)
        for rep in range(2):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for ci in range(len(calls)):
                one_call(ci)
            voxproj_host.workspace_status(ws, dev)
            torch.cuda.synchronize(dev)
            placement["ms_per_pass_after_timed_region"] = round((time.perf_counter() - t1) * 1e3, 3)
    if dist is not None and not os.environ.get("VOXPROJ_BENCH_NOVERIFY"):
        # the buffer reduced last holds the whole scene on every rank: its hit-count total must equal the sum of the
        # ranks' own (pre-pass) totals, exactly
        t = torch.tensor([hit_px], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        last_c = bufs[(state["k"] - 1) % len(bufs)][1]
        assert int(last_c.sum().item()) == int(t.item()), "all-reduced hit counts do not add up to the ranks' totals"
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    algo_local = hit_px * C * esize + touched * C * 4 * 2 + len(calls) * n_rows * 4 * 2 + len(my_views) * H * W * 4 * 2
    if dist is not None:
        t = torch.tensor([algo_local], dtype=torch.float64, device=dev)
        dist.all_reduce(t)
        algo_local = float(t.item())
    stream_gbs = voxproj_host.stream_read_gbs(feats) if rank == 0 else 0.0   # on-box streaming-read ceiling
    if rank == 0:
        ms_step = dt / a.steps * 1e3
        value = n_vox * n_views / (dt / a.steps) / 1e6
        launches = max(prof["gather_launches"], 1)
        gather_ms = prof["gather_ms"] / launches
        ach = (gather_bytes / len(calls)) / (gather_ms * 1e-3) / 1e9 if gather_ms > 0 else 0.0
        # whole-path algorithmic bytes per step: feature rows + output RMW + counts + ID image w+r
        algo_step = algo_local      # summed over ranks
        res = {
            "metric": "Mvoxel-views/sec", "value": round(value, 3), "unit": "Mvoxel-views/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if a.dtype == "f32" else "f32 accumulate, f16 feature maps",
            "data": "synthetic",
            "config": {"workload": f"{a.workload}: {n_vox} voxels x {n_views} views x {W}x{H}x{C} {'fp32' if a.dtype == 'f32' else 'fp16'} feature maps, "
                                   f"room-shell scene seed 0, dmin 0.01 dmax 10 step 0.5*voxel",
                       "views_per_call": chunk, "resident_feature_maps": pool,
                       "parallelism": (f"views r::{world} per GPU + one RCCL all-reduce of sum/count per pass"
                                       + ("" if a.no_overlap_reduce else ", overlapped with the next pass (two output buffers)"))
                       if world > 1 else "single GPU"},
            "achieved_hbm_gbs_whole_path": round(algo_step / (dt / a.steps) / 1e9, 1),
            "phase_ms_per_step": {"prep": round(prof["prep_ms"] / a.steps, 3),
                                  "first_hit": round(prof["first_hit_ms"] / a.steps, 3),
                                  "gather": round(prof["gather_ms"] / a.steps, 3),
                                  "gather_heavy": round(prof["heavy_ms"] / a.steps, 3),
                                  "overlapped": pipeline},
            "pool_placement": placement,
            "hit_pixels_per_step": hit_px, "box_miss_voxels": cnt["box_miss"], "heavy_voxels_per_step": cnt["n_heavy"], "heavy_pixels_per_step": heavy_px, "max_pixels_per_voxel_call": max_px,
            "roofline": {"bound": "hbm", "kernel": "k_gather", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "measured_stream_read_gbs": round(stream_gbs, 1),
                         "frac_of_measured_stream_read": round(ach / stream_gbs, 4) if stream_gbs > 0 else None,
                         "traffic": (int(pmc_traffic(a.workload, chunk) * len(my_views) / len(calls))
                                     if (not a.views and world == 1 and a.dtype == "f32" and pmc_traffic(a.workload, chunk)) else None),
                         "bytes_per_launch": gather_bytes // len(calls), "avg_launch_ms": round(gather_ms, 4)},
        }
        if not a.no_cpu_baseline and world == 1:      # reported on rank 0 at N=1 only (bench contract)
            ncores = host_cores()
            res["cpu_baseline"] = cpu_baseline(scene, C, min(a.cpu_views, n_views), ncores)
            res["cpu_torch_loop"] = cpu_torch_loop(scene, min(16, n_views), ncores)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
