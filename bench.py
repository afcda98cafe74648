#!/usr/bin/env python3
"""bench.py -- Mvoxel-views/s + achieved HBM GB/s of the 2D -> sparse-voxel feature projector.

Workload (BASELINE.json metric config, "R2"): 200 000 occupied voxels x 300 views x 968x548x512 fp32
synthetic feature maps (SURVEY.md section 8d generator, seed 0).  One STEP = one pass of the hot path
over the whole scene: every view's feature map is read from HBM exactly once, in calls of --chunk views
through the C-ABI (vp_project_features), features already resident in HBM.  326 GB of feature maps do
not fit one GPU, so a pool of --pool distinct maps (default: one call's worth, 60 maps = 65 GB, far beyond the 256 MiB Infinity
Cache) is cycled; the rays, the voxel assignment and the bytes moved are those of 300 distinct views.

Set-up, untimed: the pool and the output rows are allocated --pool-tries times and the placement with the fastest
pass is kept (their physical placement moves the gather by several per cent; every try is in the JSON line under
"pool_placement", --pool-tries 1 takes the first allocation as it comes).

Multi-GPU (one rank per GPU; under torchrun, or started by this file itself: `python bench.py --gpus N` with no WORLD_SIZE in the
environment runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process before anything touches
a GPU and exits with its code): rank r projects views r::G of the same 300-view scene through the entry point's own
VoxelFeatureAggregator (fast mode), then one RCCL all-reduce of the per-voxel {feature-sum f32 [N+1,512], hit-count i32 [N+1],
view-count i32 [N+1]} -- the rank's last call cut by voxel ID so that half of the sums travel under the other half's gather
(view_sharding.project_final_call_and_reduce); total work fixed, "scaling": "strong".

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

# the pool's host driver only supports dmabuf IPC: without this RCCL fails with "hipIpcGetMemHandle: invalid argument"
# (already exported on the GPU boxes; set here too so that a bare torchrun of this file works)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "3d-semantic-segmentation_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# measurement plumbing kept out of this file (tools/): the timed region and the line are here
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_calibrate import choose_arm  # noqa: E402
from bench_cpu import (colour_projection_torch_cpu, cpu_baseline, cpu_colour_loop, cpu_torch_loop, distinct_image_lines,  # noqa: E402,F401
                       host_cores)
from bench_launch import launch_check, launch_ranks  # noqa: E402
from bench_pmc import PMC_PROFILE, pmc_traffic, source_digest, write_pmc_json, write_r4_pmc_json  # noqa: E402

WORKLOADS = {
    # name: (n_vox, n_views, W, H, C)
    "R2": (200000, 300, 968, 548, 512),      # BASELINE config 3 (metric config)
    "R1": (80000, 100, 484, 274, 512),       # BASELINE config 2
    "S0": (10000, 8, 64, 64, 32),            # BASELINE config 1 (plumbing)
    # round 5: away from the benign room
    "A1": (87319, 216, 876, 584, 512),       # the problem size the reference's authors ran (AGG:18,28,106,209; cell 0.04 m, DSLR
                                             # intrinsics of camera_params/colmap_camera_params.sh:7-14 at 0.5x) on a hand-held trajectory
    "R2T": (200000, 300, 968, 548, 512),     # the metric config's shape on a hand-held trajectory (close-ups, missing wall, clutter 0.5)
}
# generator arguments per workload (synthetic_scene.make_scene); absent: the benign room of SURVEY 8d
SCENE_KW = {
    "A1": dict(trajectory=True, room=(4.4, 3.5, 2.5), voxel_size=0.04),
    "R2T": dict(trajectory=True),
}
SCENE_NAME = {"A1": "hand-held trajectory scene seed 0 (close-up dwells, opening in a wall and the ceiling, clutter 0.5; cell 0.04 m)",
              "R2T": "hand-held trajectory scene seed 0 (close-up dwells, opening in a wall and the ceiling, clutter 0.5)"}


SCENE_SEED = 0      # --scene-seed: another clutter layout and hand tremor (trajectory legs) / other poses and boxes (benign room)


def workload_scene(name, n_views=None, W=None, H=None):
    """The synthetic scene of a named workload (n_views / W / H override the workload's: tests march reduced images)."""
    from synthetic_scene import make_scene
    n_vox, V, w, h, _ = WORKLOADS[name]
    return make_scene(n_vox, n_views or V, W or w, H or h, seed=SCENE_SEED, **SCENE_KW.get(name, {}))
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-line-count", action="store_true", help="R4: skip the exact count of distinct image lines (an upper bound takes its place in bytes_per_launch)")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS) + ["R4"],
                    help="R2 (default) = BASELINE metric config; R1 = config 2; S0 = config 1; R4 = config 5, the RGB path "
                         "(500k voxels x 1000 views, uint8 images; its own kernel, no HBM-roofline claim); A1 = the problem size the "
                         "reference's authors ran (87 319 voxels of 0.04 m x 216 views x 876x584x512) on a hand-held trajectory; R2T = "
                         "the metric config's shape on a hand-held trajectory (close-up dwells, 29 %% of the rays miss, clutter 0.5)")
    ap.add_argument("--entry", default=None, choices=("parity", "fast"),
                    help="time the named entry point's in-process aggregation (VoxelFeatureAggregator.add_views over every "
                         "view of the workload, features resident) instead of raw C-ABI calls; default workload R1")
    ap.add_argument("--entry-no-pipeline", action="store_true",
                    help="--entry parity: A/B arm, plain asynchronous one-view calls instead of the aggregator's default, the "
                         "pipelined job mode (VP_FLAG_PIPELINE)")
    ap.add_argument("--collective", default="allreduce", choices=("reduce", "allreduce"),
                    help="multi-GPU: how the per-rank {sum,count} are combined each pass.  allreduce (default) = the single RCCL "
                         "all-reduce north_star names, every rank gets the scene; reduce = to rank 0 only, half the xGMI "
                         "traffic, enough when one rank writes the scene (what the entry point does)")
    ap.add_argument("--chunk", type=int, default=0,
                    help="views per vp_project_features call; 0 (default) = as many as hold --call-gb of feature maps (R2 fp32: "
                         "60, R2 fp16: 100, R1: 50), then evened out over the rank's calls")
    ap.add_argument("--call-gb", type=float, default=66.0,
                    help="feature-map bytes per call the automatic --chunk aims for (SURVEY 8d: chunks of <= 64 resident views, "
                         "69.5 GB).  Every call read-modify-writes the output rows it touches, and those stores cost far more "
                         "than their bytes (DESIGN.md section 4): 60 views per call instead of 30 = +2.7 %% on a slow-level box")
    ap.add_argument("--min-calls", type=int, default=None,
                    help="cut a rank's views into at least this many calls (in pipelined mode the march of every call but the "
                         "first hides under the previous gather).  Default 2 where passes follow one another (one GPU: the "
                         "first march of pass k+1 hides under the last gather of pass k); 1 in the multi-rank step, where a "
                         "pass stands alone: its first march is exposed either way, and 38 views of R2 project in 7.91 ms as "
                         "one call against 8.19 ms as two (the second gather re-reads the output rows and the short launches "
                         "fill the chip worse; profiles/r03_rank_workloads.log)")
    ap.add_argument("--pool", type=int, default=32, help="distinct resident feature maps")
    ap.add_argument("--views", type=int, default=0, help="override the number of views (0 = workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="plain calls: phase 1 and 2 serial on one stream")
    ap.add_argument("--cpu-views", type=int, default=96, help="views in the cpu_baseline sample (about 10 s on 16 cores at R2; the single-thread sample is an eighth of it)")
    ap.add_argument("--dtype", default="f32", choices=("f32", "f16"), help="feature-map storage type; f32 is the "
                    "BASELINE metric config, f16 the lossless half-bandwidth mode of SURVEY 8f/n4 (extra, not the headline)")
    ap.add_argument("--rehearse-dist", action="store_true", help="with one process: create the process group anyway (RCCL "
                    "communicator of ONE rank) and run the multi-rank code path -- collective inside the pass, verification, "
                    "both collective arms -- on a one-GPU box.  A rehearsal of the control flow and of the RCCL calls, not a "
                    "scaling number: a one-rank collective moves nothing over xGMI")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on a multi-GPU node; gloo only to rehearse "
                    "the multi-rank code path on a single-GPU box (together with --single-device)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--pool-tries", type=int, default=1,
                    help="1 (default) = take the first allocation of the resident feature pool and the output rows as it "
                         "comes; N > 1 = opt-in placement search: allocate N times during the untimed set-up, keep the "
                         "placement with the fastest pass (every try is reported in the JSON line)")
    ap.add_argument("--alloc", default="default", choices=("contiguous", "default"),
                    help="how this job allocates its resident buffers (feature pool, output rows): default = torch's allocator "
                         "(plain hipMalloc); contiguous = physically contiguous device memory (hipExtMallocWithFlags, "
                         "hipDeviceMallocContiguous; falls back to the plain allocator if no such range is free) -- an experiment "
                         "on the placement spread, DESIGN.md section 4: it pins a 17 GB pool to one speed level inside a process, "
                         "but does not remove the spread between processes")
    ap.add_argument("--heavy-threshold", type=int, default=0,
                    help="experiment: VP_OPT_HEAVY_THRESHOLD of the workspace (pixels per voxel and call above which a whole "
                         "workgroup sums the voxel); 0 = the library's default, min(256 + 64 x views per call, 2048)")
    ap.add_argument("--part-pixels", type=int, default=0,
                    help="experiment: VP_OPT_PART_PIXELS of the workspace (pixels per part of a voxel above the heavy threshold); "
                         "0 = the library's default: the heavy threshold, min(256 + 64 * views per call, 2048), raised to "
                         "2 * pixels of the call / part slots where that binds")
    ap.add_argument("--march-lds-kb", type=int, default=-1,
                    help="experiment: VP_OPT_MARCH_LDS_KB of the workspace (dynamic-LDS reservation of the march = its occupancy "
                         "cap beside a gather); -1 = the library's default (41 KiB = 3 workgroups per CU)")
    ap.add_argument("--one-view-gather", type=int, default=-1,
                    help="experiment: VP_OPT_ONE_VIEW_GATHER of every workspace (one-view calls: 0 = the general gather kernel, n > 0 = "
                         "the one-view kernel with n workgroups per CU); -1 = the library's default")
    ap.add_argument("--no-split-collective", action="store_true",
                    help="multi-rank step: the TIMED arm does not cut the pass's last call into two row ranges (VP_OPT_ROW_BEGIN/_END "
                         "+ VP_FLAG_GATHER_ONLY) whose first half is all-reduced under the second half's gather.  Either way the "
                         "other arm is run after the timed region and both are reported (collective.arms)")
    ap.add_argument("--collective-arm", default="auto", choices=("auto", "split", "whole"),
                    help="multi-rank step: which collective arm the timed region runs.  auto (default) = a short untimed calibration "
                         "of both arms (2 steps each, MAX over ranks) picks the faster one; the other arm runs after the timed "
                         "region either way and both are reported (collective.arms, collective.calibration)")
    ap.add_argument("--scene-seed", type=int, default=0,
                    help="seed of the synthetic scene (default 0 = the scene every committed number is quoted on; other seeds: another "
                         "clutter layout and hand tremor on the trajectory legs -- a check that nothing is tuned to one scene)")
    ap.add_argument("--no-other-arm", action="store_true",
                    help="multi-rank step: skip the steps of the other collective arm after the timed region")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU work: every rank joins a gloo process group, takes its share of the views, plans its calls, and "
                         "rank 0 prints one line with n_gpus = the group's size (value null).  What tests/ run on CPU to see that "
                         "`bench.py --gpus N` starts N ranks by itself")
    a = ap.parse_args()
    if a.workload is None:
        a.workload = "R1" if a.entry else "R2"
    return a


def plan_calls(n_views, H, W, C, esize, chunk=0, call_gb=66.0, min_calls=2, pool=32):
    """(views per call, number of calls, resident maps) for a rank that projects ``n_views`` views: --chunk, or as many views
    as hold --call-gb of feature maps; at least --min-calls calls; then the views are spread evenly over the calls; the pool
    holds a whole number of calls' worth of maps (at least one call's)."""
    per_call = chunk if chunk > 0 else max(1, int(round(call_gb * 1e9 / (H * W * C * esize))))
    per_call = max(1, min(per_call, n_views))
    n_calls = max(-(-n_views // per_call), min(min_calls, n_views))
    per_call = -(-n_views // n_calls)
    resident = max(per_call, (min(max(pool, per_call), n_views) // per_call) * per_call)
    return per_call, n_calls, resident


def device_info(dev):
    """Which GPU a line was measured on: the speed level of a run follows the box as much as the allocation."""
    p = torch.cuda.get_device_properties(dev)
    return {"name": p.name, "arch": getattr(p, "gcnArchName", None), "cus": p.multi_processor_count,
            "uuid": str(getattr(p, "uuid", "")) or None, "hbm_gb": round(p.total_memory / 1e9, 1)}


def _barrier(dist, dev):
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)


def _max_over_ranks(dist, dt, dev):
    if dist is None:
        return dt
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


R4_PMC_PROFILE = "r06_r4_pmc.json"            # tools/r4_round.sh


def r4_roofline(ach, algo, call_ms, r4pmc, lines):
    """The R4 line's roofline object.  SURVEY 8d waives the HBM claim for config 5 (3-12 B per voxel-view) and the counters say
    what bounds the kernel instead: the L1s' miss queues -- every 4-byte sample pulls a 64-byte line, ~57 line misses are in
    flight per L1 over the whole launch at ~870 cycles each, an L1 sits stalled behind lines already on their way half of the
    launch (profiles/r05_ab_colour_order.log section 8).  So `bound` says that, `peak` is what those queues sustained in the
    committed counter pass of THIS build -- misses in flight x 64 B / miss latency x 256 L1s, the clock taken from that pass's
    cycles over this run's launch time -- and `frac` = algorithmic bytes per second against it (below 1 by the misses beyond the
    distinct lines and by the launch's other kernels).  The HBM figures stay beside it as hbm_frac / hbm_peak."""
    s_ = (r4pmc or {}).get("summary") or {}
    peak = None
    if s_.get("l1_misses_in_flight_per_l1") and s_.get("l1_miss_latency_cycles") and s_.get("launch_cycles"):
        ghz = s_["launch_cycles"] / (call_ms * 1e-3) / 1e9
        peak = s_["l1_misses_in_flight_per_l1"] * 64.0 / (s_["l1_miss_latency_cycles"] / ghz) * 256.0      # GB/s (bytes per ns)
    return {"bound": "l1_miss_queue", "kernel": "vp_project_colors (k_color_cells + k_project_colors)", "achieved": round(ach, 1),
            "peak": (round(peak, 1) if peak else None), "unit": "GB/s", "frac": (round(ach / peak, 4) if peak else None),
            "peak_source": (f"profiles/{R4_PMC_PROFILE}: l1_misses_in_flight_per_l1 x 64 B / l1_miss_latency_cycles x 256 L1s at the clock "
                            "that pass's cycles give this run's launch (Little's law on TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum / "
                            "GRBM_GUI_ACTIVE) -- a measured ceiling of this build, not a datasheet number; null when the committed pass is "
                            "of other kernel sources"),
            "hbm_peak": HBM_PEAK_GBS, "hbm_frac": round(ach / HBM_PEAK_GBS, 5),
            "traffic": (int((r4pmc["FETCH_SIZE_KB_per_launch"] + r4pmc["WRITE_SIZE_KB_per_launch"]) * 1024) if r4pmc else None),
            "traffic_source": f"profiles/{R4_PMC_PROFILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this build, taken as read: "
                              "the x2 correction of gfx950 applies to 16-B-per-lane streaming reads, these are 4-byte gathers, and "
                              "FETCH_SIZE here equals TCC_MISS x 64 B) -- not measured in this run",
            "bytes_per_launch": int(algo), "avg_launch_ms": round(call_ms, 4), "counters": (s_ or None),
            "distinct_image_lines_per_launch": lines,
            "note": "one 4-byte load per voxel and view pulls a 64-byte line; with the voxels summed in Morton-curve order neighbouring "
                    "lanes share lines (L2 misses 101 M -> 56 M per 1000 views, 1.03x the distinct lines) and the launch went from 2.0 to "
                    "1.4-1.5 ms; fewer VALU instructions, deeper load rings and the cache-policy bits changed nothing or made it slower "
                    "(profiles/r05_ab_colour_order.log).  SURVEY 8d: GB/s and voxel-views/s reported, no HBM roofline claim"}


def bench_colors(a, dev, rank, world, dist):
    """BASELINE config 5 (R4): the RGB path -- 500 000 voxels x 1000 views, uint8 [1168,1752,3] images, voxel-driven nearest
    pixel, no occlusion (debug_project_colors.py:54-81 + aggregate_voxel_colors_onthefly.py:134-140).  One STEP = all 1000
    views through vp_project_colors in calls of --chunk views (default: all of a rank's views in one call; 1000 images are
    6.1 GB).  Three bytes are
    gathered per voxel-view, so the kernel is bound by scattered line fetches: the line carries the roofline object the
    contract asks for, with the honest fraction, and makes no roofline claim."""
    import voxproj_host
    from synthetic_scene import make_scene
    from view_sharding import reduce_partials, views_of_rank
    N, V, W, H = 500000, a.views or 1000, 1752, 1168
    s = make_scene(N, V, W, H, seed=0)
    my_views = views_of_rank(V, rank, world)
    chunk = max(1, min(a.chunk if a.chunk > 0 else 1000, len(my_views)))     # default: the rank's views in one call (6.1 GB of images)
    pool = chunk
    occ = torch.from_numpy(s.occ).to(dev)
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    imgs = torch.randint(0, 256, (pool, H, W, 3), dtype=torch.uint8, device=dev, generator=gen)
    c2w = torch.from_numpy(s.c2w).to(dev)
    intr = torch.from_numpy(s.intr)[None].repeat(V, 1).to(dev).contiguous()
    csum = torch.zeros(N + 1, 3, device=dev)
    hits = torch.zeros(N + 1, dtype=torch.int32, device=dev)
    first = torch.full((N + 1,), 2 ** 30, dtype=torch.int32, device=dev)
    origin = [float(v) for v in s.grid_origin]
    calls = [my_views[i:i + chunk] for i in range(0, len(my_views), chunk)]
    c2ws = [c2w[vs].contiguous() for vs in calls]
    intrs = [intr[vs].contiguous() for vs in calls]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in calls]

    def step(timed=False):
        csum.zero_(); hits.zero_()
        for ci, vs in enumerate(calls):
            if timed:
                ev[ci][0].record()
            voxproj_host.project_colors_raw(occ, c2ws[ci], intrs[ci], origin, s.voxel_size, imgs[:len(vs)], csum, hits,
                                            first_view=first, view_base=vs[0])
            if timed:
                ev[ci][1].record()
        if dist is not None:
            reduce_partials(dist, [csum, hits], dst=0 if a.collective == "reduce" else None)

    for _ in range(max(1, a.warmup)):
        step()
    _barrier(dist, dev)
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(timed=(k == a.steps - 1))
    _barrier(dist, dev)
    dt = _max_over_ranks(dist, time.perf_counter() - t0, dev)
    seen = torch.tensor([int(hits.sum().item()) if (dist is None or rank == 0 or a.collective == "allreduce") else 0], device=dev)
    call_ms = sum(e0.elapsed_time(e1) for e0, e1 in ev) / len(ev)
    if rank == 0:
        n_seen = int(seen.item())
        per_call_hits = n_seen / max(1, len(calls) * world)
        # Algorithmic bytes of one call: every DISTINCT 64-byte line of an image that holds a sampled pixel has to come in once
        # (images are not shared between views), counted exactly by a float64 torch restatement of the projection on the
        # device (distinct_image_lines); plus the two passes over the dense grid that build the voxel list, 12 B of list and
        # cell table + the read-modify-write of {3 floats, count, first view} per voxel, pose + intrinsics per view.
        # (Rounds 1-4 counted 3 B per sample: 0.03 "of peak".  The first half of round 5 counted min(64 B per sample, image
        # bytes) because the lane = voxel-ID kernel's L2 misses happened to equal the images' line count; with the voxels
        # in curve order the kernel FETCHES 3.4 GB of the 6.1 GB, so that was no floor.)
        if a.no_line_count:      # counter passes: a thousand views of torch kernels would swamp the profiler's tables
            lines = min(per_call_hits, chunk * H * W * 3 / 64.0)
        else:
            lines = distinct_image_lines(s, my_views, H, W, dev) / len(calls)          # per call, like call_ms
        img_bytes = lines * 64
        algo = img_bytes + 2 * occ.numel() * 4 + (N + 1) * (12 + 2 * 20) + chunk * 80
        ach = algo / (call_ms * 1e-3) / 1e9
        r4pmc = None
        try:
            with open(os.path.join(ROOT, "profiles", R4_PMC_PROFILE)) as f:
                r4pmc = json.load(f)
            if r4pmc.get("source_digest") != source_digest() or r4pmc.get("views_per_call") != chunk:
                r4pmc = None
        except OSError:
            pass
        res = {"metric": "Mvoxel-views/sec", "value": round(N * V / (dt / a.steps) / 1e6, 1), "unit": "Mvoxel-views/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64 projection, f32 colour sums",
               "data": "synthetic",
               "config": {"workload": f"R4: {N} voxels x {V} views x {W}x{H}x3 uint8 images, DPC semantics (nearest pixel, no "
                                      f"occlusion), room-shell scene seed 0", "views_per_call": chunk, "resident_images": pool,
                          "parallelism": f"views r::{world} per GPU + one RCCL {a.collective} of colour sums/counts per pass" if world > 1 else "single GPU"},
               "voxel_view_hits_per_step": n_seen,
               "roofline": r4_roofline(ach, algo, call_ms, r4pmc, None if a.no_line_count else int(lines))}
        if not a.no_cpu_baseline and world == 1:
            from oracle import oracle
            nv = min(400, V)                                           # ~10 s on one core
            img_np = imgs[0].cpu().numpy()
            t1 = time.perf_counter()
            for v in range(nv):
                oracle.rgb_project(s.occ, s.c2w[v], s.intr, s.grid_origin, s.voxel_size, img_np)
            dtc = time.perf_counter() - t1
            res["cpu_baseline"] = dict(value=round(N * nv / dtc / 1e6, 3), unit="Mvoxel-views/s", cores=1, kind="port",
                                       sample=f"{nv} of the workload's views, all {N} voxels, {dtc:.1f} s wall "
                                              "(oracle_rgb_project: scalar C restatement of debug_project_colors.py:54-81)")
            res["cpu_torch_loop"] = cpu_colour_loop(s, img_np, min(400, V), host_cores())
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_entry(a, dev, rank, world, dist):
    """The named entry point measured in-process: one STEP = every view of the workload through
    VoxelFeatureAggregator.add_views (aggregate_voxel_features_onthefly.py) with the feature maps resident, then flush().
    parity: one view per projector call + the fp16 per-view accumulate of the reference (AGG:307-313); fast: 8 views per
    pipelined call, fp32 sums.  Reported beside it, same run and data: the drop-in module called the way the reference
    calls it (one view per blocking project_features_cuda call, debug_project_features.py:201-208)."""
    import voxproj_host
    from aggregate_voxel_features_onthefly import VoxelFeatureAggregator
    from synthetic_scene import make_features_torch, make_scene
    from view_sharding import views_of_rank
    n_vox, n_views, W, H, C = WORKLOADS[a.workload]
    if a.views:
        n_views = a.views
    s = workload_scene(a.workload, n_views)
    my_views = views_of_rank(n_views, rank, world)
    pool = min(a.pool if a.pool != 32 else 128, len(my_views))
    feats = torch.empty((pool, H, W, C), dtype=torch.float32, device=dev)
    make_features_torch(pool, H, W, C, dev, seed=0, out=feats)
    c2w = torch.from_numpy(s.c2w).to(dev)
    intr4 = torch.from_numpy(s.intr)
    agg = VoxelFeatureAggregator(torch.from_numpy(s.occ), s.grid_origin.astype(np.float64), s.voxel_size, C, a.entry, dev,
                                 parity_pipeline=not a.entry_no_pipeline)
    per_call = 1 if a.entry == "parity" else max(1, min(8, a.chunk or 8))
    calls = [my_views[i:i + per_call] for i in range(0, len(my_views), per_call)]
    # the poses of every call, on the device and READY before the first call (a pipelined call's side stream reads them
    # without waiting for the caller's stream: an index kernel still pending there would be a contract violation)
    c2ws = [c2w[vs].contiguous() for vs in calls]
    torch.cuda.synchronize(dev)

    def step():
        agg.reset()
        for ci, vs in enumerate(calls):
            slot = vs[0] % pool if (vs[0] % pool) + len(vs) <= pool else 0
            agg.add_views(feats[slot:slot + len(vs)], c2ws[ci], intr4)
        if dist is not None:
            agg.all_reduce(dst=0 if a.collective == "reduce" else None)
        else:
            agg.flush()

    # untimed pre-pass through the raw C-ABI: algorithmic bytes of the dominant kernel for exactly these calls
    occ64 = torch.from_numpy(s.occ[None].astype(np.int64)).to(dev)
    cnt = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
    out = torch.zeros(n_vox + 1, C, device=dev)
    intr_d = torch.from_numpy(s.intr[None]).to(dev)
    ws0 = voxproj_host.Workspace()
    algo = 0
    for ci, vs in enumerate(calls):
        slot = vs[0] % pool if (vs[0] % pool) + len(vs) <= pool else 0
        cnt.zero_()
        voxproj_host.project_features_raw(feats[slot:slot + len(vs)][None], occ64, c2w[vs].reshape(-1).contiguous(), intr_d,
                                          [float(v) for v in s.opts()], cnt, out, [float(v) for v in s.grid_origin],
                                          s.voxel_size, workspace=ws0, sync=True, reuse_accel=(ci > 0 or None))
        ph, nt = int(cnt.sum().item()), int((cnt > 0).sum().item())
        algo += ph * C * 4 + nt * C * 4 * 2 + len(vs) * H * W * 4 + (n_vox + 1) * 4 * 2
    ws0.release()
    del ws0, out
    step()                                                     # builds the aggregator's tables, sizes its workspace
    for _ in range(a.warmup):
        step()
    _barrier(dist, dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    _barrier(dist, dev)
    dt = _max_over_ranks(dist, time.perf_counter() - t0, dev)
    # per-kernel HIP events in a pass of their own, after the timed region: with one view per call the event records (two
    # per kernel group) are a measurable share of a call, so they stay out of the number that is reported
    voxproj_host.profile_enable(True)
    for _ in range(a.steps):
        step()
    _barrier(dist, dev)
    prof = voxproj_host.profile_read()
    voxproj_host.profile_enable(False)
    if rank == 0:
        r = agg.result()
        n_out = int(r["xyz"].shape[0])
        # the drop-in module, one view per blocking call, same maps (compiled front; DPF:141-208 call pattern)
        import project_features_cuda as dropin
        out = torch.zeros(n_vox + 1, C, device=dev)
        opts = torch.from_numpy(s.opts()); org = torch.from_numpy(s.grid_origin); pm = torch.tensor([False])
        nv = min(32, len(my_views))
        vm = [c2w[v].reshape(-1).contiguous() for v in my_views[:nv]]
        best = None
        for rep in range(3):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for k in range(nv):
                dropin.project_features_cuda(feats[k % pool][None, None], occ64, vm[k], intr_d, opts, cnt, out, pm, org, s.voxel_size)
            d1 = (time.perf_counter() - t1) / nv
            best = d1 if best is None else min(best, d1)
        ms_view = dt / a.steps / len(my_views) * 1e3
        launches = max(prof["gather_launches"], 1)
        gather_ms = prof["gather_ms"] / launches
        res = {"metric": "Mvoxel-views/sec", "value": round(n_vox * n_views / (dt / a.steps) / 1e6, 3), "unit": "Mvoxel-views/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{a.workload} through the entry point's aggregator, --mode {a.entry}: {n_vox} voxels x {n_views} "
                                      f"views x {W}x{H}x{C} fp32 feature maps resident, {SCENE_NAME.get(a.workload, 'room-shell scene seed 0')}",
                          "views_per_call": per_call, "resident_feature_maps": pool,
                          "parallelism": f"views r::{world} per GPU + RCCL {a.collective}" if world > 1 else "single GPU"},
               "entry": {"mode": a.entry, "ms_per_view": round(ms_view, 4), "rows_out": n_out,
                         "dropin_ms_per_view": round(best * 1e3, 4), "entry_over_dropin": round(ms_view / (best * 1e3), 3),
                         "dropin_what": "project_features_cuda (compiled module), one view per blocking call"},
               "phase_ms_per_step": {"prep": round(prof["prep_ms"] / a.steps, 3), "first_hit": round(prof["first_hit_ms"] / a.steps, 3),
                                     "gather": round(prof["gather_ms"] / a.steps, 3), "combine_parts": round(prof["heavy_ms"] / a.steps, 3),
                                     "note": "HIP events of an extra pass after the timed region"},
               "roofline": {"bound": "hbm", "kernel": "k_gather", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                            "avg_launch_ms": round(gather_ms, 4)}}
        ach = algo / len(calls) / (gather_ms * 1e-3) / 1e9 if gather_ms > 0 else 0.0
        res["roofline"].update(achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4), bytes_per_launch=int(algo / len(calls)))
        if not a.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(s, C, min(a.cpu_views, n_views), host_cores())
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--write-r4-pmc":
        print(json.dumps(write_r4_pmc_json(sys.argv[2], sys.argv[3])))
        return
    if len(sys.argv) == 4 and sys.argv[1] == "--write-pmc-json":
        print(json.dumps(write_pmc_json(sys.argv[2], sys.argv[3]))[:300])
        return
    a = parse()
    global SCENE_SEED
    SCENE_SEED = a.scene_seed
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        return launch_ranks(a)                   # the parent: no GPU call before, none after
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: one rank per GPU, started by torchrun or by "
                         f"`python bench.py --gpus {a.gpus}` itself")
    if a.launch_check:
        return launch_check(a, rank, world, WORKLOADS, plan_calls)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the projector has no CPU fallback")
    if not a.single_device and local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs cuda:{local}, {torch.cuda.device_count()} GPU(s) visible")
    dev = torch.device("cuda", 0 if a.single_device else local)
    torch.cuda.set_device(dev)
    dist = None
    if world == 1 and a.rehearse_dist:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if world > 1 or a.rehearse_dist:
        import torch.distributed as dist
        if a.dist_backend == "nccl":
            # RCCL's kernels on a high-priority stream: its hardware queue is then never the one the projector's gather sits
            # on, so a collective started under a gather really runs under it
            dist.init_process_group("nccl", device_id=dev, pg_options=dist.ProcessGroupNCCL.Options(is_high_priority_stream=True))
        else:
            dist.init_process_group(a.dist_backend)
        assert dist.get_world_size() == world
        # communicator set-up is lazy: pay it here, not in the first timed step (the driver may pass --warmup 0)
        _t = torch.zeros(1, device=dev)
        dist.all_reduce(_t)
        torch.cuda.synchronize(dev)

    import voxproj_host
    from synthetic_scene import make_features_torch, make_scene
    if a.one_view_gather >= 0:
        voxproj_host.set_default_option(voxproj_host.VP_OPT_ONE_VIEW_GATHER, a.one_view_gather)
        try:
            import project_features_cuda as _m
            _m.set_workspace_option(voxproj_host.VP_OPT_ONE_VIEW_GATHER, a.one_view_gather)
        except ImportError:
            pass

    if a.workload == "R4":
        return bench_colors(a, dev, rank, world, dist)
    if a.entry:
        return bench_entry(a, dev, rank, world, dist)
    n_vox, n_views, W, H, C = WORKLOADS[a.workload]
    if a.views:
        n_views = a.views
    scene = workload_scene(a.workload, n_views)
    from view_sharding import reduce_partials, views_of_rank
    my_views = views_of_rank(n_views, rank, world)
    esize = 4 if a.dtype == "f32" else 2
    # views per call: --chunk, or as many as hold --call-gb of maps; then the rank's views are spread evenly over its calls
    min_calls = a.min_calls if a.min_calls is not None else (1 if dist is not None else 2)
    chunk, n_calls, pool = plan_calls(len(my_views), H, W, C, esize, a.chunk, a.call_gb, min_calls, a.pool)

    alloc_kind = {}

    def resident(shape, dtype, what):
        if a.alloc == "contiguous":
            t, kind = voxproj_host.resident_empty(shape, dtype, dev)
        else:
            t, kind = torch.empty(shape, dtype=dtype, device=dev), "default"
        alloc_kind[what] = kind
        return t

    # which synthetic map sits in which pool slot: when the pool holds ALL of the rank's views, slot j holds the map of the
    # rank's j-th view (seeded by its global view index), so a sharded run reads exactly the maps a single process reads
    # for the same scene; a cycled pool (fewer maps than views) holds maps 0 .. pool-1
    map_ids = list(my_views[:pool]) if pool >= len(my_views) else list(range(pool))

    def alloc_pool():
        if a.dtype == "f32":
            f = resident((1, pool, H, W, C), torch.float32, "feature_pool")
            make_features_torch(pool, H, W, C, dev, seed=0, out=f[0], view_ids=map_ids)
        else:
            f = resident((1, pool, H, W, C), torch.float16, "feature_pool")
            for v in range(pool):
                f[0, v] = make_features_torch(1, H, W, C, dev, seed=0, view_ids=[map_ids[v]])[0].half()
        return f

    occ = torch.from_numpy(scene.occ[None].astype(np.int64)).to(dev)
    c2w = torch.from_numpy(scene.c2w).to(dev)
    intr = torch.from_numpy(scene.intr[None]).to(dev)
    n_rows = n_vox + 1
    count = out = feats = None       # allocated by the placement loop below
    opts = [float(v) for v in scene.opts()]
    origin = [float(v) for v in scene.grid_origin]
    ws = voxproj_host.Workspace()
    if a.heavy_threshold > 0:
        ws.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, a.heavy_threshold)
    if a.march_lds_kb >= 0:
        ws.set_option(voxproj_host.VP_OPT_MARCH_LDS_KB, a.march_lds_kb)
    if a.part_pixels > 0:
        ws.set_option(voxproj_host.VP_OPT_PART_PIXELS, a.part_pixels)

    # calls of one step: (pool slot of the first view, view indices)
    calls = []
    for i in range(0, len(my_views), chunk):
        vs = my_views[i:i + chunk]
        slot = (i % pool) if (i % pool) + len(vs) <= pool else 0
        calls.append((slot, vs))
    vmis = [c2w[vs].reshape(-1).contiguous() for _, vs in calls]

    pipeline = not a.no_pipeline

    # Multi-rank: the step goes through the entry point's own aggregator (aggregate_voxel_features_onthefly.py, fast mode) --
    # add_views for every call of the rank but the last, add_final_views for the last call and the scene's collective.  What
    # the bench times at N > 1 is the code that ships, not a copy of it.
    agg = None
    if dist is not None:
        if not pipeline:
            raise SystemExit("bench.py: --no-pipeline is a single-process arm; the multi-rank step runs the entry point's "
                             "aggregator, whose fast mode is always pipelined")
        from aggregate_voxel_features_onthefly import VoxelFeatureAggregator
        from view_sharding import split_point
        agg = VoxelFeatureAggregator(torch.from_numpy(scene.occ), scene.grid_origin.astype(np.float64), scene.voxel_size, C, "fast", dev)
        assert agg.n_rows == n_rows and np.array_equal(np.asarray(agg._opts(W, H), np.float32), np.asarray(opts, np.float32))
        if a.heavy_threshold > 0:
            agg.ws.set_option(voxproj_host.VP_OPT_HEAVY_THRESHOLD, a.heavy_threshold)
        if a.march_lds_kb >= 0:
            agg.ws.set_option(voxproj_host.VP_OPT_MARCH_LDS_KB, a.march_lds_kb)
        if a.part_pixels > 0:
            agg.ws.set_option(voxproj_host.VP_OPT_PART_PIXELS, a.part_pixels)
        intr4 = intr.reshape(4)                                     # device tensors: the caller's promise that they are ready
        c2ws = [c2w[vs].contiguous() for _, vs in calls]
        torch.cuda.synchronize(dev)

    def one_call(ci, sync=False, o=None, c=None, gather_only=False):
        slot, vs = calls[ci]
        voxproj_host.project_features_raw(feats[:, slot:slot + len(vs)], occ, vmis[ci], intr, opts,
                                          count if c is None else c, out if o is None else o,
                                          origin, scene.voxel_size, workspace=ws, sync=sync,
                                          reuse_accel=(ci > 0 or None), pipeline=(pipeline and not sync), gather_only=gather_only)

    # Placement of the resident buffers (untimed set-up).  Where the driver puts the physical pages of the feature
    # pool and of the output rows moves the gather's speed by several per cent from one allocation to the next
    # (DESIGN.md section 4), stable for the life of the allocation.  --pool-tries N > 1 is an opt-in search: allocate, time one
    # whole pass, park the allocation (so that the next one lands elsewhere) and try again; the fastest placement is kept,
    # the others are freed.  Every try is reported.  (Multi-rank: the output rows are the aggregator's own, one try.)
    placement = {"tries": [], "picked": 0}
    parked, best = [], None
    for t in range(max(1, a.pool_tries if agg is None else 1)):
        f_try = alloc_pool()
        c_try = resident((n_rows,), torch.int32, "hit_counts").zero_() if agg is None else agg.count
        o_try = resident((n_rows, C), torch.float32, "output_rows").zero_() if agg is None else agg.sum32
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
    placement["allocation"] = dict(alloc_kind)
    feats, count, out = best[2], best[3], best[4]
    del parked, best, f_try, c_try, o_try
    torch.cuda.empty_cache()

    # Multi-GPU.  A step never overlaps passes: zero, project the rank's views, the collective, waited for -- a scene is ONE
    # pass and pays its collective inside it.  Two arms (view_sharding.project_final_call_and_reduce):
    #   split  the rank's LAST call is cut into two row ranges (the library gathers voxel IDs [0, h) first, then [h, n_rows)
    #          from the same first-hit images); the rows below h, final after the first gather, are reduced on RCCL's stream
    #          while the second gather runs.  Same bytes on the links, about half of them hidden; the cut itself costs a
    #          second, shorter gather launch.
    #   whole  the last call as it is, then one collective per tensor.
    # The timed region runs the default arm (split; --no-split-collective: whole), the other arm runs after it for the same
    # number of steps, and the line carries both (collective.arms).  collective_ms_exposed = pass - (zeroing + projection),
    # the latter from HIP events around it.
    state = {"exposed_s": 0.0, "proj_s": 0.0}
    dst = 0 if a.collective == "reduce" else None
    h_rows = split_point(n_rows) if dist is not None else 0
    arm_arg = "whole" if a.no_split_collective else a.collective_arm
    default_split = dist is not None and arm_arg != "whole" and h_rows > 0       # "auto": decided by the calibration below
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def step_dist(split_arm):
        t_a = time.perf_counter()
        ev_a.record()
        agg.reset()
        last = len(calls) - 1
        for ci in range(last):
            slot, vs = calls[ci]
            agg.add_views(feats[0, slot:slot + len(vs)], c2ws[ci], intr4)
        slot, vs = calls[last]
        agg.add_final_views(feats[0, slot:slot + len(vs)], c2ws[last], intr4, dst=dst, split=split_arm, on_projected=ev_b.record)
        torch.cuda.synchronize(dev)
        proj = ev_a.elapsed_time(ev_b) * 1e-3
        state["proj_s"] += proj
        state["exposed_s"] += (time.perf_counter() - t_a) - proj

    def step():
        if dist is not None:
            return step_dist(default_split)
        count.zero_()
        out.zero_()
        for ci in range(len(calls)):
            one_call(ci)

    # untimed pre-pass: algorithmic bytes of the dominant kernel per launch (deterministic across steps)
    hit_px, touched, gather_bytes, cnt, max_px, heavy_px = 0, 0, 0, {}, 0, 0
    per_call = []
    out.zero_()
    for ci in range(len(calls)):
        count.zero_()
        # alone on the device, HIP events around its kernels: what each call's march and gather do by themselves
        voxproj_host.profile_enable(True)
        one_call(ci, sync=True)
        pc = voxproj_host.profile_read()
        voxproj_host.profile_enable(False)
        ph, nt = int(count.sum().item()), int((count > 0).sum().item())
        hit_px += ph
        touched += nt
        # voxels above the library's per-call threshold (min(256 + 64*B*V, 2048) pixels) are summed in parts by wavefronts of the same
        # k_gather launch: every hit pixel's row and every touched output row count for this kernel
        c1 = voxproj_host.counters(ws, dev)
        heavy = count > c1["heavy_t"]                  # the threshold in force: --heavy-threshold or min(256 + 64 * views per call, 2048)
        heavy_px += int(count[heavy].sum().item())
        bytes_call = ph * C * esize + nt * C * 4 * 2 + len(calls[ci][1]) * H * W * 4 + n_rows * 4 * 2
        gather_bytes += bytes_call
        for k in c1:
            cnt[k] = max(cnt.get(k, 0), c1[k]) if k == "heavy_t" else cnt.get(k, 0) + c1[k]
        max_px = max(max_px, int(count.max().item()))
        per_call.append({"views": len(calls[ci][1]), "hit_pixels": ph, "miss": round(1.0 - ph / float(len(calls[ci][1]) * H * W), 3),
                         "touched": nt, "heavy": c1["n_heavy"], "parts": c1["n_parts"], "max_pixels_per_voxel": int(count.max().item()),
                         "bytes": bytes_call})
        if pc["gather_launches"] == 1:
            per_call[-1].update(gather_ms_alone=round(pc["gather_ms"], 3), march_ms_alone=round(pc["first_hit_ms"], 3),
                                frac_alone=round(bytes_call / (pc["gather_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))

    ref_checksum = out.double().sum(0)       # plain (unpipelined) result of one full pass on this rank
    ref_abs = out.double().abs().sum(0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if dist is not None:
        # one untimed collective at the real size (RCCL sizes its channels and staging on first use), whatever --warmup says
        scratch = [torch.zeros_like(out), torch.zeros_like(count)]
        reduce_partials(dist, scratch, dst=dst)
        torch.cuda.synchronize(dev)
        del scratch
    calibration = None
    if dist is not None and h_rows > 0 and arm_arg == "auto":
        # Which arm is faster depends on the links, the backend and the rank count (gloo rehearsals on one GPU: split 160 vs whole
        # 171 ms at 2 ranks, 2117 vs 353 ms at 4) and no 8-GPU node was ever available to measure it: the timed arm is chosen by a
        # short untimed calibration of both (tools/bench_calibrate.py: settle each arm until two consecutive steps agree, then the
        # minimum of three alternating steps per arm; a step far above its arm's minimum -> no pick, the split arm).  Every step is
        # bracketed like the timed region and MAX-reduced over the ranks, so every rank picks the same arm.  Both arms still
        # run and are reported.
        def timed_step(arm):
            barrier()
            t_c = time.perf_counter()
            step_dist(arm == "split")
            barrier()
            return _max_over_ranks(dist, time.perf_counter() - t_c, dev)

        calibration = choose_arm(timed_step)
        default_split = calibration["pick"] == "split"
    for _ in range(a.warmup):
        step()
    barrier()
    voxproj_host.profile_enable(True)
    state["exposed_s"] = 0.0
    state["proj_s"] = 0.0
    # (an event at the head of every step and one behind the last: per-step device time for `step_ms` without touching the timed
    # region -- the stream is never waited on between steps)
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        step_ev[i].record()
        step()
    step_ev[a.steps].record()
    barrier()
    dt = time.perf_counter() - t0
    step_list = sorted(step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(a.steps))
    voxproj_host.workspace_status(ws, dev)
    prof = voxproj_host.profile_read()
    voxproj_host.profile_enable(False)
    exposed_ms = state["exposed_s"] / max(1, a.steps) * 1e3
    proj_ms = state["proj_s"] / max(1, a.steps) * 1e3
    reduced = {}      # checksums of the scene for the line (compared across runs by tests/test_gpu_bench_contract.py)
    verify = not os.environ.get("VOXPROJ_BENCH_NOVERIFY")
    if dist is None and verify:
        # the timed passes must have produced the same result as the plain pre-pass
        assert int(count.sum().item()) == hit_px, "hit-count total changed between the pre-pass and the timed steps"
        assert ((out.double().sum(0) - ref_checksum).abs() <= 1e-6 * ref_abs + 1e-9).all(), "feature sums changed"
    if verify:
        last = voxproj_host.counters(ws if agg is None else agg.ws, dev)
        assert last["bad_id"] == 0 and last["box_miss"] == 0, last
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
        reduced = {"checksum": float(ref_checksum.sum().item()), "checksum_abs": float(ref_abs.sum().item())}

    ref_abs_all = 0.0

    def verify_reduced():
        # the aggregator's buffers hold the whole scene (on rank 0 after a reduce, everywhere after an all-reduce): the hit
        # counts must add up EXACTLY to the ranks' own (pre-pass) totals, and the feature sums, channel by channel, to the sum
        # of the ranks' pre-pass checksums (float64) within fp32 summation rounding
        t = torch.tensor([hit_px], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        chk = torch.stack([ref_checksum, ref_abs])
        dist.all_reduce(chk)
        nonlocal ref_abs_all
        ref_abs_all = float(chk[1].sum().item())
        if rank == 0 or a.collective == "allreduce":
            assert int(agg.count.sum().item()) == int(t.item()), "reduced hit counts do not add up to the ranks' totals"
            assert ((agg.sum32.double().sum(0) - chk[0]).abs() <= 1e-6 * chk[1] + 1e-9).all(), \
                "reduced feature sums differ from the sum of the ranks' single-rank results"
            assert agg.n_seen == n_views
            return {"reduced_hit_pixels": int(agg.count.sum().item()), "reduced_checksum": float(agg.sum32.double().sum().item())}
        return {}

    arms = None
    if dist is not None:
        if verify:
            reduced = verify_reduced()
        # per-rank projection and exposed-collective time of the timed arm: min and max over the ranks show a straggler
        per_rank = torch.zeros(world, 2, dtype=torch.float64, device=dev)
        per_rank[rank] = torch.tensor([proj_ms, exposed_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(per_rank)
        t = torch.tensor([dt, exposed_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, exposed_ms = float(t[0].item()), float(t[1].item())
        name = {True: "split", False: "whole"}
        arms = {name[default_split]: {"ms_per_step": round(dt / a.steps * 1e3, 3), "collective_ms_exposed": round(exposed_ms, 3)}}
        if not a.no_other_arm and h_rows > 0:
            # the other arm, same number of steps, bracketed the same way (never the headline)
            state["exposed_s"] = 0.0
            barrier()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                step_dist(not default_split)
            barrier()
            t = torch.tensor([time.perf_counter() - t1, state["exposed_s"] / max(1, a.steps) * 1e3], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            arms[name[not default_split]] = {"ms_per_step": round(float(t[0].item()) / a.steps * 1e3, 3),
                                             "collective_ms_exposed": round(float(t[1].item()), 3)}
            if verify:
                # each rank's own part is the same bits in both arms; the cross-rank sum may round differently (a collective
                # over half the rows is chunked differently by the backend's ring), so: counts exactly, sums to fp32 rounding
                other = verify_reduced()
                if reduced:
                    assert other["reduced_hit_pixels"] == reduced["reduced_hit_pixels"], "the two collective arms left different counts"
                    assert abs(other["reduced_checksum"] - reduced["reduced_checksum"]) <= 1e-6 * float(ref_abs_all), \
                        "the two collective arms left different scenes"
    split = default_split

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
        if split:
            launches = a.steps * len(calls)      # the split call's two launches count as the one call whose bytes they move
        gather_ms = prof["gather_ms"] / launches
        ach = (gather_bytes / len(calls)) / (gather_ms * 1e-3) / 1e9 if gather_ms > 0 else 0.0
        # whole-path algorithmic bytes per step: feature rows + output RMW + counts + ID image w+r
        algo_step = algo_local      # summed over ranks
        res = {
            "metric": "Mvoxel-views/sec", "value": round(value, 3), "unit": "Mvoxel-views/s",
            "n_gpus": dist.get_world_size() if dist is not None else 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if a.dtype == "f32" else "f32 accumulate, f16 feature maps",
            "data": "synthetic",
            "config": {"workload": f"{a.workload}: {n_vox} voxels x {n_views} views x {W}x{H}x{C} {'fp32' if a.dtype == 'f32' else 'fp16'} feature maps, "
                                   f"{SCENE_NAME.get(a.workload, 'room-shell scene seed 0').replace('seed 0', 'seed ' + str(a.scene_seed))}, dmin 0.01 dmax 10 step 0.5*voxel",
                       "views_per_call": chunk, "resident_feature_maps": pool,
                       "parallelism": (f"views r::{world} per GPU + one RCCL {'reduce to rank 0' if a.collective == 'reduce' else 'all-reduce'} of sum/count "
                                       f"per pass, waited for inside the pass (backend {a.dist_backend})")
                       if world > 1 else "single GPU (one-rank process group: rehearsal of the multi-rank path)" if dist is not None
                       else "single GPU"},
            "achieved_hbm_gbs_whole_path": round(algo_step / (dt / a.steps) / 1e9, 1),
            "step_ms": {"min": round(step_list[0], 3), "median": round(step_list[len(step_list) // 2], 3), "max": round(step_list[-1], 3)},
            "phase_ms_per_step": {"prep": round(prof["prep_ms"] / a.steps, 3),
                                  "first_hit": round(prof["first_hit_ms"] / a.steps, 3),
                                  "gather": round(prof["gather_ms"] / a.steps, 3),
                                  "combine_parts": round(prof["heavy_ms"] / a.steps, 3),
                                  "overlapped": pipeline},
            "pool_placement": placement, "device": device_info(dev), **reduced,
            **({"collective": {"op": "reduce to rank 0" if a.collective == "reduce" else "all-reduce", "backend": a.dist_backend,
                               "through": "VoxelFeatureAggregator.add_views / add_final_views (aggregate_voxel_features_onthefly.py, fast mode)",
                               "bytes_per_rank": n_rows * C * 4 + 2 * n_rows * 4,
                               "collective_ms_exposed": round(exposed_ms, 3),
                               "projection_ms_per_step": round(ms_step - exposed_ms, 3),
                               "timed_arm": "split" if split else "whole", "arms": arms,
                               "timed_arm_chosen_by": (calibration["chosen_by"] if calibration is not None else "--collective-arm " + arm_arg),
                               "calibration": calibration,
                               "collectives_per_pass": ("feature sums in two pieces + one int32 tensor {pixel counts, view counts, views seen}"
                                                        if split else "feature sums + one int32 tensor {pixel counts, view counts, views seen}"),
                               "per_rank": {"projection_ms": {"min": round(float(per_rank[:, 0].min().item()), 3), "max": round(float(per_rank[:, 0].max().item()), 3)},
                                            "collective_ms_exposed": {"min": round(float(per_rank[:, 1].min().item()), 3), "max": round(float(per_rank[:, 1].max().item()), 3)}},
                               "split": ({"rows_reduced_under_the_last_gather": h_rows, "of": n_rows} if split else None),
                               "note": "max over ranks; the headline value includes the collective (no overlap between passes); "
                                       "collective_ms_exposed = pass - (zeroing + projection), the latter from HIP events.  split: the "
                                       "rank's last call is cut into two row ranges and the first half's sums are reduced under the "
                                       "second half's gather; whole: the call as it is, then the collectives.  The arm that is not "
                                       "timed runs after the timed region, same number of steps"}} if dist is not None else {}),
            "hit_pixels_per_step": hit_px, "miss_fraction": round(1.0 - hit_px / float(len(my_views) * H * W), 4),
            "march_share_of_device_time": round(prof["first_hit_ms"] / max(1e-9, prof["first_hit_ms"] + prof["gather_ms"] + prof["heavy_ms"]), 4),
            "box_miss_voxels": cnt["box_miss"], "heavy_voxels_per_step": cnt["n_heavy"], "heavy_pixels_per_step": heavy_px, "heavy_threshold": cnt["heavy_t"],
            "parts_per_step": cnt["n_parts"], "max_pixels_per_voxel_call": max_px, "per_call": per_call,
            "roofline": {"bound": "hbm", "kernel": "k_gather", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "measured_stream_read_gbs": round(stream_gbs, 1),
                         "frac_of_measured_stream_read": round(ach / stream_gbs, 4) if stream_gbs > 0 else None,
                         # one-word diagnostic of the placement level this run landed on: the gather
                         # against the SAME box's plain streaming read -- fast >= 0.98, mid >= 0.93, else slow
                         "level": (None if stream_gbs <= 0 else "fast" if ach / stream_gbs >= 0.98 else "mid" if ach / stream_gbs >= 0.93 else "slow"),
                         "traffic": (int(pmc_traffic(a.workload, chunk, a.dtype) * len(my_views) / len(calls))
                                     if (not a.views and world == 1 and pmc_traffic(a.workload, chunk, a.dtype)) else None),
                         "traffic_source": f"profiles/{PMC_PROFILE} (rocprofv3 --pmc passes of this build, rescaled to this "
                                           "run's views per launch; null when the kernel sources changed since) -- not measured in this run",
                         "bytes_per_launch": gather_bytes // len(calls), "avg_launch_ms": round(gather_ms, 4)},
        }
        if not a.no_cpu_baseline and world == 1:      # reported on rank 0 at N=1 only (bench contract)
            ncores = host_cores()
            res["cpu_baseline"] = cpu_baseline(scene, C, min(a.cpu_views, n_views), ncores)
            res["cpu_baseline_1t"] = cpu_baseline(scene, C, max(1, min(a.cpu_views // 8, n_views)), 1)      # the oracle single-threaded too
            res["cpu_torch_loop"] = cpu_torch_loop(scene, min(300, n_views), ncores)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
