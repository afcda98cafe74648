"""Latency of the drop-in call used the way the reference uses it (debug_project_features.py:141-208): ONE view per
blocking call, (a) the same occupancy tensor every call, (b) a fresh `.long()` copy per call as DPF:143 makes -- the library
then compares the grid with the copy its tables were built from (VP_FLAG_VERIFY_ACCEL).

    python tools/bench_dropin.py [--shape R2|R1|both] [--occ same|fresh|both] [--front compiled|python|both]
                                 [--views 16] [--reps 3] [--phases]

Per line: ms per call, Mvoxel-views/s, GB/s of ALGORITHMIC bytes of the call (hit pixels' rows + output-row read-modify-write
+ ID image write and read + counts: SURVEY 8d) and that as a fraction of the 8 TB/s HBM peak.  --phases adds the library's
own HIP-event times per kernel group (an extra pass; the events cost a few microseconds per call, so they stay out of the
timed loop)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import project_features_cuda as m  # noqa: E402  (the compiled extension)
import project_features_front as front  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_features_torch, make_scene  # noqa: E402

SHAPES = {"R2": (200000, 968, 548, 512), "R1": (80000, 484, 274, 512),
          # the hand-held trajectory legs of bench.py (round 5): --view-ids picks frames (0 = close-up dwell, ~130 = the look through the opening)
          "A1": (87319, 876, 584, 512), "R2T": (200000, 968, 548, 512)}


def _bench_module():
    import importlib.util
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
        bm = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bm)
    finally:
        sys.argv = argv
    return bm


def run(shape, occ_modes, fronts, NV, reps, phases, view_ids=None):
    dev = torch.device("cuda", 0)
    n_vox, W, H, C = SHAPES[shape]
    s = _bench_module().workload_scene(shape) if shape in ("A1", "R2T") else make_scene(n_vox, 300 if shape == "R2" else 100, W, H, seed=0)
    if view_ids:
        s.c2w = s.c2w[view_ids]
        NV = len(view_ids)
    feats = make_features_torch(NV, H, W, C, dev, seed=0)
    occ32 = torch.from_numpy(s.occ).to(dev)
    occ = occ32.unsqueeze(0).long().contiguous()
    c2w = torch.from_numpy(s.c2w).to(dev)
    vm = [c2w[v].reshape(-1).contiguous() for v in range(NV)]
    intr = torch.from_numpy(s.intr[None]).to(dev)
    opts = torch.from_numpy(s.opts())
    origin = torch.from_numpy(s.grid_origin)
    count = torch.zeros(n_vox + 1, dtype=torch.int32, device=dev)
    out = torch.zeros(n_vox + 1, C, device=dev)
    pm = torch.tensor([False])
    # algorithmic bytes of the NV calls (deterministic): one untimed pass
    algo = 0
    for v in range(NV):
        count.zero_()
        m.project_features_cuda(feats[v:v + 1].unsqueeze(0), occ, vm[v], intr, opts, count, out, pm, origin, s.voxel_size)
        ph, nt = int(count.sum().item()), int((count > 0).sum().item())
        algo += ph * C * 4 + nt * C * 4 * 2 + H * W * 4 * 2 + (n_vox + 1) * 4 * 2
    algo /= NV
    for fresh in occ_modes:
        for name in fronts:
            fn = m.project_features_cuda if name == "compiled" else front.project_features_cuda_py
            ts = []
            for rep in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for v in range(NV):
                    o = occ32.unsqueeze(0).long().contiguous() if fresh else occ
                    fn(feats[v:v + 1].unsqueeze(0), o, vm[v], intr, opts, count, out, pm, origin, s.voxel_size)
                ts.append((time.perf_counter() - t0) / NV)
            t = min(ts)
            line = (f"{shape}{' views ' + ','.join(map(str, view_ids)) if view_ids else ''} {'fresh occupancy tensor per call' if fresh else 'same occupancy tensor':32s} {name:9s} {t * 1e3:.4f} ms/call  "
                    f"{n_vox / t / 1e6:.0f} Mvoxel-views/s  {algo / t / 1e9:.0f} GB/s algorithmic = {algo / t / 8e12:.3f} of peak "
                    f"({W * H * C * 4 / t / 1e9:.0f} GB/s of feature-map bytes)")
            if phases:
                voxproj_host.profile_enable(True)
                for v in range(NV):
                    o = occ32.unsqueeze(0).long().contiguous() if fresh else occ
                    fn(feats[v:v + 1].unsqueeze(0), o, vm[v], intr, opts, count, out, pm, origin, s.voxel_size)
                p = voxproj_host.profile_read()
                voxproj_host.profile_enable(False)
                line += (f"  | per call, HIP events: prep {p['prep_ms'] / NV * 1e3:.1f} us, march+worklist {p['first_hit_ms'] / NV * 1e3:.1f} us, "
                         f"gather {p['gather_ms'] / NV * 1e3:.1f} us, combine {p['heavy_ms'] / NV * 1e3:.1f} us")
            print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="both", choices=("R2", "R1", "both", "A1", "R2T"))
    ap.add_argument("--view-ids", default="", help="comma-separated frames of the workload's camera path (default: its first --views)")
    ap.add_argument("--occ", default="both", choices=("same", "fresh", "both"))
    ap.add_argument("--front", default="both", choices=("compiled", "python", "both"))
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--phases", action="store_true")
    ap.add_argument("--one-view", type=int, default=-1, help="VP_OPT_ONE_VIEW_GATHER of both fronts' workspaces: 0 = the general "
                    "gather kernel (A/B arm), n > 0 = the one-view kernel with n workgroups per CU, -1 = the library's default")
    ap.add_argument("--one-view-split", type=int, default=-1, help="VP_OPT_ONE_VIEW_SPLIT of both fronts' workspaces: 0 = round 5's path "
                    "(a workgroup per voxel above 320 pixels, no parts), n > 0 = split voxels above n pixels, -1 = the library's default "
                    "(parts sized on the device from the view's hit total)")
    a = ap.parse_args()
    if a.one_view_split >= 0:
        m.set_workspace_option(voxproj_host.VP_OPT_ONE_VIEW_SPLIT, a.one_view_split)
        voxproj_host.set_default_option(voxproj_host.VP_OPT_ONE_VIEW_SPLIT, a.one_view_split)
    if a.one_view >= 0:
        m.set_workspace_option(voxproj_host.VP_OPT_ONE_VIEW_GATHER, a.one_view)
        voxproj_host.set_default_option(voxproj_host.VP_OPT_ONE_VIEW_GATHER, a.one_view)
    for shape in (("R2", "R1") if a.shape == "both" else (a.shape,)):
        run(shape, {"same": (False,), "fresh": (True,), "both": (False, True)}[a.occ],
            ("compiled", "python") if a.front == "both" else (a.front,), a.views, a.reps, a.phases,
            [int(v) for v in a.view_ids.split(",")] if a.view_ids else None)


if __name__ == "__main__":
    main()
