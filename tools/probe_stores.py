"""Do the output-row stores cost what they cost because they are RANDOM rows, or because they are stores at all?

    python3 tools/probe_stores.py [--allocs 4]

tools/probe_rows.hip's random whole-row gather (the gather's access shape without the projector), 63 k wavefronts of 270
rows each -- one wavefront per voxel of a 32-view R2 call -- on --allocs copies of a 35 GB pool, ending (0) without a store,
(1) with one 2-KiB row stored to a RANDOM row of a 410 MB buffer (the gather's output rows), (2) with the row stored to the
wave's own slot of a compact buffer (a staging area).  Prints ms per launch per copy and per destination buffer.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--allocs", type=int, default=4)
ap.add_argument("--dsts", type=int, default=3)
ap.add_argument("--dst-flags", default="", help="comma list of hipExtMallocWithFlags flags for extra destination buffers "
                "(1 = fine-grained, 3 = uncached): is the store cost a property of the memory type?")
a = ap.parse_args()
dev = torch.device("cuda", 0)
L = ctypes.CDLL(os.path.join(ROOT, "tools", "libprobe_rows.so"))
L.probe_rows_store.restype = ctypes.c_int
L.probe_rows_store.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                               ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
ROWS = 32 * 548 * 968          # 2-KiB rows of a 32-map pool
pools = [torch.empty(ROWS * 512, dtype=torch.float32, device=dev).normal_() for _ in range(1)]
for _ in range(1, a.allocs):
    p = torch.empty_like(pools[0]); p.copy_(pools[0]); pools.append(p)
dsts = [torch.zeros(200001 * 512, dtype=torch.float32, device=dev) for _ in range(a.dsts)]
if a.dst_flags:
    sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
    import voxproj_host
    for fl in a.dst_flags.split(","):
        t, kind = voxproj_host.resident_empty((200001 * 512,), torch.float32, dev, fallback=False, flags=int(fl))
        t.zero_()
        dsts.append(t)
        print(f"destination buffer {len(dsts) - 1}: hipExtMallocWithFlags flags {fl}")
sink = torch.zeros(4, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
WAVES, ITERS = 63000, 272


def run(pool, dst, mode):
    best = None
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = L.probe_rows_store(pool.data_ptr(), 0, ROWS, 2048, WAVES, ITERS, 99 + rep, sink.data_ptr(), dst.data_ptr(), 200001, mode, stream)
        assert rc == 0
        e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1)
        if rep:
            best = t if best is None else min(best, t)
    return best


print("ms per launch (63 k wavefronts x 272 random 2-KiB rows = 35 GB read); columns: destination buffers")
for k, pool in enumerate(pools):
    base = run(pool, dsts[0], 0)
    rnd = [run(pool, d, 1) for d in dsts]
    seq = [run(pool, d, 2) for d in dsts]
    print(f"pool {k}: no store {base:.3f} | random-row store " + " ".join(f"{t:.3f}" for t in rnd) + " | own-slot store " + " ".join(f"{t:.3f}" for t in seq), flush=True)
    names = {3: "store half-way", 4: "non-temporal store", 5: "1 KiB only", 6: "two rows", 7: "row read first (RMW)",
             8: "sc1", 9: "sc0 sc1", 10: "sc0 sc1 nt"}
    print("         " + " | ".join(f"{names[m]} {run(pool, dsts[0], m):.3f}" for m in sorted(names)), flush=True)
