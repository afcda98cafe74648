"""A/B of builds of libvoxproj.so on the RGB path (config 5, R4: 500 000 voxels x 1000 views x 1752x1168x3 uint8), all arms on
the same image allocation:   python tools/probe_colors.py libA.so libB.so [...] [--views 1000] [--rounds 3]
Prints the mean time of one vp_project_colors call over all views and checks that every arm leaves the same bits."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-semantic-segmentation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import voxproj_host  # noqa: E402
from synthetic_scene import make_scene  # noqa: E402


def arg(name, default):
    for i, a in enumerate(sys.argv):
        if a == name:
            return sys.argv[i + 1]
    return default


libs = [os.path.abspath(a) for a in sys.argv[1:] if a.endswith(".so")]
V, rounds = int(arg("--views", "1000")), int(arg("--rounds", "3"))
N, W, H = 500000, 1752, 1168
dev = torch.device("cuda", 0)
s = make_scene(N, V, W, H, seed=0)
order = arg("--order", "scan")
if order != "scan":      # relabel the occupied cells (what a sorted ID list inside the library would give): "morton" = full Z-order of
    # the cells; "blockN" = N x N x N blocks in scan order of the blocks (x fastest), Z-order inside a block
    occ_np = s.occ
    zz, yy, xx = np.nonzero(occ_np)

    def spread(v):
        v = v.astype(np.uint64)
        out = np.zeros_like(v)
        for b in range(12):
            out |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return out

    def morton(x, y, z):
        return spread(x) | (spread(y) << np.uint64(1)) | (spread(z) << np.uint64(2))
    if order == "morton":
        key = morton(xx, yy, zz)
    else:
        n = int(order[5:])
        dz, dy, dx = occ_np.shape
        nbx, nby = -(-dx // n), -(-dy // n)
        blk = ((zz // n).astype(np.uint64) * np.uint64(nby) + (yy // n).astype(np.uint64)) * np.uint64(nbx) + (xx // n).astype(np.uint64)
        key = (blk << np.uint64(36)) | morton(xx % n, yy % n, zz % n)
    rank = np.empty(len(key), np.int32)
    rank[np.argsort(key, kind="stable")] = np.arange(1, len(key) + 1, dtype=np.int32)
    occ_np = occ_np.copy()
    occ_np[zz, yy, xx] = rank
    s.occ = occ_np
print("# grid", s.occ.shape, "voxel", s.voxel_size)
occ = torch.from_numpy(s.occ).to(dev)
gen = torch.Generator(device=dev); gen.manual_seed(0)
imgs = torch.randint(0, 256, (V, H, W, 3), dtype=torch.uint8, device=dev, generator=gen)
c2w = torch.from_numpy(s.c2w).to(dev)
intr = torch.from_numpy(s.intr)[None].repeat(V, 1).to(dev).contiguous()
origin = [float(v) for v in s.grid_origin]
res, ref = {}, None
for rnd in range(rounds):
    for path in libs:
        voxproj_host._lib = None
        voxproj_host.LIB_PATH = path
        csum = torch.zeros(N + 1, 3, device=dev)
        hits = torch.zeros(N + 1, dtype=torch.int32, device=dev)
        first = torch.full((N + 1,), 2 ** 30, dtype=torch.int32, device=dev)
        for rep in range(3):
            csum.zero_(); hits.zero_()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            voxproj_host.project_colors_raw(occ, c2w, intr, origin, s.voxel_size, imgs, csum, hits, first_view=first, view_base=0)
            torch.cuda.synchronize(dev)
            if rep > 0:
                res.setdefault(path, []).append((time.perf_counter() - t0) * 1e3)
        chk = (csum.cpu().numpy().tobytes(), hits.cpu().numpy().tobytes(), first.cpu().numpy().tobytes())
        if ref is None:
            ref = chk
        assert chk == ref, f"{path}: colour sums / counts / first views differ from the first arm's"
print(f"# ID order: {order}")
print(f"# R4: {N} voxels x {V} views, one blocking vp_project_colors call (k_color_cells + k_project_colors + a status read-back), "
      f"{rounds} rounds x 2 timed calls per arm, same bits from every arm")
for path in libs:
    t = np.array(res[path])
    print(f"{os.path.basename(path):28s} {t.mean():7.3f} ms per call (min {t.min():7.3f})  -> {N * V / (t.mean() * 1e-3) / 1e9:7.1f} Gvoxel-views/s")
