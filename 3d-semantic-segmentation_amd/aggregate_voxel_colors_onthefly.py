"""Scene-level RGB aggregation in one process (counterpart of aggregate_voxel_colors_onthefly.py).

The reference (cuda_project_image_to_sparse_voxel/aggregate_voxel_colors_onthefly.py:62-220) runs two
sub-processes per view and adds each voxel's colour into dicts keyed by (z,y,x): float32 colour sums
(AGGC:136-139), ``hit_count`` = number of views that saw the voxel (AGGC:140), mean = sum / count (AGGC:186),
rows in dict-insertion order (first view that saw the voxel, then raster (z,y,x) inside a view).  Here all
views of a batch go through ONE kernel launch (k_project_colors) that keeps exactly that arithmetic:
each voxel adds its views in view order.

Output keys (AGGC:213-218): xyz f32 [n,3], avg_color f32 [n,3], hit_count i64 [n], voxel_coords i32 [n,3].
"""
import numpy as np
import torch

import voxproj_host

_NEVER = 2 ** 30


class VoxelColorAggregator:
    def __init__(self, occ_zyx, grid_origin, voxel_size, device="cuda"):
        self.dev = torch.device(device)
        self.occ = occ_zyx.to(self.dev, torch.int32).contiguous()
        self.grid_origin = [float(v) for v in grid_origin]
        self.voxel_size = float(voxel_size)
        self.n_rows = int(self.occ.max().item()) + 1
        self.csum = torch.zeros(self.n_rows, 3, dtype=torch.float32, device=self.dev)
        self.hits = torch.zeros(self.n_rows, dtype=torch.int32, device=self.dev)
        self.first_view = torch.full((self.n_rows,), _NEVER, dtype=torch.int32, device=self.dev)
        self.n_seen = 0

    def add_views(self, images_u8, c2w, intr, want_uv=False):
        """images u8 [V,H,W,3], c2w f32 [V,4,4], intr f32 [V,4] (per-view intrinsics, DPC:64).  With ``want_uv``
        returns the sampled pixel per view and voxel ID, int32 [V,n_rows,2] ((-1,-1) where unseen)."""
        V = int(images_u8.shape[0])
        uv = torch.empty(V, self.n_rows, 2, dtype=torch.int32, device=self.dev) if want_uv else None
        voxproj_host.project_colors_raw(self.occ, c2w.to(self.dev, torch.float32).contiguous(),
                                        intr.to(self.dev, torch.float32).contiguous(), self.grid_origin,
                                        self.voxel_size, images_u8.to(self.dev).contiguous(), self.csum, self.hits,
                                        first_view=self.first_view, view_base=self.n_seen, pixel_uv=uv)
        self.n_seen += V
        return uv

    def result(self):
        zyx = (self.occ > 0).nonzero(as_tuple=False)                   # raster (z,y,x)
        ids = self.occ[zyx[:, 0], zyx[:, 1], zyx[:, 2]].long()
        keep = self.hits[ids] > 0
        zyx, ids = zyx[keep], ids[keep]
        order = torch.argsort(self.first_view[ids].long(), stable=True)   # insertion order of AGGC:136-137
        zyx, ids = zyx[order], ids[order]
        avg = self.csum[ids] / self.hits[ids].float()[:, None]             # AGGC:186 (float32 / int)
        z = zyx.cpu().numpy()
        xyz = (z[:, [2, 1, 0]].astype(np.int64) * self.voxel_size + np.array(self.grid_origin, dtype=np.float64))
        return dict(xyz=torch.from_numpy(xyz.astype(np.float32)), avg_color=avg.cpu(),
                    hit_count=self.hits[ids].long().cpu(), voxel_coords=torch.from_numpy(z.astype(np.int32)))


# ----------------------------------------------------------------------------------------------------------
# entry point (reference: the top-level script body of aggregate_voxel_colors_onthefly.py:13-220)
# ----------------------------------------------------------------------------------------------------------
CHECKPOINT_DIR = "voxel_color_checkpoints"          # AGGC:17
CHECKPOINT_EVERY = 50                               # AGGC:145


def main(argv=None):
    import argparse
    import glob
    import os

    from PIL import Image

    import build_sparse_occupancy as bso
    import prepare_tensor_data as ptd

    ap = argparse.ArgumentParser(description="Aggregate voxel colors pipeline")
    ap.add_argument("--first_only", action="store_true", help="Only process the first input image for debug")
    ap.add_argument("--lseg_dir", default=os.environ.get("LSEG_DIR", "lseg_embed_features/features"),
                    help="the reference enumerates the views through the feature files (AGGC:62)")
    ap.add_argument("--images_dir", default=os.environ.get("IMAGES_DIR", "images"))
    ap.add_argument("--cam_params", default=os.environ.get("CAM_PARAMS", "camera_params/camera_params.json"))
    ap.add_argument("--voxel_ply", default=os.environ.get("VOXEL_PLY", "minkowski_grid.ply"))
    ap.add_argument("--checkpoint_dir", default=os.environ.get("CHECKPOINT_DIR", CHECKPOINT_DIR))
    ap.add_argument("--views_per_call", type=int, default=32)
    args = ap.parse_args(argv)
    os.makedirs(args.checkpoint_dir, exist_ok=True)

    voxel_size, grid_origin, grid_shape, n_from_name = bso.extract_voxel_params(args.voxel_ply)      # AGGC:24-59
    num_voxels = n_from_name if n_from_name is not None else (int(np.prod(grid_shape)) if grid_shape else "unknown")
    feature_files = sorted(glob.glob(os.path.join(args.lseg_dir, "*.npy")))                          # AGGC:62
    if not feature_files:
        raise RuntimeError(f"No .npy feature files found in {args.lseg_dir}")
    if args.first_only:
        feature_files = feature_files[:1]
    occ = bso.build_occupancy(bso.read_voxel_ply(args.voxel_ply), grid_origin, voxel_size, device="cuda")
    by_name, cams = ptd.load_camera_params(args.cam_params)
    agg = VoxelColorAggregator(occ, grid_origin, voxel_size)

    def save(idx, final):
        r = agg.result()
        if r["xyz"].shape[0] == 0:
            return
        name = (f"ALL_nonzero_voxel_colors_{idx}_vox{num_voxels}.pt" if final else f"checkpoint_voxel_colors_{idx}.pt")
        torch.save(r, os.path.join(args.checkpoint_dir, name))                                       # AGGC:173-179,212-218

    imgs, c2ws, intrs, idx = [], [], [], 0

    def flush():
        if imgs:
            agg.add_views(torch.from_numpy(np.stack(imgs)), torch.stack(c2ws), torch.stack(intrs))
            imgs.clear(); c2ws.clear(); intrs.clear()

    for k, fpath in enumerate(feature_files):
        base = os.path.basename(fpath)[:-4]
        img_path = next((c for c in (os.path.join(args.images_dir, base + e) for e in ("", ".jpg", ".JPG", ".png", ".PNG"))
                         if os.path.exists(c)), None)                                                # AGGC:85-99
        entry = by_name.get(base)
        if img_path is None or entry is None:
            print(f"[WARN] No image file or camera entry found for {base}")
            continue
        img = np.array(Image.open(img_path).convert("RGB"))                                          # AGGC:100-101
        if imgs and img.shape != imgs[0].shape:
            flush()
        intr, c2w = ptd.camera_for(entry, cams, None)                                                # PTDC: unscaled intrinsics
        imgs.append(img); c2ws.append(c2w); intrs.append(intr)
        idx = k + 1
        if len(imgs) >= args.views_per_call or idx % CHECKPOINT_EVERY == 0:
            flush()
        if idx % CHECKPOINT_EVERY == 0:                                                              # AGGC:145
            save(idx, final=False)
    flush()
    save(idx, final=True)
    print(f"[DONE] COLOR PROJECTION PIPELINE COMPLETED. Checkpoint directory: {args.checkpoint_dir}")


if __name__ == "__main__":
    main()
