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

    def add_views(self, images_u8, c2w, intr):
        """images u8 [V,H,W,3], c2w f32 [V,4,4], intr f32 [V,4] (per-view intrinsics, DPC:64)."""
        V = int(images_u8.shape[0])
        voxproj_host.project_colors_raw(self.occ, c2w.to(self.dev, torch.float32).contiguous(),
                                        intr.to(self.dev, torch.float32).contiguous(), self.grid_origin,
                                        self.voxel_size, images_u8.to(self.dev).contiguous(), self.csum, self.hits,
                                        first_view=self.first_view, view_base=self.n_seen)
        self.n_seen += V

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
