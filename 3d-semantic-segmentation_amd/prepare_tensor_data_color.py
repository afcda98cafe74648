"""Per-view tensor packing for the colour pipeline (counterpart of the reference's prepare_tensor_data_color.py).

The reference script (cuda_project_image_to_sparse_voxel/prepare_tensor_data_color.py:28-160) is run once per view by
aggregate_voxel_colors_onthefly.py:103-115 and writes the same ``tensor_data.pt`` as prepare_tensor_data.py plus the key
``image``: the view's RGB image as a uint8 [H,W,3] array (PTDC:144), which is all debug_project_colors.py reads besides
the pose (DPC:24-52).  This module keeps the command line and the file's keys / shapes / dtypes:

  encoded_2d_features  f32 [1,V,H,W,C]  the LSeg map, bilinearly up-sampled to the IMAGE size when the image file is found
                                         (torch interpolate, align_corners=False: PTDC:101-105 -- the same library call
                                         here, so the values are the reference's), else as stored (PTDC:109-112)
  occupancy_3D         as loaded        (PTDC:41)
  intrinsicParams      f32 [1,V,4]      UNscaled fx fy cx cy (PTDC:113-120; one row per view here, unlike PTD's two)
  viewMatrixInv        f32 [1,V,4,4]    c2w = [R^T | -R^T t] (PTDC:121-126)
  grid_origin, voxel_size
  image                u8 [H,W,3]       the LAST view's image, or zeros of the feature map's size (PTDC:144)

The image directory is a hard-coded path in the reference (PTDC:70); here it is ``--images_dir`` (default: that path).
The colour aggregator in this package reads images itself and never writes this file; the script exists so that a user of
the reference's three-script colour pipeline finds the same middle step.
"""
import argparse
import os

import numpy as np
import torch

import prepare_tensor_data as ptd

IMAGES_DIR = "/home/neural_fields/Unified-Lift-Gabor/data/scannetpp/officescene/images"        # PTDC:70


def find_image(images_dir, base):
    """PTDC:80-96: the bare name, then common extensions, then a case-insensitive match of the stem."""
    cand = os.path.join(images_dir, base)
    if os.path.exists(cand):
        return cand
    for ext in (".jpg", ".jpeg", ".png", ".JPG", ".JPEG", ".PNG"):
        cand = os.path.join(images_dir, base + ext)
        if os.path.exists(cand):
            return cand
    if os.path.isdir(images_dir):
        for name in os.listdir(images_dir):
            if os.path.splitext(name)[0].lower() == base.lower():
                return os.path.join(images_dir, name)
    return None


def main(argv=None):
    from PIL import Image
    p = argparse.ArgumentParser()
    p.add_argument("--lseg_dir", required=True, help="Folder of .npy LSeg features")
    p.add_argument("--scaled_camera_params", required=True, help="Path to scaled camera params JSON")
    p.add_argument("--occupancy", required=True, help="Path to occupancy.pt")
    p.add_argument("--voxel_size", type=float, required=True, help="Voxel size")
    p.add_argument("--grid_origin", nargs=3, type=float, required=True, help="Grid origin (x y z)")
    p.add_argument("--max_images", type=int, default=1, help="Max images to use (should be 1 for color pipeline)")
    p.add_argument("--output", required=True, help="Output tensor_data.pt")
    p.add_argument("--images_dir", default=os.environ.get("IMAGES_DIR", IMAGES_DIR))
    args = p.parse_args(argv)

    occ = torch.load(args.occupancy)
    by_name, cams = ptd.load_camera_params(args.scaled_camera_params)
    files = sorted(f for f in os.listdir(args.lseg_dir) if f.endswith(".npy"))
    if args.max_images:
        files = files[:args.max_images]
    feats, intrs, exts = [], [], []
    image_array, last_hw = None, None
    for fname in files:
        base = fname[:-4]
        entry = by_name.get(base)
        if entry is None:
            print(f"[WARN] No camera entry for feature file: {fname} (expected name: {base}), skipping.")
            continue
        arr = np.load(os.path.join(args.lseg_dir, fname))
        img_path = find_image(args.images_dir, base)
        if img_path is not None:
            img = Image.open(img_path).convert("RGB")
            orig_w, orig_h = img.size
            image_array = np.array(img)                                                           # PTDC:99
            up = torch.nn.functional.interpolate(torch.from_numpy(arr).unsqueeze(0).float(), size=(orig_h, orig_w),
                                                 mode="bilinear", align_corners=False)           # PTDC:101-104
            arr = up.squeeze(0).cpu().numpy()
        else:
            print(f"[DEBUG] No original image found for {base}, using feature shape as is: {arr.shape}")
        last_hw = arr.shape[1:]
        feats.append(torch.from_numpy(arr).float())
        intr, c2w = ptd.camera_for(entry, cams, None)                                             # PTDC:113-126
        intrs.append(intr)
        exts.append(c2w)
    if not feats:
        raise RuntimeError("No valid feature/camera pairs found!")
    out = {
        "encoded_2d_features": torch.stack(feats, 0).unsqueeze(0).permute(0, 1, 3, 4, 2).contiguous(),   # PTDC:131-132
        "occupancy_3D": occ,
        "intrinsicParams": torch.stack(intrs, 0).unsqueeze(0),
        "viewMatrixInv": torch.stack(exts, 0).unsqueeze(0),
        "grid_origin": torch.tensor(args.grid_origin, dtype=torch.float32),
        "voxel_size": float(args.voxel_size),
        "image": image_array if image_array is not None else np.zeros((last_hw[0], last_hw[1], 3), dtype=np.uint8),   # PTDC:144
    }
    print(f"Saving tensor_data to: {args.output} (encoded_2d_features shape: {out['encoded_2d_features'].shape})")
    torch.save(out, args.output)
    print("Done.")


if __name__ == "__main__":
    main()
