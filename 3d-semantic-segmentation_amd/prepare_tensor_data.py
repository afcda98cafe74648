"""Per-view tensor packing (counterpart of the reference's prepare_tensor_data.py).

The reference script (cuda_project_image_to_sparse_voxel/prepare_tensor_data.py:38-201) turns LSeg
feature files + COLMAP-style camera JSON into ``tensor_data.pt``.  This module keeps its command line
and the keys/shapes/dtypes of that file, and exposes the steps as functions so that the aggregator can
call them in-process:

  load_camera_params   JSON schema of PTD:56-72  ({"images": {id: {name, camera_id, R, tvec}}, "cameras":
                       {id: {params: [fx,fy,cx,cy] | [f,cx,cy]}}})
  camera_for           intrinsics x downsample factor (PTD:132-143) and c2w = [R^T | -R^T t] (PTD:165-172)
  upsample_features    fp16 [C,h,w] -> bilinear (OpenCV's INTER_LINEAR rule, PTD:119-127) -> cast back to the file's
                       dtype (PTD:126) -> float32 channels-last [H,W,C] (PTD:152,183-185)

The resize is the hand-written HIP up-sampler behind the C-ABI (vp_upsample_features, csrc/vp_prep.h: one transpose pass
and one wavefront per output pixel instead of C separate ``cv2.resize`` calls, a cast and a permute); it needs a GPU and
there is no CPU fallback for it.  cv2 is third-party and unpinned in the reference (cuda_requirement.txt:10); the
arithmetic is pinned to OpenCV's published rule, spelled out in csrc/vp_prep.h and restated in oracle/resize_oracle.py.
"""
import argparse
import json
import os

import numpy as np
import torch


def load_camera_params(path):
    with open(path, "r") as f:
        cam_params = json.load(f)
    imgs, cams = cam_params["images"], cam_params["cameras"]
    by_name = {}
    for _, v in (imgs.items() if isinstance(imgs, dict) else enumerate(imgs)):
        if isinstance(v, dict) and "name" in v:
            by_name[v["name"]] = v
    return by_name, cams


def camera_for(entry, cams, downsample_factor=None):
    """(intr float32 [4] = fx,fy,cx,cy, c2w float32 [4,4]) of one image entry."""
    params = cams[str(entry["camera_id"])]["params"]
    if len(params) == 4:
        fx, fy, cx, cy = params
    else:
        fx, cx, cy = params
        fy = fx
    if downsample_factor is not None:
        fx, fy, cx, cy = (v * downsample_factor for v in (fx, fy, cx, cy))
    intr = torch.tensor([fx, fy, cx, cy], dtype=torch.float32)
    R = np.array(entry["R"], dtype=np.float32)
    t = np.array(entry["tvec"], dtype=np.float32)
    c2w = np.eye(4, dtype=np.float32)
    c2w[:3, :3] = R.T
    c2w[:3, 3] = -R.T @ t
    return intr, torch.from_numpy(c2w)


def upsample_features(arr, size=None, device="cpu", keep_dtype=False, out=None):
    """[C,h,w] array (float16 or float32) -> [H,W,C] tensor on ``device``: float32 like the reference (PTD:152), or,
    with ``keep_dtype``, in the file's own dtype (fp16 for LSeg features) -- the values are the same because
    PTD:126 casts the resized map back to the file's dtype before widening it.  A change of size runs on the GPU
    (``device`` must be a CUDA device then; the result stays there unless ``device`` says "cpu" -- no: it raises)."""
    if isinstance(arr, torch.Tensor) and arr.is_cuda:              # already on the device (the aggregator's feeder)
        import voxproj_host
        t = arr if arr.dtype in (torch.float16, torch.float32) else arr.float()
        C, h, w = t.shape
        H, W = (int(v) for v in size) if size is not None else (h, w)
        return voxproj_host.upsample_features(t, H, W, keep_dtype=keep_dtype, out=out)
    arr = np.ascontiguousarray(arr)
    if arr.dtype not in (np.float16, np.float32):
        arr = arr.astype(np.float32)
    dev = torch.device(device)
    C, h, w = arr.shape
    H, W = (int(v) for v in size) if size is not None else (h, w)
    if dev.type != "cuda":
        if (H, W) != (h, w):
            raise RuntimeError("upsample_features: resizing runs on the GPU (vp_upsample_features); pass a CUDA device")
        t = torch.from_numpy(arr)
        if not (keep_dtype and t.dtype == torch.float16):
            t = t.float()
        return t.permute(1, 2, 0).contiguous()
    import voxproj_host
    return voxproj_host.upsample_features(torch.from_numpy(arr).to(dev), H, W, keep_dtype=keep_dtype, out=out)


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--lseg_dir", required=True)
    p.add_argument("--scaled_camera_params", required=True)
    p.add_argument("--occupancy", required=True)
    p.add_argument("--voxel_size", type=float, required=True)
    p.add_argument("--grid_origin", nargs=3, type=float, required=True)
    p.add_argument("--max_images", type=int, default=10)
    p.add_argument("--output", required=True)
    p.add_argument("--image_size", nargs=2, type=int)
    p.add_argument("--downsample_factor", type=float, default=None)
    p.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu",
                   help="where the feature maps are resized (a resize needs a GPU)")
    args = p.parse_args(argv)

    occ = torch.load(args.occupancy)
    by_name, cams = load_camera_params(args.scaled_camera_params)
    files = sorted(f for f in os.listdir(args.lseg_dir) if f.endswith(".npy"))
    if args.max_images:
        files = files[:args.max_images]
    feats, intrs, exts = [], [], []
    for fname in files:
        entry = by_name.get(fname[:-4])
        if entry is None:
            print(f"[WARN] No camera entry for feature file: {fname}, skipping.")
            continue
        arr = np.load(os.path.join(args.lseg_dir, fname))
        feats.append(upsample_features(arr, tuple(args.image_size) if args.image_size else None, device=args.device).cpu())
        intr, c2w = camera_for(entry, cams, args.downsample_factor)
        intrs.append(intr)                          # PTD:143 (scaled) or PTD:152 (no factor)
        intrs.append(camera_for(entry, cams, None)[0])   # PTD:162 appends the unscaled row again, always
        exts.append(c2w)
    if not feats:
        raise RuntimeError("No valid feature/camera pairs found!")
    out = {
        "encoded_2d_features": torch.stack(feats, 0).unsqueeze(0),           # [1,V,H,W,C]
        "occupancy_3D": occ,
        "intrinsicParams": torch.stack(intrs, 0).unsqueeze(0),
        "viewMatrixInv": torch.stack(exts, 0).unsqueeze(0),
        "grid_origin": torch.tensor(args.grid_origin, dtype=torch.float32),
        "voxel_size": float(args.voxel_size),
    }
    torch.save(out, args.output)
    print(f"Saved tensor_data to: {args.output}")


if __name__ == "__main__":
    main()
