"""Generates tests/golden/ptdc_reference_golden.npz by RUNNING the reference's own prepare_tensor_data_color.py in the build
container (numpy / torch / PIL only, PTDC:22-27 -- it runs here as it is; the script cannot travel to the GPU box, its
outputs can).

Inputs are synthetic: fp16 LSeg-style maps [C,h,w], a COLMAP-style camera JSON with one 4-parameter and one 3-parameter
camera (PTDC:113-118), random rotations R and translations tvec, a small occupancy tensor, random RGB images.  The reference
looks for the RGB images in a directory hard-coded at PTDC:70 and lists it unconditionally (PTDC:92), so the directory must
exist: the script is run UNMODIFIED inside a private mount namespace (`unshare -m`, a tmpfs over /home that only that
process sees; nothing outside this repository is written).  Two runs:

    noimg  the directory is empty  -> "no original image" branch (PTDC:109-112): maps as stored, image = zeros (PTDC:144)
    img    the images are there    -> maps up-sampled to the image size with F.interpolate(bilinear, align_corners=False)
                                      (PTDC:98-108), image = the LAST view's pixels (PTDC:99,144)

What the fixture pins, bit for bit, in both:

    viewMatrixInv   c2w = [R^T | -R^T t] in float32 (PTDC:121-126)      intrinsicParams  fx fy cx cy (PTDC:113-120)
    encoded_2d_features  stack + permute to [1,V,H,W,C] float32 (PTDC:131-132)     image  (PTDC:144)
    file order = sorted(os.listdir)[:max_images] (PTDC:62-65), views without a camera entry skipped (PTDC:74-77)

The fixture also stores the inputs, so the test (tests/test_ptdc_golden_cpu.py) rebuilds the same files and runs this
package's prepare_tensor_data_color.main on them.

Usage (build container only):  python tests/golden/make_ptdc_golden.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/cuda_project_image_to_sparse_voxel"
REF_IMAGES_DIR = "/home/neural_fields/Unified-Lift-Gabor/data/scannetpp/officescene/images"      # PTDC:70
NAMES = ["DSC00010.JPG", "DSC00002.JPG", "DSC00031.JPG"]      # the middle one sorts first; the last has no camera entry


def random_rotation(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def make_inputs(seed=11):
    rng = np.random.default_rng(seed)
    C, h, w = 12, 9, 14
    feats = {n: rng.normal(size=(C, h, w)).astype(np.float16) for n in NAMES}
    R = {n: random_rotation(rng) for n in NAMES[:2]}
    tvec = {n: rng.normal(size=3) * 2.0 for n in NAMES[:2]}
    params = {"1": [623.966, 624.818, 876.0, 584.0], "2": [511.25, 437.5, 290.125]}
    occ = rng.integers(0, 5, size=(4, 5, 6)).astype(np.int32)
    imgs = {n: rng.integers(0, 256, size=(23, 37, 3), dtype=np.uint8) for n in NAMES[:2]}       # [H,W,3], not a multiple of (h,w)
    return dict(feats=feats, R=R, tvec=tvec, params=params, occ=occ, imgs=imgs, grid_origin=[0.25, -1.5, 0.125], voxel_size=0.05)


def write_images(images_dir, inp):
    """The views' RGB images under the very names PTDC:80-82 tries first (lossless PNG data, whatever the suffix says)."""
    from PIL import Image
    os.makedirs(images_dir, exist_ok=True)
    for n, a in inp["imgs"].items():
        Image.fromarray(np.asarray(a)).save(os.path.join(images_dir, n), format="PNG")


def write_inputs(tmp, inp):
    """Lays the synthetic inputs out as files; returns the argv shared by the reference script and this package's."""
    lseg = os.path.join(tmp, "lseg")
    os.makedirs(lseg)
    for n, a in inp["feats"].items():
        np.save(os.path.join(lseg, n + ".npy"), a)
    images = {str(i + 1): {"name": n, "camera_id": i + 1, "R": np.asarray(inp["R"][n]).tolist(), "tvec": np.asarray(inp["tvec"][n]).tolist()}
              for i, n in enumerate(NAMES[:2])}
    cams = {k: {"params": [float(x) for x in v]} for k, v in inp["params"].items()}
    with open(os.path.join(tmp, "cams.json"), "w") as f:
        json.dump({"images": images, "cameras": cams}, f)
    torch.save(torch.from_numpy(np.asarray(inp["occ"])), os.path.join(tmp, "occ.pt"))
    return ["--lseg_dir", lseg, "--scaled_camera_params", os.path.join(tmp, "cams.json"), "--occupancy", os.path.join(tmp, "occ.pt"),
            "--voxel_size", str(inp["voxel_size"]), "--grid_origin"] + [str(v) for v in inp["grid_origin"]] + \
           ["--max_images", "3", "--output", os.path.join(tmp, "tensor_data.pt")]


def main():
    inp = make_inputs()
    env = dict(os.environ, TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD="1")
    out = {}
    for run in ("noimg", "img"):
        with tempfile.TemporaryDirectory() as tmp:
            argv = write_inputs(tmp, inp)
            stage = os.path.join(tmp, "images")
            os.makedirs(stage)
            if run == "img":
                write_images(stage, inp)
            # private mount namespace: the hard-coded directory exists for this one process only
            sh = (f"mount -t tmpfs none /home && mkdir -p '{REF_IMAGES_DIR}' && cp -r '{stage}'/. '{REF_IMAGES_DIR}'/ && "
                  f"exec '{sys.executable}' '{os.path.join(REF, 'prepare_tensor_data_color.py')}' " + " ".join(f"'{v}'" for v in argv))
            subprocess.run(["unshare", "-m", "sh", "-c", sh], check=True, env=env, cwd=tmp, stdout=subprocess.DEVNULL)
            r = torch.load(os.path.join(tmp, "tensor_data.pt"), weights_only=False)
        out.update({f"ref_{run}_" + k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in r.items()})
    for n in NAMES[:2]:
        out["img_" + n] = inp["imgs"][n]
    for n in NAMES:
        out["feat_" + n] = inp["feats"][n]
    for n in NAMES[:2]:
        out["R_" + n] = inp["R"][n]
        out["tvec_" + n] = inp["tvec"][n]
    out["params_1"] = np.asarray(inp["params"]["1"])
    out["params_2"] = np.asarray(inp["params"]["2"])
    out.update(occ=inp["occ"], grid_origin=np.asarray(inp["grid_origin"]), voxel_size=np.float64(inp["voxel_size"]))
    path = os.path.join(HERE, "ptdc_reference_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape, str(v.dtype)) for k, v in out.items() if k.startswith("ref_")})


if __name__ == "__main__":
    main()
