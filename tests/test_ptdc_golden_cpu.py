"""prepare_tensor_data_color.py (and, through the shared camera code, prepare_tensor_data.camera_for) against outputs of the
REFERENCE's own prepare_tensor_data_color.py, run unmodified in the build container (tests/golden/make_ptdc_golden.py ->
tests/golden/ptdc_reference_golden.npz: inputs and expected outputs; the reference script itself never travels).

Bit for bit: c2w = [R^T | -R^T t] in float32 (PTDC:121-126), fx fy cx cy incl. the 3-parameter camera (PTDC:113-120),
the [1,V,H,W,C] float32 packing (PTDC:131-132), F.interpolate to the image size (PTDC:101-105), the `image` key (PTDC:99,144),
the sorted file order and the skipped view without a camera entry (PTDC:62-77)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_ptdc_golden as gen  # noqa: E402  (input layout shared with the generator; it does not touch the reference on import)

GOLD = os.path.join(HERE, "golden", "ptdc_reference_golden.npz")


def _inputs(g):
    return dict(feats={n: g["feat_" + n] for n in gen.NAMES}, R={n: g["R_" + n] for n in gen.NAMES[:2]},
                tvec={n: g["tvec_" + n] for n in gen.NAMES[:2]}, params={"1": g["params_1"].tolist(), "2": g["params_2"].tolist()},
                occ=g["occ"], imgs={n: g["img_" + n] for n in gen.NAMES[:2]}, grid_origin=g["grid_origin"].tolist(),
                voxel_size=float(g["voxel_size"]))


@pytest.mark.parametrize("run", ["noimg", "img"])
def test_color_packer_reproduces_the_reference_run_bit_for_bit(tmp_path, run):
    import prepare_tensor_data_color as ptdc
    g = np.load(GOLD)
    inp = _inputs(g)
    argv = gen.write_inputs(str(tmp_path), inp)
    images = tmp_path / "images"
    images.mkdir()
    if run == "img":
        gen.write_images(str(images), inp)
    ptdc.main(argv + ["--images_dir", str(images)])
    d = torch.load(tmp_path / "tensor_data.pt", weights_only=False)
    assert set(d) == {"encoded_2d_features", "occupancy_3D", "intrinsicParams", "viewMatrixInv", "grid_origin", "voxel_size", "image"}
    for k, v in d.items():
        ref = g[f"ref_{run}_{k}"]
        got = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
        assert got.dtype == ref.dtype and got.shape == ref.shape, (k, got.dtype, ref.dtype, got.shape, ref.shape)
        assert np.array_equal(got, ref), f"{k} differs from the reference's output ({run})"
    # what the fixture exercises: two views in sorted-name order, the third file has no camera entry and is skipped
    assert d["encoded_2d_features"].shape[1] == 2 and d["intrinsicParams"][0, 0, 1] == d["intrinsicParams"][0, 0, 0]   # 3-parameter camera first
    assert (d["image"].shape[:2] == (23, 37) and d["image"].any()) if run == "img" else not d["image"].any()


def test_feature_packer_shares_the_pinned_camera_code():
    """prepare_tensor_data.camera_for (PTD:132-143,165-172) on the fixture's JSON entries gives the reference's c2w and
    intrinsics -- the half of a10 that does not depend on cv2."""
    import prepare_tensor_data as ptd
    g = np.load(GOLD)
    order = sorted(gen.NAMES[:2])                       # the packer's view order
    cams = {"1": {"params": g["params_1"].tolist()}, "2": {"params": g["params_2"].tolist()}}
    for v, n in enumerate(order):
        entry = {"name": n, "camera_id": gen.NAMES.index(n) + 1, "R": g["R_" + n].tolist(), "tvec": g["tvec_" + n].tolist()}
        intr, c2w = ptd.camera_for(entry, cams, None)
        assert np.array_equal(c2w.numpy(), g["ref_img_viewMatrixInv"][0, v]) and c2w.dtype == torch.float32
        assert np.array_equal(intr.numpy(), g["ref_img_intrinsicParams"][0, v])
        # with a down-sample factor the intrinsics scale in Python floats before the float32 cast (PTD:132-143)
        intr2, c2w2 = ptd.camera_for(entry, cams, 0.5)
        p = cams[str(entry["camera_id"])]["params"]
        fx, fy, cx, cy = p if len(p) == 4 else (p[0], p[0], p[1], p[2])
        assert intr2.tolist() == torch.tensor([fx * 0.5, fy * 0.5, cx * 0.5, cy * 0.5], dtype=torch.float32).tolist()
        assert torch.equal(c2w2, c2w)
