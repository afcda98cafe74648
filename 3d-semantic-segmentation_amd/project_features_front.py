"""Python-side companions of the compiled drop-in module ``project_features_cuda``.

The drop-in itself is a COMPILED extension, like the reference's (cuda_project_image_to_sparse_voxel/setup.py:10-27,
project_image_cuda.cpp:78-79): ``setup.py build_ext --inplace`` here produces ``project_features_cuda.*.so`` from
csrc/project_features_ext.cpp, and ``import project_features_cuda`` fails -- as it does in the reference -- when it has
not been built.  This module adds what lives beside it:

  project_features_cuda_py   the same front written in Python over ctypes: the ten positional arguments, the in-place
                             ``+=`` semantics of the two output tensors (project_image_cuda_kernel.cu:77,88), the ``None``
                             return and the argument checks of project_image_cuda.cpp:38-61 (same messages, raised as
                             RuntimeError like TORCH_CHECK does), ending in the same C-ABI call (include/voxproj.h).  The
                             GPU tests run every case through both fronts; it is NOT a fallback that is picked silently.
  last_workspace             test/diagnostic hook: the scratch buffer of the last call of either front
  IMPLEMENTATION             "compiled" when the extension module is importable, else "python" (informational)

Differences from the reference, all fenced in DESIGN.md:
  * no stdout banners / device printf (project_image_cuda.cpp:35, kernel.cu:148-156,202-254);
  * device errors and out-of-range voxel IDs raise instead of being printed or corrupting memory;
  * ``pred_mode_t = True`` raises (the reference's branch reinterprets float32 data as int32,
    project_image_cuda.cpp:46 vs kernel.cu:445, and is never used by the pipeline);
  * V is not limited to 16 views per call and feature offsets are 64-bit.

There is no CPU fallback: without libvoxproj.so or without a GPU tensor every call raises.
"""
import torch

import voxproj_host as _host

try:
    import project_features_cuda as _ext      # the compiled module (needs torch imported first: libtorch, libc10_hip)
except ImportError:
    _ext = None

__all__ = ["project_features_cuda_py", "IMPLEMENTATION", "last_workspace"]


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check_input(t, name):
    # CHECK_INPUT = CHECK_CUDA + CHECK_CONTIGUOUS (project_image_cuda.cpp:5-7)
    _check(isinstance(t, torch.Tensor) and t.is_cuda, f"{name} must be a CUDA tensor")
    _check(t.is_contiguous(), f"{name} must be contiguous")


def project_features_cuda_py(encoded_2d_features, occupancy_3D, viewMatrixInv, intrinsicParams, opts,
                             mapping2dto3d_num, projected_features, pred_mode_t, grid_origin, voxel_size):
    """Projecting from 2D to 3D: accumulate per-voxel feature sums and pixel hit counts in place.

    Arguments exactly as project_image_cuda.cpp:23-33 / project_image_cuda_kernel.cu:374-385:
      encoded_2d_features f32 [B,V,H,W,C] cuda; occupancy_3D i64 [B,Z,Y,X] cuda (0 = empty, else ID);
      viewMatrixInv f32 1-D [B*V*16] cuda (row-major camera->world); intrinsicParams f32 [B,4] cuda;
      opts f32 1-D [5] = [W, H, depth_min, depth_max, ray_increment] (any device);
      mapping2dto3d_num i32 [>= max_id+1] cuda (in/out); projected_features f32 [>= max_id+1, C] cuda
      (in/out); pred_mode_t bool [1]; grid_origin f32 [3] CPU; voxel_size float.
    """
    # Device and contiguity checks (project_image_cuda.cpp:38-43)
    _check_input(encoded_2d_features, "encoded_2d_features")
    _check_input(occupancy_3D, "occupancy_3D")
    _check_input(viewMatrixInv, "viewMatrixInv")
    _check_input(intrinsicParams, "intrinsicParams")
    _check_input(mapping2dto3d_num, "mapping2dto3d_num")
    _check_input(projected_features, "projected_features")
    # Dtype checks (project_image_cuda.cpp:46-53)
    _check(encoded_2d_features.dtype == torch.float32, "encoded_2d_features must be float32")
    _check(occupancy_3D.dtype == torch.int64, "occupancy_3D must be int64")
    _check(viewMatrixInv.dtype == torch.float32, "viewMatrixInv must be float32")
    _check(intrinsicParams.dtype == torch.float32, "intrinsicParams must be float32")
    _check(isinstance(opts, torch.Tensor) and opts.dtype == torch.float32, "opts must be float32")
    _check(mapping2dto3d_num.dtype == torch.int32, "mapping2dto3d_num must be int32")
    _check(projected_features.dtype == torch.float32, "projected_features must be float32")
    _check(isinstance(pred_mode_t, torch.Tensor) and pred_mode_t.dtype == torch.bool, "pred_mode_t must be bool")
    # Shape checks (project_image_cuda.cpp:56-61)
    _check(encoded_2d_features.dim() == 5, "encoded_2d_features must be 5D [B,V,H,W,C]")
    _check(occupancy_3D.dim() == 4, "occupancy_3D must be 4D [B,Z,Y,X]")
    _check(viewMatrixInv.dim() == 1, "viewMatrixInv must be 1D flattened")
    _check(intrinsicParams.dim() == 2, "intrinsicParams must be 2D [B,4]")
    _check(opts.dim() == 1 and opts.numel() == 5, "opts must be 1D with 5 elements")
    _check(pred_mode_t.dim() == 1 and pred_mode_t.numel() == 1, "pred_mode_t must be scalar")

    # What the reference's launcher assumes without checking (kernel.cu:390-414); checked here so
    # that a bad call raises instead of reading or writing out of bounds.
    B, V, H, W, C = encoded_2d_features.shape
    _check(occupancy_3D.shape[0] == B, "occupancy_3D batch size must match encoded_2d_features")
    _check(viewMatrixInv.numel() == B * V * 16, "viewMatrixInv must hold B*V*16 floats")
    _check(intrinsicParams.shape[0] >= B and intrinsicParams.shape[1] == 4, "intrinsicParams must be [B,4]")
    _check(isinstance(grid_origin, torch.Tensor) and not grid_origin.is_cuda and grid_origin.dtype == torch.float32
           and grid_origin.dim() == 1 and grid_origin.numel() >= 3,
           "grid_origin must be a 1D float32 CPU tensor with 3 elements")
    _check(projected_features.dim() == 2 and projected_features.shape[1] == C,
           "projected_features must be [num_ids, C]")
    _check(mapping2dto3d_num.dim() == 1 and mapping2dto3d_num.shape[0] == projected_features.shape[0],
           "mapping2dto3d_num must be [num_ids] with num_ids = projected_features.size(0)")
    dev = encoded_2d_features.device
    for t, name in ((occupancy_3D, "occupancy_3D"), (viewMatrixInv, "viewMatrixInv"),
                    (intrinsicParams, "intrinsicParams"), (mapping2dto3d_num, "mapping2dto3d_num"),
                    (projected_features, "projected_features")):
        _check(t.device == dev, f"{name} must be on the same device as encoded_2d_features")

    pred_mode = bool(pred_mode_t.cpu()[0])                 # kernel.cu:427-429
    _check(not pred_mode, "pred_mode_t = True (integer label projection) is not supported: the reference "
                          "branch reads float32 storage as int32 and is unreachable in its pipeline")
    opts_cpu = opts.detach().to("cpu")                     # kernel.cu:400-401
    _host.project_features_raw(
        encoded_2d_features, occupancy_3D, viewMatrixInv, intrinsicParams,
        [float(v) for v in opts_cpu.tolist()], mapping2dto3d_num, projected_features,
        [float(v) for v in grid_origin[:3].tolist()], float(voxel_size), sync=True, verify_accel=True)
    return None


class _ExtWorkspace:
    """What voxproj_host.hit_image / counters need to know about the compiled front's scratch buffer."""

    def __init__(self, ptr, shape):
        self._ptr, self.last_shape = int(ptr), tuple(int(v) for v in shape)

    def ptr(self):
        return self._ptr


def last_workspace(device, front=None):
    """Test/diagnostic hook: the scratch buffer of the last call on ``device`` (for voxproj_host.hit_image and
    voxproj_host.counters).  ``front``: "compiled" / "python", default the one ``project_features_cuda`` is."""
    device = torch.device(device)
    if (front or IMPLEMENTATION) == "compiled":
        idx = device.index if device.index is not None else torch.cuda.current_device()
        return _ExtWorkspace(*_ext.last_call(idx))
    return _host.get_workspace(device)


IMPLEMENTATION = "compiled" if _ext is not None else "python"
