"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/voxproj.h declares, and the extension-module mirror validates like the reference wrapper
(project_image_cuda.cpp:38-61).  No compute call is made here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "voxproj.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vp_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import voxproj_host
    voxproj_host.build()
    lib = ctypes.CDLL(voxproj_host.LIB_PATH)
    names = _declared_functions()
    assert "vp_project_features" in names and len(names) >= 6
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(voxproj_host.EXPORTS) == names
    lib.vp_abi_version.restype = ctypes.c_int
    assert lib.vp_abi_version() == 4


def test_workspace_record_calls_validate_their_arguments_on_the_host():
    """vp_workspace_create / _set_option / _release keep host-side records only: their argument checks run without a GPU."""
    import voxproj_host
    lib = voxproj_host.lib()
    vp = ctypes.c_void_p
    assert lib.vp_workspace_create(vp(None), 1 << 20) == -1                         # VP_EINVAL: null workspace
    assert b"null" in lib.vp_last_error()
    assert lib.vp_workspace_create(vp(0x1000_0010), 1 << 20) == -2                  # VP_EWORKSPACE: not 256-byte aligned
    assert lib.vp_workspace_create(vp(0x1000_0000), 100) == -2                      # too small for the header
    assert lib.vp_workspace_set_option(vp(None), voxproj_host.VP_OPT_HEAVY_THRESHOLD, 5) == -1
    fake = vp(0x7000_0000_0000)                                                     # never dereferenced by these calls
    assert lib.vp_workspace_create(fake, 1 << 20) == 0
    assert lib.vp_workspace_set_option(fake, voxproj_host.VP_OPT_HEAVY_THRESHOLD, 5) == 0
    assert lib.vp_workspace_set_option(fake, voxproj_host.VP_OPT_MARCH_LDS_KB, -1) == 0
    assert lib.vp_workspace_set_option(fake, 99, 1) == -1 and b"unknown workspace option" in lib.vp_last_error()
    assert lib.vp_workspace_table_builds(fake) == 0
    assert lib.vp_workspace_release(fake) == 0 and lib.vp_workspace_release(fake) == 0      # releasing twice is harmless


def test_project_features_validates_its_arguments_before_touching_the_device():
    """Every refusal of vp_project_features[_f16] that the header documents is decided on the host (K.cu:374-414 checks
    nothing; W.cpp:38-61 checks dtypes only): device pointers here are fakes that are never dereferenced."""
    import voxproj_host
    lib = voxproj_host.lib()
    P = 0x7000_0000_0000                                   # "device" addresses, 256-byte aligned
    o = (ctypes.c_float * 5)(48, 32, 0.01, 10.0, 0.05)
    g = (ctypes.c_float * 3)(0, 0, 0)
    base = dict(feats=P, occ=P + 4096, vmi=P + 8192, intr=P + 12288, opts=o, count=P + 16384, out=P + 20480, views_hit=None,
                origin=g, vs=0.1, B=1, V=2, H=32, W=48, C=8, dz=10, dy=20, dx=30, n_rows=1001, ws=P + (1 << 20), ws_bytes=1 << 30,
                stream=None, flags=0)

    def call(entry=lib.vp_project_features, **kw):
        a = dict(base, **kw)
        return entry(a["feats"], a["occ"], a["vmi"], a["intr"], a["opts"], a["count"], a["out"], a["views_hit"], a["origin"],
                     ctypes.c_float(a["vs"]), a["B"], a["V"], a["H"], a["W"], a["C"], a["dz"], a["dy"], a["dx"], a["n_rows"],
                     a["ws"], a["ws_bytes"], a["stream"], a["flags"])

    cases = [
        (dict(feats=None), -1, b"null pointer"), (dict(count=None), -1, b"null pointer"), (dict(ws=None), -1, b"null pointer"),
        (dict(V=0), -1, b"non-positive"), (dict(C=-3), -1, b"non-positive"), (dict(n_rows=0), -1, b"non-positive"),
        (dict(B=256, V=257), -1, b"exceeds 65535"),
        (dict(dz=2048, dy=1024, dx=1024), -1, b"2^31 cells"),
        (dict(H=65536, W=32768, opts=(ctypes.c_float * 5)(32768, 65536, 0.01, 10.0, 0.05)), -1, b">= 2^31"),
        (dict(n_rows=1 << 31), -1, b">= 2^31"),
        (dict(flags=voxproj_host.VP_FLAG_SYNC | voxproj_host.VP_FLAG_PIPELINE), -1, b"exclude each other"),
        (dict(opts=(ctypes.c_float * 5)(47, 32, 0.01, 10.0, 0.05)), -1, b"must equal the feature map"),
        (dict(opts=(ctypes.c_float * 5)(48, 32, 0.01, 10.0, 0.0)), -1, b"rayIncrement must be > 0"),
        (dict(opts=(ctypes.c_float * 5)(48, 32, 0.01, 10.0, float("nan"))), -1, b"rayIncrement must be > 0"),
        (dict(ws_bytes=4096), -2, b"need"), (dict(ws=P + (1 << 20) + 16), -2, b"256-byte aligned"),
    ]
    for kw, rc, msg in cases:
        assert call(**kw) == rc, kw
        assert msg in lib.vp_last_error(), (kw, lib.vp_last_error())
    f16 = lib.vp_project_features_f16
    assert call(f16, C=12) == -1 and b"C % 8 == 0" in lib.vp_last_error()
    assert call(f16, out=P + 20480 + 4) == -1 and b"16-byte aligned" in lib.vp_last_error()
    assert call(f16, feats=None) == -1 and b"null pointer" in lib.vp_last_error()


def test_workspace_bytes_is_pure_host_arithmetic():
    import voxproj_host
    n = voxproj_host.workspace_bytes(1, 2, 48, 64, 16, 10, 20, 30, 1001)
    assert n >= 2 * 48 * 64 * 4 + 2 * 1001 * 4 and n % 256 == 0
    assert voxproj_host.workspace_bytes(0, 2, 48, 64, 16, 10, 20, 30, 1001) == 0


def _args(device="cpu"):
    B, V, H, W, C = 1, 1, 4, 4, 8
    return [torch.zeros(B, V, H, W, C, device=device), torch.zeros(B, 2, 2, 2, dtype=torch.int64, device=device),
            torch.eye(4, device=device).reshape(-1), torch.ones(B, 4, device=device),
            torch.tensor([W, H, 0.01, 10.0, 0.5]), torch.zeros(3, dtype=torch.int32, device=device),
            torch.zeros(3, C, device=device), torch.tensor([False]), torch.zeros(3), 1.0]


TEN = ["encoded_2d_features", "occupancy_3D", "viewMatrixInv", "intrinsicParams", "opts",
       "mapping2dto3d_num", "projected_features", "pred_mode_t", "grid_origin", "voxel_size"]


@pytest.fixture(scope="module")
def dropin():
    """The compiled drop-in module (built by the package's setup.py, ~1 min the first time) and the Python front."""
    import types

    import voxproj_host
    voxproj_host.build_ext()
    import torch  # noqa: F401  (libtorch must be loaded before the extension)

    import project_features_cuda as m
    import project_features_front as front
    assert m.__file__ == voxproj_host.ext_path(), "project_features_cuda is not the compiled extension module"
    return types.SimpleNamespace(project_features_cuda=m.project_features_cuda,
                                 project_features_cuda_py=front.project_features_cuda_py, module=m)


def test_wrapper_rejects_cpu_tensors_like_check_cuda(dropin):
    for fn in (dropin.project_features_cuda, dropin.project_features_cuda_py):
        with pytest.raises(RuntimeError, match="encoded_2d_features must be a CUDA tensor"):
            fn(*_args())


def test_wrapper_signature_is_ten_positional_arguments(dropin):
    import inspect
    assert list(inspect.signature(dropin.project_features_cuda_py).parameters) == TEN
    # the compiled front is a pybind11 builtin: its generated docstring carries the signature
    sig = dropin.project_features_cuda.__doc__.splitlines()[0]
    assert re.findall(r"(\w+): ", sig) == TEN and sig.endswith("-> None")
    # wrong arity raises TypeError on both, like any pybind11 / Python callable
    for fn in (dropin.project_features_cuda, dropin.project_features_cuda_py):
        with pytest.raises(TypeError):
            fn(*_args()[:9])


def test_the_product_path_has_no_cpu_fallback(tmp_path, monkeypatch):
    """A missing libvoxproj.so and a CPU tensor both RAISE: nothing in the product path routes through the oracle or any other CPU
    implementation (the oracle is test infrastructure; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs load it)."""
    import torch
    import voxproj_host
    import project_features_front
    # (1) the library is gone
    monkeypatch.setattr(voxproj_host, "LIB_PATH", str(tmp_path / "libvoxproj.so"))
    monkeypatch.setattr(voxproj_host, "_lib", None)
    with pytest.raises(voxproj_host.VoxprojError, match="no CPU fallback"):
        voxproj_host.lib()
    with pytest.raises(voxproj_host.VoxprojError, match="is missing"):
        voxproj_host.workspace_bytes(1, 1, 4, 4, 4, 2, 2, 2, 3)
    monkeypatch.undo()
    # (2) CPU tensors: the front refuses them with the reference wrapper's message (W.cpp:5: CHECK_CUDA), before any library call
    feats = torch.zeros(1, 1, 4, 4, 4)
    occ = torch.zeros(1, 2, 2, 2, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        project_features_front.project_features_cuda_py(feats, occ, torch.zeros(16), torch.zeros(1, 4), torch.zeros(5),
                                                        torch.zeros(3, dtype=torch.int32), torch.zeros(3, 4), torch.tensor([False]),
                                                        torch.zeros(3), 0.1)
    # (3) no module under the package imports the oracle
    pkg = os.path.join(ROOT, "3d-semantic-segmentation_amd")
    for name in os.listdir(pkg):
        if name.endswith(".py"):
            src = open(os.path.join(pkg, name)).read()
            assert "import oracle" not in src and "from oracle" not in src, name
