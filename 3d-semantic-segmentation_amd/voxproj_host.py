"""ctypes binding of libvoxproj.so (the C-ABI declared in include/voxproj.h) plus workspace plumbing.

PyTorch is used only for device memory and streams.  There is NO CPU fallback: if the HIP library is
missing or no GPU is visible, the product entry points raise.
"""
import ctypes
import os
import subprocess
import threading
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VOXPROJ_LIB") or os.path.join(_HERE, "libvoxproj.so")

VP_OK = 0
VP_FLAG_SYNC = 1
VP_FLAG_REUSE_ACCEL = 2
VP_FLAG_EXACT_MARCH = 4
VP_FLAG_PIPELINE = 8
VP_FLAG_VERIFY_ACCEL = 16
VP_FLAG_SERIAL_SUMS = 32
VP_FLAG_GATHER_ONLY = 64

_lib = None
_lock = threading.Lock()

# test / A-B switches of the ctypes front (module attributes, not environment variables; the compiled front has
# project_features_cuda.set_exact_march / set_accel_cache)
EXACT_MARCH = False      # evaluate every ray sample like K.cu:47-82 (VP_FLAG_EXACT_MARCH)
ACCEL_CACHE = True       # keep the occupancy-derived tables between calls on the same, unmodified occupancy tensor
_default_options = {}    # {VP_OPT_*: value} applied to every Workspace of this module (set_default_option)
_options_version = 0


def set_default_option(option, value):
    """Default of a workspace option (VP_OPT_HEAVY_THRESHOLD, VP_OPT_MARCH_LDS_KB) for every Workspace object of this
    process, existing ones included (they pick it up at their next call); None = the library's default.  A Workspace's
    own set_option wins.  Test / A-B switch: production code leaves the defaults alone."""
    global _options_version
    if value is None:
        _default_options.pop(int(option), None)
    else:
        _default_options[int(option)] = int(value)
    _options_version += 1

EXPORTS = [
    "vp_abi_version", "vp_last_error", "vp_workspace_bytes", "vp_project_features",
    "vp_workspace_status", "vp_workspace_counters", "vp_copy_hit_image",
    "vp_profile_enable", "vp_profile_read", "vp_workspace_release", "vp_project_colors",
    "vp_project_features_f16", "vp_nearest_voxel",
    "vp_stream_read", "vp_workspace_table_builds", "vp_colors_workspace_bytes",
    "vp_upsample_workspace_bytes", "vp_upsample_features", "vp_voxel_coords", "vp_scatter_occupancy",
    "vp_aggregate_view_f16", "vp_workspace_create", "vp_workspace_set_option",
]
VP_ABI_VERSION = 4
VP_OPT_HEAVY_THRESHOLD = 1
VP_OPT_MARCH_LDS_KB = 2
VP_OPT_ROW_BEGIN = 3
VP_OPT_ROW_END = 4
VP_OPT_ONE_VIEW_GATHER = 5
VP_OPT_PART_PIXELS = 6
VP_OPT_ONE_VIEW_SPLIT = 7


class VoxprojError(RuntimeError):
    pass


def build(force=False):
    """Compile libvoxproj.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f == "voxproj.hip" or (f.startswith("vp_") and f.endswith(".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "voxproj.h"))
    newest = max(os.path.getmtime(f) for f in srcs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-s"] + (["-B"] if force else []))
    return LIB_PATH


def ext_path():
    """Where ``setup.py build_ext --inplace`` puts the compiled drop-in module ``project_features_cuda``."""
    import sysconfig
    return os.path.join(_HERE, "project_features_cuda" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_ext(force=False):
    """Build the compiled drop-in module with the package's setup.py (the counterpart of the reference's
    ``python setup.py install`` step, cuda_project_image_to_sparse_voxel/setup.py:10-27), in-tree: csrc/Makefile compiles
    the HIP kernels into libvoxproj.so, torch's BuildExtension compiles csrc/project_features_ext.cpp and links it to that
    library.  Returns the module's path."""
    import sys
    build(force)
    out = ext_path()
    src = os.path.join(_HERE, "csrc", "project_features_ext.cpp")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "voxproj.h")
    newest = max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(os.path.join(_HERE, "setup.py")))
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    subprocess.check_call([sys.executable, "setup.py", "-q", "build_ext", "--inplace"] + (["--force"] if force else []), cwd=_HERE)
    return out


def lib():
    """Load the shared library (dlopen only; no device call is made here)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise VoxprojError(
                    f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                    "(there is no CPU fallback)")
            L = ctypes.CDLL(LIB_PATH)
            L.vp_abi_version.restype = ctypes.c_int
            L.vp_last_error.restype = ctypes.c_char_p
            L.vp_workspace_bytes.restype = ctypes.c_size_t
            L.vp_workspace_bytes.argtypes = [ctypes.c_int] * 8 + [ctypes.c_int64]
            vp = ctypes.c_void_p
            L.vp_project_features.restype = ctypes.c_int
            L.vp_project_features.argtypes = [
                vp, vp, vp, vp, ctypes.POINTER(ctypes.c_float), vp, vp, vp, ctypes.POINTER(ctypes.c_float),
                ctypes.c_float] + [ctypes.c_int] * 8 + [ctypes.c_int64, vp, ctypes.c_size_t, vp, ctypes.c_int]
            L.vp_project_features_f16.restype = ctypes.c_int
            L.vp_project_features_f16.argtypes = L.vp_project_features.argtypes
            L.vp_nearest_voxel.restype = ctypes.c_int
            L.vp_nearest_voxel.argtypes = [vp, vp, vp, ctypes.POINTER(ctypes.c_double), ctypes.c_double, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, vp, ctypes.c_int64, vp, vp]
            L.vp_workspace_status.restype = ctypes.c_int
            L.vp_workspace_status.argtypes = [vp, vp]
            L.vp_workspace_counters.restype = ctypes.c_int
            L.vp_workspace_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, vp]
            L.vp_copy_hit_image.restype = ctypes.c_int
            L.vp_copy_hit_image.argtypes = [vp, vp] + [ctypes.c_int] * 8 + [ctypes.c_int64, vp]
            L.vp_profile_enable.restype = ctypes.c_int
            L.vp_profile_enable.argtypes = [ctypes.c_int]
            L.vp_profile_read.restype = ctypes.c_int
            L.vp_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)]
            L.vp_workspace_release.restype = ctypes.c_int
            L.vp_workspace_release.argtypes = [vp]
            L.vp_workspace_create.restype = ctypes.c_int
            L.vp_workspace_create.argtypes = [vp, ctypes.c_size_t]
            L.vp_workspace_set_option.restype = ctypes.c_int
            L.vp_workspace_set_option.argtypes = [vp, ctypes.c_int, ctypes.c_longlong]
            L.vp_workspace_table_builds.restype = ctypes.c_longlong
            L.vp_workspace_table_builds.argtypes = [vp]
            L.vp_project_colors.restype = ctypes.c_int
            L.vp_project_colors.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_float), ctypes.c_double, vp, ctypes.c_int,
                                            ctypes.c_int, vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int, vp,
                                            ctypes.c_size_t, vp]
            L.vp_colors_workspace_bytes.restype = ctypes.c_size_t
            L.vp_colors_workspace_bytes.argtypes = [ctypes.c_int64]
            L.vp_upsample_workspace_bytes.restype = ctypes.c_size_t
            L.vp_upsample_workspace_bytes.argtypes = [ctypes.c_int] * 4
            L.vp_upsample_features.restype = ctypes.c_int
            L.vp_upsample_features.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, vp, ctypes.c_size_t, vp]
            L.vp_voxel_coords.restype = ctypes.c_int
            L.vp_voxel_coords.argtypes = [vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_float), ctypes.c_float, vp, vp,
                                          ctypes.POINTER(ctypes.c_int32), vp]
            L.vp_scatter_occupancy.restype = ctypes.c_int
            L.vp_scatter_occupancy.argtypes = [vp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32), ctypes.c_int, ctypes.c_int,
                                               ctypes.c_int, vp, vp, vp]
            L.vp_aggregate_view_f16.restype = ctypes.c_int
            L.vp_aggregate_view_f16.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_int64, ctypes.c_int, vp]
            if L.vp_abi_version() != VP_ABI_VERSION:
                raise VoxprojError(f"{LIB_PATH} has ABI version {L.vp_abi_version()}, this package needs {VP_ABI_VERSION}: rebuild it")
            _lib = L
    return _lib


def last_error():
    return lib().vp_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != VP_OK:
        raise VoxprojError(f"voxproj error {rc}: {last_error()}")


def workspace_bytes(B, V, H, W, C, dimz, dimy, dimx, n_rows):
    return int(lib().vp_workspace_bytes(B, V, H, W, C, dimz, dimy, dimx, n_rows))


class Workspace:
    """Grow-only device scratch buffer (one per device), allocated through torch's allocator, announced to the library with
    vp_workspace_create and withdrawn with vp_workspace_release.  ``options``: {VP_OPT_*: value}, applied to every buffer
    this object ever holds (set_option)."""

    def __init__(self):
        self.buf = None
        self.accel_key = None     # (weakref to the occupancy tensor, its _version, shape, n_rows)
        self.options = {}
        self._applied = None      # (module options version, own options) last pushed to the library for self.buf

    def ensure(self, nbytes, device):
        import torch
        if self.buf is None or self.buf.numel() < nbytes + 256 or self.buf.device != device:
            if self.buf is not None:
                # growing while pipelined calls may be in flight: finish them before the old buffer goes away
                import torch as _t
                check(lib().vp_workspace_status(self.ptr(), _t.cuda.current_stream(self.buf.device).cuda_stream))
            self.release()
            self.buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
            self.accel_key = None
            check(lib().vp_workspace_create(self.ptr(), self.capacity()))      # whatever this address was before is forgotten
            self._applied = None
        self._push_options()
        return self.ptr()

    def _push_options(self):
        state = (_options_version, tuple(sorted(self.options.items())))
        if self.buf is None or self._applied == state:
            return
        merged = {VP_OPT_HEAVY_THRESHOLD: -1, VP_OPT_MARCH_LDS_KB: -1, VP_OPT_ROW_BEGIN: -1, VP_OPT_ROW_END: -1, VP_OPT_ONE_VIEW_GATHER: -1,
                  VP_OPT_PART_PIXELS: -1, VP_OPT_ONE_VIEW_SPLIT: -1}
        merged.update(_default_options)
        merged.update(self.options)
        for opt, val in merged.items():
            check(lib().vp_workspace_set_option(self.ptr(), int(opt), int(val)))
        self._applied = state

    def set_option(self, option, value):
        """vp_workspace_set_option (VP_OPT_HEAVY_THRESHOLD, VP_OPT_MARCH_LDS_KB); None = fall back to the module default /
        the library's.  Remembered, so it survives the buffer growing."""
        if value is None:
            self.options.pop(int(option), None)
        else:
            self.options[int(option)] = int(value)
        self._push_options()

    def set_row_range(self, begin=None, end=None):
        """Phase 2 of the following calls gathers only the voxel IDs in [begin, end) (VP_OPT_ROW_BEGIN / _END); no arguments:
        every row again.  With ``project_features_raw(..., gather_only=True)`` a call is cut into row ranges whose output rows
        are final one range after the other."""
        self.set_option(VP_OPT_ROW_BEGIN, begin)
        self.set_option(VP_OPT_ROW_END, end)

    def ptr(self):
        return (self.buf.data_ptr() + 255) & ~255

    def capacity(self):
        return self.buf.numel() - (self.ptr() - self.buf.data_ptr())

    def release(self):
        """Drop the library's side stream/events for this buffer (before the memory is recycled)."""
        if self.buf is not None and _lib is not None:
            _lib.vp_workspace_release(self.ptr())

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


_workspaces = {}


def get_workspace(device):
    """The ctypes front's per-device workspace (project_features_front.last_workspace covers both fronts)."""
    key = (device.type, device.index)
    ws = _workspaces.get(key)
    if ws is None:
        ws = _workspaces[key] = Workspace()
    return ws


def project_features_raw(feats, occ, vmi, intr, opts5, count, out, grid_origin3, voxel_size,
                         workspace=None, sync=True, reuse_accel=None, exact_march=None, pipeline=False,
                         views_hit=None, verify_accel=False, gather_only=False, serial_sums=False, extra_flags=0):
    """Call vp_project_features (or vp_project_features_f16 when ``feats`` is float16) on torch CUDA tensors
    (already validated by the caller).

    opts5 / grid_origin3 are python sequences of floats.  Returns the Workspace used.
    ``reuse_accel``: None = reuse the occupancy-derived tables only if ``occ`` is the very same (still
    alive) tensor object as in the previous call on this workspace, with an unchanged torch version
    counter -- a data_ptr match alone is not enough, the caching allocator hands freed addresses out
    again; True/False = force.  ``exact_march``: evaluate every ray sample (A/B arm of the leaping march;
    default: module attribute EXACT_MARCH).  ``pipeline``: asynchronous job mode (VP_FLAG_PIPELINE): phase 1 of
    this call overlaps the previous call's gather; the caller must keep occ/vmi/intr alive and unchanged
    until ``workspace_status`` (or a device synchronise) and must not pass sync.  ``views_hit``: optional
    int32 [n_rows] tensor, += number of views of this call that hit each voxel.  ``verify_accel``: when the tables
    are not reused by identity (blocking calls only), let the library compare the grid with the copy the tables were
    built from and rebuild only if it changed (VP_FLAG_VERIFY_ACCEL) -- for callers that make a new, equal
    occupancy tensor for every call.  ``gather_only``: VP_FLAG_GATHER_ONLY -- no ray-march, phase 2 of the PREVIOUS call on
    this workspace (same tensors) once more, for the row range now set with ``Workspace.set_row_range``.
    ``serial_sums``: VP_FLAG_SERIAL_SUMS -- every voxel summed by one wavefront in (b, v, y, x) order, the oracle's bits (no
    workgroup path for voxels with very many pixels).
    """
    import torch
    B, V, H, W, C = feats.shape
    _, dimz, dimy, dimx = occ.shape
    n_rows = int(count.shape[0])
    ws = workspace if workspace is not None else get_workspace(feats.device)
    need = workspace_bytes(B, V, H, W, C, dimz, dimy, dimx, n_rows)
    ptr = ws.ensure(need, feats.device)
    key = (occ._version, occ.data_ptr(), tuple(occ.shape), n_rows)
    if reuse_accel is None:
        prev = ws.accel_key
        reuse_accel = (prev is not None and prev[0]() is occ and prev[1] == key and ACCEL_CACHE)
    if exact_march is None:
        exact_march = EXACT_MARCH
    flags = ((VP_FLAG_SYNC if (sync and not pipeline) else 0) | (VP_FLAG_REUSE_ACCEL if reuse_accel else 0)
             | (VP_FLAG_EXACT_MARCH if exact_march else 0) | (VP_FLAG_PIPELINE if pipeline else 0)
             | (VP_FLAG_VERIFY_ACCEL if (verify_accel and sync and not pipeline and not reuse_accel and ACCEL_CACHE) else 0)
             | (VP_FLAG_GATHER_ONLY if gather_only else 0) | (VP_FLAG_SERIAL_SUMS if serial_sums else 0) | int(extra_flags))
    o = (ctypes.c_float * 5)(*[float(v) for v in opts5])
    g = (ctypes.c_float * 3)(*[float(v) for v in grid_origin3])
    stream = torch.cuda.current_stream(feats.device).cuda_stream
    entry = lib().vp_project_features_f16 if feats.dtype == torch.float16 else lib().vp_project_features
    with torch.cuda.device(feats.device):
        rc = entry(
            feats.data_ptr(), occ.data_ptr(), vmi.data_ptr(), intr.data_ptr(), o,
            count.data_ptr(), out.data_ptr(), views_hit.data_ptr() if views_hit is not None else None,
            g, ctypes.c_float(float(voxel_size)),
            B, V, H, W, C, dimz, dimy, dimx, n_rows, ptr, ws.capacity(), stream, flags)
    if rc != VP_OK:
        ws.accel_key = None
        check(rc)
    ws.accel_key = (weakref.ref(occ), key)
    ws.last_shape = (B, V, H, W, C, dimz, dimy, dimx, n_rows)
    return ws


class PreparedViewCalls:
    """vp_project_features for a sequence of same-shaped calls on one scene, with everything that does not change between
    calls bound once (shapes, intrinsics, ray options, outputs, workspace, stream): the per-call work on the host is two
    pointer reads and one foreign call.  Used by the aggregator's one-view-per-call parity mode, where the kernels of a
    call run for ~0.2 ms and the general wrapper's Python would otherwise set the pace.  The first call builds the
    occupancy tables, the following ones reuse them (the caller keeps ``occ`` alive and unmodified meanwhile)."""

    def __init__(self, occ, intr, opts5, count, out, grid_origin3, voxel_size, workspace, shape, views_hit=None, flags=0):
        import torch
        self.B, self.V, self.H, self.W, self.C = (int(v) for v in shape)
        _, self.dimz, self.dimy, self.dimx = (int(v) for v in occ.shape)
        self.n_rows = int(count.shape[0])
        self.dev = occ.device
        self.keep = (occ, intr, count, out, views_hit)
        self.ws = workspace
        need = workspace_bytes(self.B, self.V, self.H, self.W, self.C, self.dimz, self.dimy, self.dimx, self.n_rows)
        self.ptr = workspace.ensure(need, self.dev)
        self.cap = workspace.capacity()
        self.o = (ctypes.c_float * 5)(*[float(v) for v in opts5])
        self.g = (ctypes.c_float * 3)(*[float(v) for v in grid_origin3])
        self.vs = ctypes.c_float(float(voxel_size))
        self.stream = torch.cuda.current_stream(self.dev).cuda_stream
        self.fn32, self.fn16 = lib().vp_project_features, lib().vp_project_features_f16
        self.occ_ptr, self.intr_ptr = occ.data_ptr(), intr.data_ptr()
        self.count_ptr, self.out_ptr = count.data_ptr(), out.data_ptr()
        self.views_ptr = views_hit.data_ptr() if views_hit is not None else None
        self.flags = int(flags)
        self.built = False
        workspace.accel_key = None
        workspace.last_shape = (self.B, self.V, self.H, self.W, self.C, self.dimz, self.dimy, self.dimx, self.n_rows)

    def __call__(self, feats, vmi):
        """feats: CUDA tensor [B,V,H,W,C] (float32 or float16, contiguous), vmi: CUDA float32 [B*V*16]; asynchronous."""
        import torch
        if torch.cuda.current_device() != self.dev.index:
            torch.cuda.set_device(self.dev)       # the C-ABI works on the calling thread's current device
        fn = self.fn16 if feats.dtype == torch.float16 else self.fn32
        rc = fn(feats.data_ptr(), self.occ_ptr, vmi.data_ptr(), self.intr_ptr, self.o, self.count_ptr, self.out_ptr,
                self.views_ptr, self.g, self.vs, self.B, self.V, self.H, self.W, self.C, self.dimz, self.dimy, self.dimx,
                self.n_rows, self.ptr, self.cap, self.stream, self.flags | (VP_FLAG_REUSE_ACCEL if self.built else 0))
        if rc != VP_OK:
            self.built = False
            check(rc)
        self.built = True


def hit_image(ws, device):
    """First-hit ID image of the last call on ``ws`` as an int32 [B,V,H,W] tensor (test hook)."""
    import torch
    B, V, H, W, C, dimz, dimy, dimx, n_rows = ws.last_shape
    dst = torch.empty((B, V, H, W), dtype=torch.int32, device=device)
    ptr = ws.ptr()
    stream = torch.cuda.current_stream(device).cuda_stream
    check(lib().vp_copy_hit_image(ptr, dst.data_ptr(), B, V, H, W, C, dimz, dimy, dimx, n_rows, stream))
    torch.cuda.current_stream(device).synchronize()
    return dst


def counters(ws, device):
    """Device-side diagnostic counters of the last call: dict(bad_id, box_miss, n_heavy, heavy_t, n_parts, n_split) -- n_heavy = voxels
    above the heavy threshold in force (heavy_t: the option or min(256 + 64*B*V, 2048), raised to the part-slot bound where that binds;
    one-view calls: the device's choice, see below), n_split = the voxels cut into parts, n_parts = their parts; part_t / part_px =
    pixels above which a voxel was cut and pixels per part; n_hit = pixels whose ray hit a voxel (counted only by one-view calls
    that size their parts from it: part_px = max(32, 2 * n_hit / slots), part_t = heavy_t = 2 * part_px)."""
    import torch
    arr = (ctypes.c_int32 * 32)()
    ptr = ws.ptr()
    stream = torch.cuda.current_stream(device).cuda_stream
    check(lib().vp_workspace_counters(ptr, arr, 32, stream))
    return dict(bad_id=int(arr[0]), box_miss=int(arr[1]), n_heavy=int(arr[2]), heavy_t=int(arr[7]), n_parts=int(arr[24]), n_split=int(arr[25]),
                n_hit=int(arr[8]), part_t=int(arr[9]), part_px=int(arr[10]))


def table_builds(ws):
    """How many times the occupancy-derived tables of workspace ``ws`` have been built (diagnostic)."""
    return int(lib().vp_workspace_table_builds(ws.ptr()))


def profile_enable(on=True):
    check(lib().vp_profile_enable(1 if on else 0))


def profile_read():
    """Summed HIP-event milliseconds and launch counts per kernel group since the last read."""
    ms = (ctypes.c_double * 4)()
    n = (ctypes.c_int64 * 4)()
    check(lib().vp_profile_read(ms, n))
    return dict(prep_ms=ms[0], first_hit_ms=ms[1], gather_ms=ms[2], heavy_ms=ms[3],
                prep_launches=int(n[0]), first_hit_launches=int(n[1]), gather_launches=int(n[2]),
                heavy_launches=int(n[3]))


def workspace_status(ws, device):
    """Synchronise and raise if any call on ``ws`` reported a device-side error (asynchronous callers)."""
    import torch
    stream = torch.cuda.current_stream(device).cuda_stream
    check(lib().vp_workspace_status(ws.ptr(), stream))


def project_colors_raw(occ_zyx, c2w, intr, grid_origin3, voxel_size, images, color_sum, hit_count, first_view=None,
                       view_base=0, pixel_uv=None):
    """vp_project_colors on torch CUDA tensors: occ i32 [Z,Y,X], c2w f32 [V,4,4], intr f32 [V,4],
    images u8 [V,H,W,3]; color_sum f32 [n_rows,3], hit_count i32 [n_rows], first_view i32 [n_rows] or None,
    pixel_uv i32 [V,n_rows,2] or None (receives the sampled pixel per view and voxel ID, -1 where unseen)."""
    import torch
    dev = occ_zyx.device
    assert occ_zyx.is_cuda and occ_zyx.dtype == torch.int32 and occ_zyx.is_contiguous()
    V = int(images.shape[0])
    assert images.dtype == torch.uint8 and images.is_contiguous() and images.shape[-1] == 3
    assert c2w.dtype == torch.float32 and c2w.is_contiguous() and c2w.numel() == V * 16
    assert intr.dtype == torch.float32 and intr.is_contiguous() and intr.numel() == V * 4
    assert color_sum.dtype == torch.float32 and color_sum.is_contiguous() and hit_count.dtype == torch.int32
    n_rows = int(hit_count.shape[0])
    if pixel_uv is not None:
        assert pixel_uv.dtype == torch.int32 and pixel_uv.is_contiguous() and tuple(pixel_uv.shape) == (V, n_rows, 2)
    need = int(lib().vp_colors_workspace_bytes(n_rows))
    scratch = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    ptr = (scratch.data_ptr() + 255) & ~255
    g = (ctypes.c_float * 3)(*[float(v) for v in grid_origin3])
    Z, Y, X = occ_zyx.shape
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        check(lib().vp_project_colors(
            occ_zyx.data_ptr(), Z, Y, X, c2w.data_ptr(), intr.data_ptr(), V, g, ctypes.c_double(float(voxel_size)),
            images.data_ptr(), int(images.shape[1]), int(images.shape[2]), color_sum.data_ptr(), hit_count.data_ptr(),
            first_view.data_ptr() if first_view is not None else None,
            pixel_uv.data_ptr() if pixel_uv is not None else None, n_rows, int(view_base), ptr, need, stream))


def upsample_features(src_chw, H, W, keep_dtype=False, out=None):
    """vp_upsample_features: CUDA tensor [C,h,w] (float16 or float32) -> channels-last [H,W,C], float32 or (float16
    source with ``keep_dtype``) float16.  The counterpart of prepare_tensor_data.py:119-127,152,183-185; asynchronous on
    the current stream.  ``out``: optional destination tensor (e.g. a slot of a resident pool)."""
    import torch
    assert src_chw.is_cuda and src_chw.dim() == 3 and src_chw.dtype in (torch.float16, torch.float32)
    src = src_chw.contiguous()
    C, h, w = (int(v) for v in src.shape)
    src16 = src.dtype == torch.float16
    dst_dtype = torch.float16 if (keep_dtype and src16) else torch.float32
    if out is None:
        out = torch.empty((H, W, C), dtype=dst_dtype, device=src.device)
    assert out.is_cuda and out.is_contiguous() and tuple(out.shape) == (H, W, C) and out.dtype == dst_dtype
    need = int(lib().vp_upsample_workspace_bytes(C, h, w, int(src16)))
    scratch = torch.empty(need + 256, dtype=torch.uint8, device=src.device)
    ptr = (scratch.data_ptr() + 255) & ~255
    stream = torch.cuda.current_stream(src.device).cuda_stream
    with torch.cuda.device(src.device):
        check(lib().vp_upsample_features(src.data_ptr(), int(src16), C, h, w, out.data_ptr(), int(dst_dtype == torch.float16),
                                         int(H), int(W), ptr, need, stream))
    return out


def build_occupancy_device(points_xyz, grid_origin3, voxel_size):
    """vp_voxel_coords + vp_scatter_occupancy: CUDA float32 [N,3] points -> (occ int32 [Z,Y,X] on the same device,
    min_coord int[3] before the shift).  build_sparse_occupancy.py:30-53: half-to-even rounding in float32, the grid is
    shifted to start at zero when any coordinate is negative, the last vertex wins on duplicates."""
    import torch
    assert points_xyz.is_cuda and points_xyz.dtype == torch.float32 and points_xyz.dim() == 2 and points_xyz.shape[1] == 3
    pts = points_xyz.contiguous()
    N = int(pts.shape[0])
    dev = pts.device
    coords = torch.empty((N, 3), dtype=torch.int32, device=dev)
    scratch = torch.zeros(8, dtype=torch.int32, device=dev)
    mm = (ctypes.c_int32 * 6)()
    g = (ctypes.c_float * 3)(*[float(v) for v in grid_origin3])
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        check(lib().vp_voxel_coords(pts.data_ptr(), N, g, ctypes.c_float(float(voxel_size)), coords.data_ptr(),
                                    scratch.data_ptr(), mm, stream))
        lo, hi = [int(mm[k]) for k in range(3)], [int(mm[3 + k]) for k in range(3)]
        shift = lo if min(lo) < 0 else [0, 0, 0]                       # BSO:36-39
        dx, dy, dz = (hi[k] - shift[k] + 1 for k in range(3))          # BSO:40-41
        occ = torch.empty((dz, dy, dx), dtype=torch.int32, device=dev)
        sh = (ctypes.c_int32 * 3)(*shift)
        check(lib().vp_scatter_occupancy(coords.data_ptr(), N, sh, dz, dy, dx, occ.data_ptr(), scratch.data_ptr(), stream))
    return occ, lo


_agg_cache = {}


def aggregate_view_f16(view_sum, view_count, run16, views, first_view, view_index, nonfinite, flag_index=0, stream=None):
    """vp_aggregate_view_f16 (aggregate_voxel_features_onthefly.py:307-313 over the rows hit in this view; leaves
    view_sum / view_count zeroed).  ``nonfinite`` int32 tensor, element ``flag_index`` receives the view's NaN/Inf flag.
    Asynchronous on ``stream`` (a raw hipStream_t value; default: torch's current stream).  The tensors' checks and
    pointers are cached per argument set (the aggregator calls this once per view with the same tensors)."""
    import torch
    # the cache key holds the SHAPES as well as the addresses: the caching allocator hands freed addresses out again, and a
    # second aggregator with fewer rows or channels can receive the very same six pointers
    key = (view_sum.data_ptr(), view_count.data_ptr(), run16.data_ptr(), views.data_ptr(), first_view.data_ptr(), nonfinite.data_ptr(),
           tuple(view_sum.shape), tuple(run16.shape), view_count.numel(), views.numel(), first_view.numel())
    c = _agg_cache.get("k")
    if c is None or c[0] != key:
        n_rows, C = (int(v) for v in view_sum.shape)
        assert view_sum.dtype == torch.float32 and view_sum.is_contiguous() and view_count.dtype == torch.int32
        assert run16.dtype == torch.float16 and run16.is_contiguous() and tuple(run16.shape) == (n_rows, C)
        assert views.dtype == torch.int32 and first_view.dtype == torch.int32 and nonfinite.dtype == torch.int32
        assert view_count.numel() == n_rows and views.numel() == n_rows and first_view.numel() == n_rows
        assert view_count.is_contiguous() and views.is_contiguous() and first_view.is_contiguous()
        c = _agg_cache["k"] = (key, n_rows, C, lib().vp_aggregate_view_f16, view_sum.device)
    _, n_rows, C, fn, dev = c
    if stream is None:
        stream = torch.cuda.current_stream(dev).cuda_stream
    assert 0 <= flag_index < nonfinite.numel()
    if torch.cuda.current_device() != dev.index:
        torch.cuda.set_device(dev)
    check(fn(key[0], key[1], key[2], key[3], key[4], int(view_index), key[5] + 4 * int(flag_index), n_rows, C, stream))


class _ContiguousDeviceMemory:
    """Owner of one hipExtMallocWithFlags(hipDeviceMallocContiguous) allocation, exported to torch through the CUDA array
    interface (torch keeps this object alive for as long as a tensor views the memory; freeing happens here)."""

    _hip = None

    def __init__(self, nbytes, shape, typestr, flags=0x4):
        cls = _ContiguousDeviceMemory
        if cls._hip is None:
            cls._hip = ctypes.CDLL("libamdhip64.so")
            cls._hip.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
            cls._hip.hipExtMallocWithFlags.restype = ctypes.c_int
            cls._hip.hipFree.argtypes = [ctypes.c_void_p]
        p = ctypes.c_void_p()
        # 0x4 hipDeviceMallocContiguous (default); experiments also use 0x1 hipDeviceMallocFinegrained, 0x3 hipDeviceMallocUncached
        rc = cls._hip.hipExtMallocWithFlags(ctypes.byref(p), max(int(nbytes), 256), int(flags))
        if rc != 0 or not p.value:
            raise MemoryError(f"hipExtMallocWithFlags(flags {flags:#x}, {nbytes} bytes) failed: hip error {rc}")
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": typestr, "data": (self.ptr, False),
                                         "version": 2}

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self._hip.hipFree(ctypes.c_void_p(self.ptr))        # hipFree waits for the device to be done with it
                self.ptr = None
        except Exception:
            pass


def resident_empty(shape, dtype, device, fallback=True, flags=0x4):
    """An uninitialised CUDA tensor for a long-lived, heavily gathered buffer (the resident feature maps, the output rows)
    in PHYSICALLY CONTIGUOUS device memory (hipExtMallocWithFlags, hipDeviceMallocContiguous).  An experiment on the
    placement spread of DESIGN.md section 4 (round 2's probe, in git history): inside one process a 17 GB pool re-allocated this way
    kept one speed level (2.700-2.705 ms per 16-view launch, against 2.72-3.09 ms over plain re-allocations), but a 35 GB
    pool showed no difference and the spread between processes stayed -- so nothing uses it by default (bench.py --alloc
    contiguous).  With ``fallback`` a failed contiguous allocation (no contiguous range that large) falls back to torch's
    allocator.  Returns (tensor, "contiguous" | "default")."""
    import torch
    dev = torch.device(device)
    typestr = {torch.float32: "<f4", torch.float16: "<f2", torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1"}[dtype]
    n = 1
    for v in shape:
        n *= int(v)
    try:
        with torch.cuda.device(dev):
            mem = _ContiguousDeviceMemory(n * torch.empty((), dtype=dtype).element_size(), shape, typestr, flags)
            return torch.as_tensor(mem, device=dev), "contiguous"
    except (MemoryError, OSError):
        if not fallback:
            raise
        return torch.empty(tuple(int(v) for v in shape), dtype=dtype, device=dev), "default"


def stream_read_gbs(buf, repeats=3):
    """Measured streaming-read rate (GB/s) of this GPU over the float32 CUDA tensor ``buf`` (non-temporal loads)."""
    import torch
    L = lib()
    L.vp_stream_read.restype = ctypes.c_int
    L.vp_stream_read.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    sink = torch.zeros(4, dtype=torch.float32, device=buf.device)
    n = (buf.numel() * buf.element_size() // 16) * 4
    stream = torch.cuda.current_stream(buf.device).cuda_stream
    check(L.vp_stream_read(buf.data_ptr(), n, sink.data_ptr(), stream))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(repeats):
        check(L.vp_stream_read(buf.data_ptr(), n, sink.data_ptr(), stream))
    e1.record()
    e1.synchronize()
    return n * 4 * repeats / (e0.elapsed_time(e1) * 1e-3) / 1e9
