// project_features_ext.cpp -- compiled pybind11 front of the drop-in module `project_features_cuda`.
//
// Counterpart of the reference's C++ wrapper (cuda_project_image_to_sparse_voxel/project_image_cuda.cpp:23-79):
// the same ten positional arguments, the same TORCH_CHECK messages (raised to Python as RuntimeError), the same
// blocking, in-place "+=" behaviour -- and then ONE call of the C-ABI in include/voxproj.h.  Nothing is computed
// here and nothing falls back to the CPU: torch supplies the tensors, the current HIP stream and a scratch buffer.
//
// Built by voxproj_host.build_ext() (hipcc as a host C++17 compiler; links libvoxproj.so next to it).
#include <torch/extension.h>

#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>

#include <atomic>
#include <map>
#include <mutex>
#include <optional>
#include <vector>

#include "voxproj.h"

namespace {

// CHECK_INPUT = CHECK_CUDA + CHECK_CONTIGUOUS (project_image_cuda.cpp:5-7)
#define VP_CHECK_INPUT(x)                                           \
    TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor");        \
    TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")

// Per-device scratch and the identity of the occupancy grid whose derived tables it holds.
struct DeviceState {
    at::Tensor buf;                                                   // grow-only, from torch's allocator
    std::optional<c10::weak_intrusive_ptr<c10::TensorImpl>> occ;      // occupancy tensor of the previous call
    uint32_t occ_version = 0;
    const void *occ_ptr = nullptr;
    std::vector<int64_t> occ_shape;
    int64_t n_rows = 0;
    std::vector<int64_t> last_call;                                   // B,V,H,W,C,dimz,dimy,dimx,n_rows (test hook)
    uint64_t options_version = 0;                                     // g_options as last pushed to this device's workspace
};
std::mutex g_mu;
std::map<int, DeviceState> g_state;
// test / A-B switches (module functions set_exact_march / set_accel_cache; no environment variable is read)
std::atomic<bool> g_exact_march{false};   // VP_FLAG_EXACT_MARCH: evaluate every ray sample like kernel.cu:47-82
std::atomic<bool> g_accel_cache{true};    // keep the occupancy-derived tables between calls on an unchanged grid
std::map<int, long long> g_options;       // vp_workspace_set_option values for this module's workspaces (set_workspace_option)
uint64_t g_options_version = 1;           // both guarded by g_mu

void *aligned_ptr(const at::Tensor &buf)
{
    return reinterpret_cast<void *>((reinterpret_cast<uintptr_t>(buf.data_ptr()) + 255) & ~uintptr_t(255));
}

void project_features_cuda(at::Tensor encoded_2d_features, at::Tensor occupancy_3D, at::Tensor viewMatrixInv,
                           at::Tensor intrinsicParams, at::Tensor opts, at::Tensor mapping2dto3d_num,
                           at::Tensor projected_features, at::Tensor pred_mode_t, at::Tensor grid_origin,
                           float voxel_size)
{
    // device and contiguity (project_image_cuda.cpp:38-43)
    VP_CHECK_INPUT(encoded_2d_features);
    VP_CHECK_INPUT(occupancy_3D);
    VP_CHECK_INPUT(viewMatrixInv);
    VP_CHECK_INPUT(intrinsicParams);
    VP_CHECK_INPUT(mapping2dto3d_num);
    VP_CHECK_INPUT(projected_features);
    // dtypes (project_image_cuda.cpp:46-53)
    TORCH_CHECK(encoded_2d_features.scalar_type() == at::kFloat, "encoded_2d_features must be float32");
    TORCH_CHECK(occupancy_3D.scalar_type() == at::kLong, "occupancy_3D must be int64");
    TORCH_CHECK(viewMatrixInv.scalar_type() == at::kFloat, "viewMatrixInv must be float32");
    TORCH_CHECK(intrinsicParams.scalar_type() == at::kFloat, "intrinsicParams must be float32");
    TORCH_CHECK(opts.scalar_type() == at::kFloat, "opts must be float32");
    TORCH_CHECK(mapping2dto3d_num.scalar_type() == at::kInt, "mapping2dto3d_num must be int32");
    TORCH_CHECK(projected_features.scalar_type() == at::kFloat, "projected_features must be float32");
    TORCH_CHECK(pred_mode_t.scalar_type() == at::kBool, "pred_mode_t must be bool");
    // shapes (project_image_cuda.cpp:56-61)
    TORCH_CHECK(encoded_2d_features.dim() == 5, "encoded_2d_features must be 5D [B,V,H,W,C]");
    TORCH_CHECK(occupancy_3D.dim() == 4, "occupancy_3D must be 4D [B,Z,Y,X]");
    TORCH_CHECK(viewMatrixInv.dim() == 1, "viewMatrixInv must be 1D flattened");
    TORCH_CHECK(intrinsicParams.dim() == 2, "intrinsicParams must be 2D [B,4]");
    TORCH_CHECK(opts.dim() == 1 && opts.numel() == 5, "opts must be 1D with 5 elements");
    TORCH_CHECK(pred_mode_t.dim() == 1 && pred_mode_t.numel() == 1, "pred_mode_t must be scalar");

    // what the reference's launcher assumes without checking (project_image_cuda_kernel.cu:390-414): checked here
    // so that a bad call raises instead of reading or writing out of bounds
    const int64_t B = encoded_2d_features.size(0), V = encoded_2d_features.size(1), H = encoded_2d_features.size(2),
                  W = encoded_2d_features.size(3), C = encoded_2d_features.size(4);
    TORCH_CHECK(occupancy_3D.size(0) == B, "occupancy_3D batch size must match encoded_2d_features");
    TORCH_CHECK(viewMatrixInv.numel() == B * V * 16, "viewMatrixInv must hold B*V*16 floats");
    TORCH_CHECK(intrinsicParams.size(0) >= B && intrinsicParams.size(1) == 4, "intrinsicParams must be [B,4]");
    TORCH_CHECK(!grid_origin.is_cuda() && grid_origin.scalar_type() == at::kFloat && grid_origin.dim() == 1 &&
                    grid_origin.numel() >= 3,
                "grid_origin must be a 1D float32 CPU tensor with 3 elements");
    TORCH_CHECK(projected_features.dim() == 2 && projected_features.size(1) == C,
                "projected_features must be [num_ids, C]");
    TORCH_CHECK(mapping2dto3d_num.dim() == 1 && mapping2dto3d_num.size(0) == projected_features.size(0),
                "mapping2dto3d_num must be [num_ids] with num_ids = projected_features.size(0)");
    const auto dev = encoded_2d_features.device();
    TORCH_CHECK(occupancy_3D.device() == dev, "occupancy_3D must be on the same device as encoded_2d_features");
    TORCH_CHECK(viewMatrixInv.device() == dev, "viewMatrixInv must be on the same device as encoded_2d_features");
    TORCH_CHECK(intrinsicParams.device() == dev, "intrinsicParams must be on the same device as encoded_2d_features");
    TORCH_CHECK(mapping2dto3d_num.device() == dev, "mapping2dto3d_num must be on the same device as encoded_2d_features");
    TORCH_CHECK(projected_features.device() == dev, "projected_features must be on the same device as encoded_2d_features");

    const bool pred_mode = pred_mode_t.to(at::kCPU).item<bool>();                       // kernel.cu:427-429
    TORCH_CHECK(!pred_mode, "pred_mode_t = True (integer label projection) is not supported: the reference "
                            "branch reads float32 storage as int32 and is unreachable in its pipeline");
    const at::Tensor opts_cpu = opts.detach().to(at::kCPU).contiguous();                // kernel.cu:400-401
    const at::Tensor origin_cpu = grid_origin.contiguous();                             // kernel.cu:412-413
    TORCH_CHECK(B < (1ll << 31) && V < (1ll << 31) && H < (1ll << 31) && W < (1ll << 31) && C < (1ll << 31),
                "encoded_2d_features dimension too large");

    const int dimz = (int)occupancy_3D.size(1), dimy = (int)occupancy_3D.size(2), dimx = (int)occupancy_3D.size(3);
    const int64_t n_rows = mapping2dto3d_num.size(0);

    c10::hip::HIPGuard guard(dev.index());
    hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();

    std::lock_guard<std::mutex> lock(g_mu);
    DeviceState &st = g_state[dev.index()];
    const size_t need = vp_workspace_bytes((int)B, (int)V, (int)H, (int)W, (int)C, dimz, dimy, dimx, n_rows);
    TORCH_CHECK(need > 0, "voxproj: ", vp_last_error());
    if (!st.buf.defined() || (size_t)st.buf.numel() < need + 256) {
        if (st.buf.defined()) vp_workspace_release(aligned_ptr(st.buf));
        st.buf = at::empty({(int64_t)need + 256}, at::TensorOptions().dtype(at::kByte).device(dev));
        st.occ.reset();
        st.options_version = 0;
        void *fresh = aligned_ptr(st.buf);
        const int rc_ = vp_workspace_create(fresh, (size_t)st.buf.numel() - (size_t)((char *)fresh - (char *)st.buf.data_ptr()));
        TORCH_CHECK(rc_ == VP_OK, "voxproj error ", rc_, ": ", vp_last_error());
    }
    void *ws = aligned_ptr(st.buf);
    const size_t capacity = (size_t)st.buf.numel() - (size_t)((char *)ws - (char *)st.buf.data_ptr());
    if (st.options_version != g_options_version) {
        for (const auto &kv : g_options) {
            const int rc_ = vp_workspace_set_option(ws, kv.first, kv.second);
            TORCH_CHECK(rc_ == VP_OK, "voxproj error ", rc_, ": ", vp_last_error());
        }
        st.options_version = g_options_version;
    }

    // The occupancy-derived tables are reused only when this is the very same, still living tensor as in the
    // previous call with an unchanged version counter (an address match alone is not enough: the caching
    // allocator hands freed addresses out again).
    c10::TensorImpl *impl = occupancy_3D.unsafeGetTensorImpl();
    const bool tracked = !occupancy_3D.is_inference();
    const uint32_t version = tracked ? impl->version_counter().current_version() : 0;
    const bool cache = g_accel_cache.load();
    const bool reuse = tracked && st.occ && !st.occ->expired() && st.occ->_unsafe_get_target() == impl &&
                       st.occ_version == version && st.occ_ptr == occupancy_3D.data_ptr() &&
                       st.occ_shape == occupancy_3D.sizes().vec() && st.n_rows == n_rows && cache;
    // A/B arm of the leaping march: evaluate every ray sample (same results)
    // not the same tensor: let the library compare the grid with the copy its tables were built from (the reference's
    // caller makes a new, equal `.long()` tensor for every call, debug_project_features.py:143)
    const int flags = VP_FLAG_SYNC | (reuse ? VP_FLAG_REUSE_ACCEL : (cache ? VP_FLAG_VERIFY_ACCEL : 0)) |
                      (g_exact_march.load() ? VP_FLAG_EXACT_MARCH : 0);

    // blocks until the device is done, like the reference (kernel.cu:454-457); the binding below releases the GIL around it
    const int rc = vp_project_features(encoded_2d_features.data_ptr<float>(), occupancy_3D.data_ptr<int64_t>(),
                                       viewMatrixInv.data_ptr<float>(), intrinsicParams.data_ptr<float>(),
                                       opts_cpu.data_ptr<float>(), mapping2dto3d_num.data_ptr<int32_t>(),
                                       projected_features.data_ptr<float>(), nullptr, origin_cpu.data_ptr<float>(),
                                       voxel_size, (int)B, (int)V, (int)H, (int)W, (int)C, dimz, dimy, dimx, n_rows,
                                       ws, capacity, (void *)stream, flags);
    if (rc != VP_OK) {
        st.occ.reset();
        TORCH_CHECK(false, "voxproj error ", rc, ": ", vp_last_error());
    }
    st.occ = c10::weak_intrusive_ptr<c10::TensorImpl>(occupancy_3D.getIntrusivePtr());
    st.occ_version = version;
    st.occ_ptr = occupancy_3D.data_ptr();
    st.occ_shape = occupancy_3D.sizes().vec();
    st.n_rows = n_rows;
    st.last_call = {B, V, H, W, C, dimz, dimy, dimx, n_rows};
}

// Test/diagnostic hook: (workspace address, [B,V,H,W,C,dimz,dimy,dimx,n_rows]) of the last successful call on a
// device, for vp_copy_hit_image / vp_workspace_counters (the reference has no such output).
pybind11::tuple last_call(int device_index)
{
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_state.find(device_index);
    TORCH_CHECK(it != g_state.end() && it->second.buf.defined() && !it->second.last_call.empty(),
                "project_features_cuda has not been called on device ", device_index);
    return pybind11::make_tuple(reinterpret_cast<uintptr_t>(aligned_ptr(it->second.buf)), it->second.last_call);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    namespace py = pybind11;
    m.def("project_features_cuda", &project_features_cuda, "Projecting from 2D to 3D (MI355X / HIP)",
          py::arg("encoded_2d_features"), py::arg("occupancy_3D"), py::arg("viewMatrixInv"), py::arg("intrinsicParams"),
          py::arg("opts"), py::arg("mapping2dto3d_num"), py::arg("projected_features"), py::arg("pred_mode_t"),
          py::arg("grid_origin"), py::arg("voxel_size"),
          // the call blocks until the device is done (like the reference, kernel.cu:454-457) but, unlike it, without
          // holding the GIL: other Python threads -- a feature loader, say -- keep running meanwhile
          py::call_guard<py::gil_scoped_release>());
    m.def("abi_version", []() { return vp_abi_version(); });
    m.def("set_exact_march", [](bool on) { g_exact_march.store(on); },
          "A/B switch: evaluate every ray sample like the reference loop instead of leaping (same results, slower)");
    m.def("set_workspace_option", [](int option, long long value) {
              std::lock_guard<std::mutex> lock(g_mu);
              g_options[option] = value;
              g_options_version++;
          },
          "vp_workspace_set_option for this module's scratch workspaces (VP_OPT_HEAVY_THRESHOLD = 1, VP_OPT_MARCH_LDS_KB = 2; "
          "negative = default)");
    m.def("set_accel_cache", [](bool on) { g_accel_cache.store(on); },
          "keep the occupancy-derived tables between calls on an unchanged occupancy grid (default on)");
    m.def("last_call", &last_call, "workspace address and shape of the last call on a device (test hook)");
}
