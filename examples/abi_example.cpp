// abi_example.cpp -- the C-ABI of libvoxproj.so used from a plain HIP host program: no torch, no Python.
// One camera at the origin looks down +z at a filled occupancy plane (the K1 scene of tests/test_oracle_kat.py);
// prints the per-voxel pixel counts and feature sums, which tests/test_gpu_pipeline_rows.py compares with the oracle.
//
//   hipcc -O2 -Iinclude examples/abi_example.cpp -L3d-semantic-segmentation_amd -lvoxproj \
//         -Wl,-rpath,$PWD/3d-semantic-segmentation_amd -o /tmp/abi_example && /tmp/abi_example
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "voxproj.h"

#define HIP_OK(x)                                                                       \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

int main()
{
    const int B = 1, V = 1, H = 16, W = 16, C = 4, Z = 8, Y = 17, X = 17;
    const int64_t n_rows = 1 + Y * X;
    std::vector<int64_t> occ((size_t)Z * Y * X, 0);
    for (int y = 0; y < Y; y++)
        for (int x = 0; x < X; x++) occ[((size_t)5 * Y + y) * X + x] = 1 + y * X + x;      // plane z = 5
    std::vector<float> feats((size_t)H * W * C);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < C; c++) feats[((size_t)y * W + x) * C + c] = (float)(y * W + x) + 0.25f * c;
    const float c2w[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const float intr[4] = {8.0f, 8.0f, 8.0f, 8.0f};
    const float opts[5] = {(float)W, (float)H, 0.01f, 10.0f, 0.5f};
    const float origin[3] = {-8.0f, -8.0f, 0.0f};

    float *d_feats, *d_vmi, *d_intr, *d_out;
    int64_t *d_occ;
    int32_t *d_count;
    void *d_ws;
    const size_t ws_bytes = vp_workspace_bytes(B, V, H, W, C, Z, Y, X, n_rows);
    if (!ws_bytes) { fprintf(stderr, "vp_workspace_bytes: %s\n", vp_last_error()); return 1; }
    HIP_OK(hipMalloc(&d_feats, feats.size() * sizeof(float)));
    HIP_OK(hipMalloc(&d_occ, occ.size() * sizeof(int64_t)));
    HIP_OK(hipMalloc(&d_vmi, sizeof(c2w)));
    HIP_OK(hipMalloc(&d_intr, sizeof(intr)));
    HIP_OK(hipMalloc(&d_count, n_rows * sizeof(int32_t)));
    HIP_OK(hipMalloc(&d_out, n_rows * C * sizeof(float)));
    HIP_OK(hipMalloc(&d_ws, ws_bytes));                                    // hipMalloc is 256-byte aligned
    if (vp_workspace_create(d_ws, ws_bytes) != VP_OK) { fprintf(stderr, "vp_workspace_create: %s\n", vp_last_error()); return 1; }
    HIP_OK(hipMemcpy(d_feats, feats.data(), feats.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_occ, occ.data(), occ.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_vmi, c2w, sizeof(c2w), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_intr, intr, sizeof(intr), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_count, 0, n_rows * sizeof(int32_t)));
    HIP_OK(hipMemset(d_out, 0, n_rows * C * sizeof(float)));

    for (int call = 0; call < 2; call++) {                                 // outputs accumulate across calls (K.cu:77,88)
        const int rc = vp_project_features(d_feats, d_occ, d_vmi, d_intr, opts, d_count, d_out, nullptr, origin, 1.0f,
                                           B, V, H, W, C, Z, Y, X, n_rows, d_ws, ws_bytes, /*stream*/ nullptr,
                                           VP_FLAG_SYNC | (call ? VP_FLAG_REUSE_ACCEL : 0));
        if (rc != VP_OK) { fprintf(stderr, "vp_project_features: %d %s\n", rc, vp_last_error()); return 1; }
    }
    std::vector<int32_t> count(n_rows);
    std::vector<float> out((size_t)n_rows * C);
    HIP_OK(hipMemcpy(count.data(), d_count, n_rows * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    printf("abi %d rows %lld\n", vp_abi_version(), (long long)n_rows);
    for (int64_t i = 0; i < n_rows; i++)
        if (count[i]) printf("id %lld count %d sums %.9g %.9g %.9g %.9g\n", (long long)i, count[i], out[i * C], out[i * C + 1],
                             out[i * C + 2], out[i * C + 3]);
    vp_workspace_release(d_ws);
    return 0;
}
