// probe_rows.hip -- diagnostic only (tools/probe_levels.py): random whole-row gather into registers, the access shape of
// k_gather without the projector, restricted to a WINDOW of the buffer.  Not part of libvoxproj.so.
//
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/libprobe_rows.so tools/probe_rows.hip
//
// Every wavefront reads `iters` rows of `row_bytes` (1024 or 2048) picked by a hash of (wave, iteration) from the rows
// [first_row, first_row + window_rows) of `src`, 16 B per lane per load, non-temporal, 4 rows in flight -- what a gather
// wavefront does between two output rows.  Sweeping the window from 1 GB to the whole pool separates "how many distinct
// pages are live at once" from everything else that distinguishes one allocation from the next.
#include <hip/hip_runtime.h>
#include <cstdint>

namespace {

__device__ __forceinline__ unsigned long long mix(unsigned long long x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

// store_mode: 0 = no store; 1 = every wavefront ends by writing one row to a RANDOM row of `dst` (what the gather does with
// its finished output row); 2 = to the row of its own wave index (sequential, compact: a staging buffer); `dst_rows` rows.
// Variants of mode 1: 3 = the store is issued HALF-WAY through the wavefront's reads instead of at its end; 4 = non-temporal
// store; 5 = only the first 1 KiB of the row; 6 = two rows; 7 = the row is read first (read-modify-write, like the gather);
// 8 / 9 / 10 = the store carries the cache-policy bits sc1 / sc0 sc1 / sc0 sc1 nt (write-through past L2, inline assembly).
template <int LOADS>   // 1-KiB wave-loads per row
__global__ __launch_bounds__(256) void k_probe_rows(const float *__restrict__ src, long long first_row, long long window_rows,
                                                    int iters, unsigned long long seed, float *sink,
                                                    float *dst, long long dst_rows, int store_mode)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    const long long rrow = (long long)(mix(seed ^ (wave * 0xD1B54A32D192ED03ull)) % (unsigned long long)(dst_rows > 0 ? dst_rows : 1));
    if (store_mode == 7) {
        const float *q = dst + rrow * (LOADS * 256) + lane * 4;
#pragma unroll
        for (int k = 0; k < LOADS; k++) acc += *reinterpret_cast<const v4f *>(q + k * 256);
    }
    for (int it = 0; it < iters; it += U) {
        if (store_mode == 3 && it == ((iters / 2) & ~(U - 1))) {
            float *q = dst + rrow * (LOADS * 256) + lane * 4;
#pragma unroll
            for (int k = 0; k < LOADS; k++) *reinterpret_cast<v4f *>(q + k * 256) = acc;
        }
        v4f r[U][LOADS];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const unsigned long long h = mix(seed + wave * 0x9E3779B97F4A7C15ull + (unsigned long long)(it + j));
            const long long row = first_row + (long long)(h % (unsigned long long)window_rows);
            const float *p = src + row * (LOADS * 256) + lane * 4;
#pragma unroll
            for (int k = 0; k < LOADS; k++) r[j][k] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p + k * 256));
        }
#pragma unroll
        for (int j = 0; j < U; j++)
#pragma unroll
            for (int k = 0; k < LOADS; k++) acc += r[j][k];
    }
    if (store_mode != 0 && store_mode != 3) {
        const long long row = store_mode == 2 ? (long long)(wave % (unsigned long long)dst_rows) : rrow;
        float *q = dst + row * (LOADS * 256) + lane * 4;
        if (store_mode == 4) {
#pragma unroll
            for (int k = 0; k < LOADS; k++) __builtin_nontemporal_store(acc, reinterpret_cast<v4f *>(q + k * 256));
        } else if (store_mode >= 8 && store_mode <= 10) {
#pragma unroll
            for (int k = 0; k < LOADS; k++) {
                float *a = q + k * 256;
                if (store_mode == 8) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(a), "v"(acc) : "memory");
                else if (store_mode == 9) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(a), "v"(acc) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(a), "v"(acc) : "memory");
            }
        } else if (store_mode == 5) {
            *reinterpret_cast<v4f *>(q) = acc;
        } else {
#pragma unroll
            for (int k = 0; k < LOADS; k++) *reinterpret_cast<v4f *>(q + k * 256) = acc;
            if (store_mode == 6) {
                float *q2 = dst + ((rrow + dst_rows / 2) % dst_rows) * (LOADS * 256) + lane * 4;
#pragma unroll
                for (int k = 0; k < LOADS; k++) *reinterpret_cast<v4f *>(q2 + k * 256) = acc;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) sink[0] = acc.x;   // keeps the loads alive
}

// The same random whole-row read through raw buffer loads with a compile-time cache policy (aux bits on gfx940+: sc0 = 1,
// nt = 2, sc1 = 16): does any policy stream rows faster than the plain non-temporal global load the gather uses?
template <int AUX>
__global__ __launch_bounds__(256) void k_probe_rows_policy(const float *__restrict__ src, long long window_rows, int iters,
                                                           unsigned long long seed, float *sink)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;
    for (int it = 0; it < iters; it += U) {
        v4i r[U][2];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const unsigned long long h = mix(seed + wave * 0x9E3779B97F4A7C15ull + (unsigned long long)(it + j));
            const long long row = (long long)(h % (unsigned long long)window_rows);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src) + row * 512, 0, 2048, 0x00020000);
#pragma unroll
            for (int k = 0; k < 2; k++) r[j][k] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + k * 1024, 0, AUX);
        }
#pragma unroll
        for (int j = 0; j < U; j++)
#pragma unroll
            for (int k = 0; k < 2; k++) acc += __builtin_bit_cast(v4f, r[j][k]);
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) sink[0] = acc.x;
}

}  // namespace

extern "C" int probe_rows_policy(const float *src, long long window_rows, int waves, int iters, unsigned long long seed,
                                 float *sink, int aux, void *stream)
{
    if (!src || !sink || window_rows <= 0 || waves <= 0 || iters <= 0) return 1;
    const dim3 grid((unsigned)((waves + 3) / 4));
#define VP_POL(A) case A: hipLaunchKernelGGL(k_probe_rows_policy<A>, grid, dim3(256), 0, (hipStream_t)stream, src, window_rows, iters, seed, sink); break
    switch (aux) { VP_POL(0); VP_POL(1); VP_POL(2); VP_POL(3); VP_POL(16); VP_POL(17); VP_POL(18); VP_POL(19); default: return 1; }
#undef VP_POL
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int probe_rows_store(const float *src, long long first_row, long long window_rows, int row_bytes, int waves, int iters,
                                unsigned long long seed, float *sink, float *dst, long long dst_rows, int store_mode, void *stream)
{
    if (!src || !sink || window_rows <= 0 || waves <= 0 || iters <= 0 || (row_bytes != 1024 && row_bytes != 2048)) return 1;
    if (store_mode != 0 && (!dst || dst_rows <= 0)) return 1;
    const dim3 grid((unsigned)((waves + 3) / 4));
    if (row_bytes == 2048) hipLaunchKernelGGL(k_probe_rows<2>, grid, dim3(256), 0, (hipStream_t)stream, src, first_row, window_rows, iters, seed, sink, dst, dst_rows, store_mode);
    else hipLaunchKernelGGL(k_probe_rows<1>, grid, dim3(256), 0, (hipStream_t)stream, src, first_row, window_rows, iters, seed, sink, dst, dst_rows, store_mode);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int probe_rows(const float *src, long long first_row, long long window_rows, int row_bytes, int waves, int iters,
                          unsigned long long seed, float *sink, void *stream)
{
    return probe_rows_store(src, first_row, window_rows, row_bytes, waves, iters, seed, sink, nullptr, 0, 0, stream);
}
