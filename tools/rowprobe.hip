// rowprobe.hip -- stand-alone probe: how fast can gfx950 read 2-KiB rows of a large buffer, as a function of the
// buffer's (physical) placement?  bench.py shows k_gather settling at different speeds from one allocation of the
// feature pool to the next while a linear streaming read does not move; this program reproduces the access shapes
// without the projector:
//   stream   linear 16 B/lane non-temporal read of the whole buffer
//   rows     every 2-KiB row once, in a pseudo-random order (one wavefront per row, 4 rows in flight)
//   boxes    the gather's shape: one wavefront per "voxel" walks V "views" (1 GiB apart), in each a 3x3-pixel box of
//            rows (image rows W*2 KiB apart), neighbouring wavefronts on neighbouring boxes
// for several allocations of the buffer (freed and re-allocated behind dummy allocations of changing size).
//
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/rowprobe.hip -o /tmp/rowprobe && /tmp/rowprobe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ f4 ld(const f4 *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

__global__ __launch_bounds__(256) void k_stream(const f4 *__restrict__ src, size_t n_f4, float *sink)
{
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_f4; i += stride * 4) {
        f4 a = ld<true>(src + i);
        f4 b = i + stride < n_f4 ? ld<true>(src + i + stride) : f4{0, 0, 0, 0};
        f4 c = i + 2 * stride < n_f4 ? ld<true>(src + i + 2 * stride) : f4{0, 0, 0, 0};
        f4 d = i + 3 * stride < n_f4 ? ld<true>(src + i + 3 * stride) : f4{0, 0, 0, 0};
        acc += a + b + c + d;
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = 1.0f;
}

// every row once, order = i * mult mod n_rows (mult coprime to n_rows)
template <bool NT>
__global__ __launch_bounds__(256) void k_rows(const f4 *__restrict__ src, size_t n_rows, size_t mult, float *sink)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = wave * 4; i < n_rows; i += n_waves * 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (i + k < n_rows) {
                const size_t r = ((i + k) * mult) % n_rows;
                const f4 *p = src + r * 128;
                acc += ld<NT>(p + lane) + ld<NT>(p + 64 + lane);
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = 1.0f;
}

// the gather's shape: wave w owns box (bx, by) of the image; for each view, 3 image rows x 3 pixels
template <bool NT>
__global__ __launch_bounds__(256) void k_boxes(const f4 *__restrict__ src, int V, int H, int W, float *sink)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int bw = W / 3, bh = H / 3;
    if (wave >= (size_t)bw * bh) return;
    const int bx = (int)(wave % bw) * 3, by = (int)(wave / bw) * 3;
    f4 acc = {0, 0, 0, 0};
    for (int v = 0; v < V; v++) {
        const f4 *img = src + (size_t)v * H * W * 128;
#pragma unroll
        for (int y = 0; y < 3; y++) {
            const f4 *p = img + ((size_t)(by + y) * W + bx) * 128;
            acc += ld<NT>(p + lane) + ld<NT>(p + 64 + lane) + ld<NT>(p + 128 + lane) + ld<NT>(p + 192 + lane) +
                   ld<NT>(p + 256 + lane) + ld<NT>(p + 320 + lane);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) *sink = 1.0f;
}

// Allocation strategies: 0 = hipMalloc; g > 0 = virtual-memory API, physical handles of g MiB mapped back to back
struct Buf {
    f4 *ptr = nullptr;
    size_t bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    size_t chunk = 0;
};

static Buf alloc_buf(size_t bytes, int chunk_mib)
{
    Buf b;
    if (chunk_mib == 0) {
        CHECK(hipMalloc(&b.ptr, bytes));
        b.bytes = bytes;
        return b;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    size_t chunk = (size_t)chunk_mib << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    b.bytes = n * chunk;
    b.chunk = chunk;
    void *va = nullptr;
    CHECK(hipMemAddressReserve(&va, b.bytes, chunk > (1ull << 30) ? (1ull << 30) : chunk, nullptr, 0));
    for (size_t i = 0; i < n; i++) {
        hipMemGenericAllocationHandle_t h;
        CHECK(hipMemCreate(&h, chunk, &prop, 0));
        CHECK(hipMemMap((char *)va + i * chunk, chunk, 0, h, 0));
        b.handles.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CHECK(hipMemSetAccess(va, b.bytes, &acc, 1));
    b.ptr = (f4 *)va;
    return b;
}

static void free_buf(Buf &b)
{
    if (b.handles.empty()) {
        CHECK(hipFree(b.ptr));
    } else {
        CHECK(hipMemUnmap(b.ptr, b.bytes));
        for (auto h : b.handles) CHECK(hipMemRelease(h));
        CHECK(hipMemAddressFree(b.ptr, b.bytes));
    }
    b = Buf();
}

template <class F>
static double time_ms(F f, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    f();
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) f();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int V = argc > 1 ? atoi(argv[1]) : 32, H = 548, W = 968;
    const int rounds = argc > 2 ? atoi(argv[2]) : 8;
    const int chunk_mib = argc > 3 ? atoi(argv[3]) : 0;     // 0 = hipMalloc, else VMM handles of this many MiB
    const size_t bytes = (size_t)V * H * W * 2048;
    const size_t n_rows = bytes / 2048, n_f4 = bytes / 16;
    float *sink;
    CHECK(hipMalloc(&sink, 256));
    size_t mult = 2654435761ull % n_rows;
    auto gcd = [](size_t a, size_t b) { while (b) { size_t t = a % b; a = b; b = t; } return a; };
    while (gcd(mult, n_rows) != 1) mult++;
    printf("buffer %.2f GB = %d views of %dx%d rows of 2 KiB, %s\n", bytes / 1e9, V, W, H,
           chunk_mib ? "virtual-memory API" : "hipMalloc");
    if (chunk_mib) printf("physical handles of %d MiB\n", chunk_mib);
    unsigned seed = 12345;
    for (int r = 0; r < rounds; r++) {
        std::vector<void *> dummies;
        seed = seed * 1664525u + 1013904223u;
        const int nd = (r == 0) ? 0 : (seed >> 8) % 4;
        for (int d = 0; d < nd; d++) {
            seed = seed * 1664525u + 1013904223u;
            void *p;
            CHECK(hipMalloc(&p, ((size_t)(seed >> 10) % 3000 + 1) << 20));
            dummies.push_back(p);
        }
        Buf bb = alloc_buf(bytes, chunk_mib);
        f4 *buf = bb.ptr;
        CHECK(hipMemset(buf, 0, bytes));
        const int grid = 256 * 8;
        const double s = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(grid * 4), dim3(256), 0, 0, buf, n_f4, sink); }, 3);
        const double rn = time_ms([&] { hipLaunchKernelGGL(k_rows<true>, dim3(grid * 2), dim3(256), 0, 0, buf, n_rows, mult, sink); }, 3);
        const double rd = time_ms([&] { hipLaunchKernelGGL(k_rows<false>, dim3(grid * 2), dim3(256), 0, 0, buf, n_rows, mult, sink); }, 3);
        const size_t boxes = (size_t)(W / 3) * (H / 3);
        const double box_bytes = (double)boxes * V * 9 * 2048;
        const double bn = time_ms([&] { hipLaunchKernelGGL(k_boxes<true>, dim3((unsigned)((boxes + 3) / 4)), dim3(256), 0, 0, buf, V, H, W, sink); }, 3);
        const double bd = time_ms([&] { hipLaunchKernelGGL(k_boxes<false>, dim3((unsigned)((boxes + 3) / 4)), dim3(256), 0, 0, buf, V, H, W, sink); }, 3);
        printf("alloc %d (%d dummies) buf@%p: stream %.0f  rows nt %.0f / plain %.0f  boxes nt %.0f / plain %.0f  GB/s\n", r, nd,
               (void *)buf, bytes / s / 1e6, bytes / rn / 1e6, bytes / rd / 1e6, box_bytes / bn / 1e6, box_bytes / bd / 1e6);
        fflush(stdout);
        free_buf(bb);
        for (void *p : dummies) CHECK(hipFree(p));
    }
    return 0;
}
