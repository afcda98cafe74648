/*
 * projector_oracle.c -- CPU ORACLE (test infrastructure, NOT a product path).
 *
 * A scalar fp32 restatement of the reference's 2D -> sparse-voxel feature
 * projector, written from a reading of the reference sources (paths relative
 * to /root/reference/cuda_project_image_to_sparse_voxel/):
 *
 *   project_image_cuda_kernel.cu:157-187   per-pixel ray set-up
 *   project_image_cuda_kernel.cu:24-92     ray-march, first non-zero ID,
 *                                          count += 1, out[id,:] += feat
 *   project_image_cuda_kernel.cu:390-414   opts / grid_origin unpacking
 *   include/cudaUtil.h:74-119              RayCastParams, kinectProjToCamera*
 *   include/cuda_SimpleMatrixUtil.h:807-812,888-908   row-major float4x4 * v
 *   include/cutil_math.h:896-906,1146-1149,1207-1211   /, dot, normalize (= v * rsqrtf(dot(v,v)))
 *   include/cutil_math.h:81-84             rsqrtf := 1/sqrtf -- the header's HOST fallback (inside #ifndef __CUDACC__);
 *                                          adopted here as the arithmetic contract, see "What bit-exact means" below
 *
 * PARITY STATUS: "parity unpinned" by the reference -- the reference holds no
 * runnable test, golden vector or fixture for this path (SURVEY.md section 4,
 * 8c) and its CUDA sources cannot be built in this image (cuda.h, cufft.h,
 * curand.h absent; no stand-ins are written).  The oracle is pinned instead
 * by hand-derived known-answer tests (tests/test_oracle_kat.py) and by an
 * independently written numpy twin (oracle/numpy_twin.py).  The RGB path
 * (rgb_project below) IS pinned against outputs of the reference's own
 * debug_project_colors.py run in the build container
 * (tests/golden/make_reference_goldens.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Build: oracle/Makefile (gcc, -ffp-contract=off so that
 * every multiply and add rounds separately, as written).
 *
 * Arithmetic contract (the definition of "bit-exact" for the HIP path):
 *   - every operation is an IEEE-754 binary32 operation, rounded to nearest
 *     even, in exactly the order written in the reference source;
 *   - rsqrtf(x) := 1.0f / sqrtf(x)  (the source-level definition the header gives for non-CUDA compilers,
 *     cutil_math.h:81-84);
 *   - no fused multiply-add;
 *   - roundf rounds half away from zero; float -> int conversion saturates
 *     and maps NaN to 0 (the behaviour of both cvt.rzi.s32.f32 and
 *     v_cvt_i32_f32);
 *   - the ray parameter advances by repeated addition t += inc.
 *
 * What bit-exact means -- and does not.  The contract above is the reference SOURCE read as IEEE arithmetic.  The
 * reference's actual CUDA binary is built by nvcc, which (a) replaces rsqrtf by the GPU's approximate reciprocal square
 * root (max error 2 ulp; the :81-84 definition is host-only) and (b) contracts a*b+c into FMA by default.  Neither can
 * be reproduced bit for bit without that compiler and GPU, so parity with the CUDA binary holds up to ray samples that
 * land within rounding distance of a cell or pixel boundary.  To put a number on it this file also builds as a
 * "device-like" variant (-DORACLE_DEVLIKE -ffp-contract=fast, oracle/Makefile): correctly rounded rsqrt nudged by a
 * chosen number of ulps, FMA contraction wherever the compiler finds it.  tests/test_oracle_devlike_cpu.py reports the
 * fraction of pixels whose first-hit voxel changes between the contract and those variants (a few 1e-4 .. 1e-3 on the
 * synthetic scenes); DESIGN.md quotes it as the expected divergence from a real CUDA build.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } f3;

static inline int f2i_sat(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int)v;
}

/* cutil_math.h:1146-1149 */
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

#ifdef ORACLE_DEVLIKE
/* device-like variant: a reciprocal square root within `g_rsqrt_bias` ulps of the correctly rounded one (CUDA's
 * rsqrtf is specified to 2 ulp) */
static int g_rsqrt_bias = 0;
void oracle_set_rsqrt_bias(int ulps) { g_rsqrt_bias = ulps; }
static inline float rsqrt_contract(float x)
{
    float r = (float)(1.0 / sqrt((double)x));
    for (int i = 0; i < g_rsqrt_bias; i++) r = nextafterf(r, INFINITY);
    for (int i = 0; i > g_rsqrt_bias; i--) r = nextafterf(r, -INFINITY);
    return r;
}
int oracle_is_devlike(void) { return 1; }
#else
static inline float rsqrt_contract(float x) { return 1.0f / sqrtf(x); }
int oracle_is_devlike(void) { return 0; }
#endif

/* cutil_math.h:1207-1211 with rsqrtf of cutil_math.h:81-84 */
static inline f3 normalize3(f3 v)
{
    float invLen = rsqrt_contract(dot3(v, v));
    f3 r = { v.x * invLen, v.y * invLen, v.z * invLen };
    return r;
}

/* Ray of pixel (x,y) in view matrix m (row-major camera->world, 16 floats).
 * K.cu:178-187, cudaUtil.h:106-119, cuda_SimpleMatrixUtil.h:888-908. */
typedef struct { f3 camDir, camPos, worldDir; } ray_t;

static inline ray_t make_ray(const float *m, float fx, float fy, float mx, float my,
                             float dmin, float dmax, unsigned ux, unsigned uy)
{
    ray_t r;
    /* kinectProjToCameraZ(dmin,dmax,1.0f): z*(dmax-dmin)+dmin */
    float depth = 1.0f * (dmax - dmin) + dmin;
    /* kinectDepthToSkeleton */
    float sx = ((float)ux - mx) / fx;
    float sy = ((float)uy - my) / fy;
    f3 c = { depth * sx, depth * sy, depth };
    r.camDir = normalize3(c);
    /* float4x4 * float3 (w = 1), v = (0,0,0) */
    r.camPos.x = m[0] * 0.0f + m[1] * 0.0f + m[2] * 0.0f + m[3] * 1.0f;
    r.camPos.y = m[4] * 0.0f + m[5] * 0.0f + m[6] * 0.0f + m[7] * 1.0f;
    r.camPos.z = m[8] * 0.0f + m[9] * 0.0f + m[10] * 0.0f + m[11] * 1.0f;
    /* float4x4 * float4(camDir, 0) */
    f3 w;
    w.x = m[0] * r.camDir.x + m[1] * r.camDir.y + m[2] * r.camDir.z + m[3] * 0.0f;
    w.y = m[4] * r.camDir.x + m[5] * r.camDir.y + m[6] * r.camDir.z + m[7] * 0.0f;
    w.z = m[8] * r.camDir.x + m[9] * r.camDir.y + m[10] * r.camDir.z + m[11] * 0.0f;
    r.worldDir = normalize3(w);
    return r;
}

typedef struct {
    int width, height;          /* from opts, K.cu:403-404 */
    float dmin, dmax, inc;      /* K.cu:405-407 */
    float ox, oy, oz, vs;       /* K.cu:412-414 */
    float fx, fy, mx, my;       /* intrinsics row of this batch, cudaUtil.h:86-93 */
    int dimz, dimy, dimx;
} march_t;

/* K.cu:31-82.  Returns the first non-zero occupancy ID along the ray (after
 * the long -> int truncation of K.cu:70), or 0.  *steps counts loop trips. */
static inline int march(const march_t *p, const int64_t *occ, const ray_t *r, int *steps)
{
    const float depthToRayLength = 1.0f / r->camDir.z;
    float t = depthToRayLength * p->dmin;
    const float tEnd = depthToRayLength * p->dmax;
    int n = 0;
    while (t < tEnd) {
        n++;
        float px = r->camPos.x + t * r->worldDir.x;
        float py = r->camPos.y + t * r->worldDir.y;
        float pz = r->camPos.z + t * r->worldDir.z;
        float sx = (px - p->ox) / p->vs;
        float sy = (py - p->oy) / p->vs;
        float sz = (pz - p->oz) / p->vs;
        int ix = f2i_sat(roundf(sx)), iy = f2i_sat(roundf(sy)), iz = f2i_sat(roundf(sz));
        float cx = r->camDir.x * t, cy = r->camDir.y * t, cz = r->camDir.z * t;
        float u = p->fx * (cx / cz) + p->mx;
        float v = p->fy * (cy / cz) + p->my;
        int in_bounds = (u >= 0 && u < (float)p->width && v >= 0 && v < (float)p->height);
        if (in_bounds && ix >= 0 && iy >= 0 && iz >= 0 && ix < p->dimx && iy < p->dimy && iz < p->dimz) {
            int id = (int)occ[((int64_t)iz * p->dimy + iy) * p->dimx + ix];
            if (id != 0) { if (steps) *steps = n; return id; }
        }
        t += p->inc;
    }
    if (steps) *steps = n;
    return 0;
}

/*
 * Full forward: the behaviour of project_features_cuda(...) with
 * pred_mode = false (K.cu:374-459 + the kernel).  Outputs accumulate (+=),
 * K.cu:77,88.  hit_image (nullable) receives the first-hit ID per
 * (b,v,y,x); out64 (nullable, [n_rows,C]) receives a float64 accumulation of
 * the same rows for tolerance analysis.  Pixel order of the fp32 sums: b, v,
 * y, x ascending (the reference's float atomics have no defined order; this
 * is the canonical one).  Returns 0, or -1 if a hit ID falls outside
 * [1, n_rows) (the reference would write out of bounds).
 */
int oracle_project_features(const float *feats, const int64_t *occ, const float *vmi,
                            const float *intr, const float *opts,
                            const float *grid_origin, float voxel_size,
                            int B, int V, int C, int dimz, int dimy, int dimx,
                            int32_t *count, float *out, int64_t n_rows,
                            int32_t *hit_image, double *out64, int32_t *steps_image,
                            int nthreads)
{
    march_t p;
    p.width = (int)(opts[0] + 0.5f);
    p.height = (int)(opts[1] + 0.5f);
    p.dmin = opts[2]; p.dmax = opts[3]; p.inc = opts[4];
    p.ox = grid_origin[0]; p.oy = grid_origin[1]; p.oz = grid_origin[2];
    p.vs = voxel_size;
    p.dimz = dimz; p.dimy = dimy; p.dimx = dimx;
    const int W = p.width, H = p.height;
    const int64_t npix = (int64_t)B * V * H * W;
    int32_t *hits = hit_image ? hit_image : (int32_t *)malloc(sizeof(int32_t) * (size_t)npix);
    if (!hits) return -2;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    for (int b = 0; b < B; b++) {
        march_t pb = p;
        pb.fx = intr[b * 4 + 0]; pb.fy = intr[b * 4 + 1];
        pb.mx = intr[b * 4 + 2]; pb.my = intr[b * 4 + 3];
        const int64_t *occ_b = occ + (int64_t)b * dimz * dimy * dimx;
        for (int v = 0; v < V; v++) {
            const float *m = vmi + ((int64_t)b * V + v) * 16;
            int32_t *hv = hits + ((int64_t)b * V + v) * H * W;
            int32_t *sv = steps_image ? steps_image + ((int64_t)b * V + v) * H * W : 0;
#pragma omp parallel for schedule(dynamic, 4)
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    ray_t r = make_ray(m, pb.fx, pb.fy, pb.mx, pb.my, pb.dmin, pb.dmax, (unsigned)x, (unsigned)y);
                    int st = 0;
                    hv[(int64_t)y * W + x] = march(&pb, occ_b, &r, &st);
                    if (sv) sv[(int64_t)y * W + x] = st;
                }
        }
    }
    int rc = 0;
    for (int64_t i = 0; i < npix; i++) {
        int id = hits[i];
        if (id == 0) continue;
        if (id < 0 || id >= n_rows) { rc = -1; continue; }
        count[id] += 1;
    }
    /* accumulate: channel slices in parallel, pixels serial => canonical order */
#pragma omp parallel
    {
        int nt = 1, tid = 0;
#ifdef _OPENMP
        nt = omp_get_num_threads(); tid = omp_get_thread_num();
#endif
        int c0 = (int)((int64_t)C * tid / nt), c1 = (int)((int64_t)C * (tid + 1) / nt);
        for (int64_t i = 0; i < npix && c1 > c0; i++) {
            int id = hits[i];
            if (id <= 0 || id >= n_rows) continue;
            const float *f = feats + i * C;
            float *o = out + (int64_t)id * C;
            for (int c = c0; c < c1; c++) o[c] += f[c];
            if (out64) {
                double *o64 = out64 + (int64_t)id * C;
                for (int c = c0; c < c1; c++) o64[c] += (double)f[c];
            }
        }
    }
    if (!hit_image) free(hits);
    return rc;
}

/* First-hit image only (no feature traffic): used by the cpu_baseline leg to
 * time the march separately and by tests that only need IDs. */
int oracle_first_hit(const int64_t *occ, const float *vmi, const float *intr, const float *opts,
                     const float *grid_origin, float voxel_size,
                     int B, int V, int dimz, int dimy, int dimx,
                     int32_t *hit_image, int32_t *steps_image, int nthreads)
{
    march_t p;
    p.width = (int)(opts[0] + 0.5f);
    p.height = (int)(opts[1] + 0.5f);
    p.dmin = opts[2]; p.dmax = opts[3]; p.inc = opts[4];
    p.ox = grid_origin[0]; p.oy = grid_origin[1]; p.oz = grid_origin[2];
    p.vs = voxel_size;
    p.dimz = dimz; p.dimy = dimy; p.dimx = dimx;
    const int W = p.width, H = p.height;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    for (int b = 0; b < B; b++) {
        march_t pb = p;
        pb.fx = intr[b * 4 + 0]; pb.fy = intr[b * 4 + 1];
        pb.mx = intr[b * 4 + 2]; pb.my = intr[b * 4 + 3];
        const int64_t *occ_b = occ + (int64_t)b * dimz * dimy * dimx;
        for (int v = 0; v < V; v++) {
            const float *m = vmi + ((int64_t)b * V + v) * 16;
            int32_t *hv = hit_image + ((int64_t)b * V + v) * H * W;
            int32_t *sv = steps_image ? steps_image + ((int64_t)b * V + v) * H * W : 0;
#pragma omp parallel for schedule(dynamic, 4)
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    ray_t r = make_ray(m, pb.fx, pb.fy, pb.mx, pb.my, pb.dmin, pb.dmax, (unsigned)x, (unsigned)y);
                    int st = 0;
                    hv[(int64_t)y * W + x] = march(&pb, occ_b, &r, &st);
                    if (sv) sv[(int64_t)y * W + x] = st;
                }
        }
    }
    return 0;
}

/* Per-pixel ray set-up exposed for the twin cross-check: out9 = camDir,
 * camPos, worldDir. */
void oracle_ray(const float *m, const float *intr4, float dmin, float dmax, int x, int y, float *out9)
{
    ray_t r = make_ray(m, intr4[0], intr4[1], intr4[2], intr4[3], dmin, dmax, (unsigned)x, (unsigned)y);
    out9[0] = r.camDir.x; out9[1] = r.camDir.y; out9[2] = r.camDir.z;
    out9[3] = r.camPos.x; out9[4] = r.camPos.y; out9[5] = r.camPos.z;
    out9[6] = r.worldDir.x; out9[7] = r.worldDir.y; out9[8] = r.worldDir.z;
}

/*
 * RGB path: debug_project_colors.py:54-81 (voxel-driven, nearest pixel, NO
 * occlusion test, float64 arithmetic as numpy promotes it).
 *   world = grid_origin(f32 -> f64) + voxel_size(f64) * [x,y,z]      :60
 *   cam   = R^T (world - t), R,t from the f32 c2w promoted to f64     :61-63
 *           (np.dot of a 3x3 with a 3-vector: sum in index order)
 *   cam.z > 0                                                         :65
 *   u = fx*(cam.x/cam.z)+cx with fx..cy np.float32 scalars; float32 *
 *       float64 -> float64 under NumPy 2 promotion                    :66-67
 *   u_int = int(round(u)) (Python round: half to even)                :68
 *   0 <= u_int < img_w and 0 <= v_int < img_h                         :69
 *   color = img[v_int,u_int] / 255.0  (float64), stored as float32    :70,75
 * Voxels are visited in (z,y,x) raster order of the dense grid (np.nonzero,
 * :50,58).  Outputs are compacted; returns the number of projected voxels.
 */
int64_t oracle_rgb_project(const int32_t *occ, int dimz, int dimy, int dimx,
                           const float *c2w, const float *intr4,
                           const float *grid_origin, double voxel_size,
                           const uint8_t *img, int img_h, int img_w,
                           float *colors, int32_t *zyx, int32_t *uv)
{
    double R[3][3], t[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[i][j] = (double)c2w[i * 4 + j];
        t[i] = (double)c2w[i * 4 + 3];
    }
    const double fx = intr4[0], fy = intr4[1], cx = intr4[2], cy = intr4[3];
    int64_t n = 0;
    for (int z = 0; z < dimz; z++)
        for (int y = 0; y < dimy; y++)
            for (int x = 0; x < dimx; x++) {
                if (occ[((int64_t)z * dimy + y) * dimx + x] <= 0) continue;
                double w[3] = { (double)grid_origin[0] + voxel_size * (double)x,
                                (double)grid_origin[1] + voxel_size * (double)y,
                                (double)grid_origin[2] + voxel_size * (double)z };
                double d[3] = { w[0] - t[0], w[1] - t[1], w[2] - t[2] };
                double cam[3];
                for (int i = 0; i < 3; i++) /* R^T d: sum over j of R[j][i]*d[j] */
                    cam[i] = R[0][i] * d[0] + R[1][i] * d[1] + R[2][i] * d[2];
                if (!(cam[2] > 0)) continue;
                double u = fx * (cam[0] / cam[2]) + cx;
                double v = fy * (cam[1] / cam[2]) + cy;
                double ur = nearbyint(u), vr = nearbyint(v); /* FE_TONEAREST: half to even */
                if (!(ur >= 0 && ur < (double)img_w && vr >= 0 && vr < (double)img_h)) continue;
                int ui = (int)ur, vi = (int)vr;
                const uint8_t *px = img + ((int64_t)vi * img_w + ui) * 3;
                for (int c = 0; c < 3; c++) colors[n * 3 + c] = (float)((double)px[c] / 255.0);
                zyx[n * 3 + 0] = z; zyx[n * 3 + 1] = y; zyx[n * 3 + 2] = x;
                uv[n * 2 + 0] = ui; uv[n * 2 + 1] = vi;
                n++;
            }
    return n;
}

/*
 * DPF diagnostics: debug_project_features.py:59-84 -- project every occupied
 * voxel centre with the same float64 math as the RGB path (no rounding) and
 * count those with cam.z > 0 (n_front) and those also inside
 * 0 <= u < img_w, 0 <= v < img_h (n_in_bounds); min/max of u and v over the
 * front-facing ones.  stats = {umin, umax, vmin, vmax}.
 */
void oracle_dpf_diagnostics(const int32_t *occ, int dimz, int dimy, int dimx,
                            const float *c2w, const float *intr4,
                            const float *grid_origin, double voxel_size,
                            int img_h, int img_w,
                            int64_t *n_front, int64_t *n_in_bounds, double *stats)
{
    double R[3][3], t[3];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R[i][j] = (double)c2w[i * 4 + j];
        t[i] = (double)c2w[i * 4 + 3];
    }
    const double fx = intr4[0], fy = intr4[1], cx = intr4[2], cy = intr4[3];
    int64_t nf = 0, nb = 0;
    double umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
    for (int z = 0; z < dimz; z++)
        for (int y = 0; y < dimy; y++)
            for (int x = 0; x < dimx; x++) {
                if (occ[((int64_t)z * dimy + y) * dimx + x] <= 0) continue;
                double w[3] = { (double)grid_origin[0] + voxel_size * (double)x,
                                (double)grid_origin[1] + voxel_size * (double)y,
                                (double)grid_origin[2] + voxel_size * (double)z };
                double d[3] = { w[0] - t[0], w[1] - t[1], w[2] - t[2] };
                double cam[3];
                for (int i = 0; i < 3; i++)
                    cam[i] = R[0][i] * d[0] + R[1][i] * d[1] + R[2][i] * d[2];
                if (!(cam[2] > 0)) continue;
                double u = fx * (cam[0] / cam[2]) + cx;
                double v = fy * (cam[1] / cam[2]) + cy;
                nf++;
                if (u < umin) umin = u;
                if (u > umax) umax = u;
                if (v < vmin) vmin = v;
                if (v > vmax) vmax = v;
                if (u >= 0 && u < (double)img_w && v >= 0 && v < (double)img_h) nb++;
            }
    *n_front = nf; *n_in_bounds = nb;
    stats[0] = umin; stats[1] = umax; stats[2] = vmin; stats[3] = vmax;
}

int oracle_abi_version(void) { return 1; }
