// vp_common.h -- host-side plumbing shared by every part of libvoxproj: error text, HIP call checking, optional
// per-kernel timing (HIP events), the per-workspace stream/event state of VP_FLAG_PIPELINE, kernel parameters and
// the workspace layout.  Included by voxproj.hip only (one translation unit).
#pragma once

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define VP_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(VP_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                              \
    } while (0)

// ------------------------------------------------------------------------------------------------
// optional per-kernel timing (HIP events on the launch stream)
// ------------------------------------------------------------------------------------------------
struct Profile {
    std::mutex mu;
    bool on = false;
    std::vector<hipEvent_t> pool;    // event pairs
    std::vector<int> kind;           // per pair: 0 prep, 1 first_hit (+ work list, view table), 2 gather, 3 heavy voxels' own launch (few-view calls)
    size_t used = 0;                 // pairs in use
    // returns the pair index, or -1
    int next(int k)
    {
        if (used * 2 == pool.size()) {
            hipEvent_t e0, e1;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
            pool.push_back(e0); pool.push_back(e1); kind.push_back(0);
        }
        kind[used] = k;
        return (int)used++;
    }
} g_prof;

// RAII-less helper: times [begin, end) of one kernel group on `stream` when profiling is on
struct ProfSpan {
    int idx = -1;
    hipStream_t stream = nullptr;
    void begin(int k, hipStream_t s)
    {
        std::lock_guard<std::mutex> g(g_prof.mu);
        if (!g_prof.on) return;
        idx = g_prof.next(k);
        stream = s;
        if (idx >= 0) (void)hipEventRecord(g_prof.pool[idx * 2], s);
    }
    void end()
    {
        if (idx < 0) return;
        std::lock_guard<std::mutex> g(g_prof.mu);
        (void)hipEventRecord(g_prof.pool[idx * 2 + 1], stream);
        idx = -1;
    }
};

// ------------------------------------------------------------------------------------------------
// side stream + events for VP_FLAG_PIPELINE, one state per workspace pointer
// ------------------------------------------------------------------------------------------------
std::mutex g_pipe_mu;
struct PipeState;
std::vector<std::pair<void *, PipeState *>> g_pipes;
struct PipeState {
    hipStream_t side = nullptr;    // phase 1
    hipEvent_t fh_done[2] = {nullptr, nullptr};      // phase 1 of buffer set q finished (side stream)
    hipEvent_t call_done[2] = {nullptr, nullptr};    // everything of the call that used set q finished (caller's stream)
    bool used[2] = {false, false};
    long long calls = 0;
    int last_q = 0;
};
// Host-side record of the occupancy-derived tables held by a workspace: the shape they were built for, whether the
// workspace also holds the 32-bit copy of the grid they were built from (VP_FLAG_VERIFY_ACCEL), and how often they
// have been built (vp_workspace_table_builds).  Guarded by g_pipe_mu.
struct AccelRecord {
    int B = 0, dimz = 0, dimy = 0, dimx = 0;
    long long n_rows = 0;
    bool copy_valid = false;
    long long builds = 0;
    bool status_init = false;   // the two status blocks have been zeroed once (sticky words start clean)
};
std::vector<std::pair<const void *, AccelRecord>> g_accel;
AccelRecord accel_get(const void *workspace)
{
    std::lock_guard<std::mutex> g(g_pipe_mu);
    for (auto &kv : g_accel)
        if (kv.first == workspace) return kv.second;
    return AccelRecord();
}
void accel_put(const void *workspace, const AccelRecord &r)
{
    std::lock_guard<std::mutex> g(g_pipe_mu);
    for (auto &kv : g_accel)
        if (kv.first == workspace) { kv.second = r; return; }
    g_accel.emplace_back(workspace, r);
}

// offset of the first-hit image written by the last call on each workspace (vp_copy_hit_image)
std::vector<std::pair<const void *, size_t>> g_last_hit;
void remember_hit(const void *workspace, size_t off)
{
    std::lock_guard<std::mutex> g(g_pipe_mu);
    for (auto &kv : g_last_hit)
        if (kv.first == workspace) { kv.second = off; return; }
    g_last_hit.emplace_back(workspace, off);
}
bool recall_hit(const void *workspace, size_t &off)
{
    std::lock_guard<std::mutex> g(g_pipe_mu);
    for (auto &kv : g_last_hit)
        if (kv.first == workspace) { off = kv.second; return true; }
    return false;
}

PipeState *pipe_state(void *workspace, bool create)
{
    std::lock_guard<std::mutex> g(g_pipe_mu);
    for (auto &kv : g_pipes)
        if (kv.first == workspace) return kv.second;
    if (!create) return nullptr;
    PipeState *ps = new PipeState();
    // (a high-priority side stream was measured: no effect on the pipelined step time, so plain streams)
    bool ok = hipStreamCreateWithFlags(&ps->side, hipStreamNonBlocking) == hipSuccess;
    for (int q = 0; q < 2 && ok; q++)
        ok = hipEventCreateWithFlags(&ps->fh_done[q], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ps->call_done[q], hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete ps; return nullptr; }
    g_pipes.emplace_back(workspace, ps);
    return ps;
}

// ------------------------------------------------------------------------------------------------
// Parameters shared by the kernels (by value, like the reference's RayCastParams, cudaUtil.h:74-96)
// ------------------------------------------------------------------------------------------------
struct Params {
    int width, height;        // K.cu:403-404
    float dmin, dmax, inc;    // K.cu:405-407
    float ox, oy, oz, vs;     // K.cu:412-414
    int dimz, dimy, dimx;     // K.cu:395-397
    int B, V, C;
    long long n_rows;
};

// Status block (one 256-byte slot per buffer set).  Words [0, ST_CALL_WORDS) are per call: cleared when a call starts
// on the set, read by vp_workspace_counters.  ST_STICKY_* live in the block of set 0 only, are raised together with
// their per-call twins, survive every later call on the workspace and are cleared by vp_workspace_status alone --
// so an error raised by pipelined call j is still there when the job finally asks, however many calls later.
enum { ST_BADID = 0, ST_BOXMISS = 1, ST_NHEAVY = 2, ST_STUCK = 4, ST_OCCDIFF = 5,
       ST_WORK0 = 16, WORK_CLASSES = 8,        // per-call: number of voxels in each size class of the gather's work list
       ST_CALL_WORDS = 32,
       ST_STICKY_BADID = 62, ST_STICKY_STUCK = 63, ST_WORDS = 64 };

// per (b,v) entry of the view table: world->camera affine map (inverse of the c2w 3x3) + flags
struct ViewEntry {
    float inv[9];   // row-major inverse of the upper-left 3x3 of c2w
    float pos[3];   // camera position (c2w translation)
    float ok;       // 1 if the inverse is usable, else 0 (forces whole-image boxes)
    float pad[3];
};

// ------------------------------------------------------------------------------------------------
// workspace layout (all offsets 256-byte aligned; occupancy-derived tables first so that their
// position does not depend on the image shape -> VP_FLAG_REUSE_ACCEL)
// ------------------------------------------------------------------------------------------------
struct Layout {
    size_t cell_of_id, mask64, near2, dist, dist_tmp;    // occupancy-derived tables (shared)
    size_t occ_copy;                                     // (int)occupancy the tables were built from (VP_FLAG_VERIFY_ACCEL)
    size_t status[2], cnt_call[2], heavy[2], work[2], viewtab[2], hit[2];   // per-call buffers, two sets (VP_FLAG_PIPELINE)
    size_t total;
    int nbx, nby, nbz;
    long long nblk;   // occupancy blocks (4x4x4 cells) per batch
};

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// `capacity` = bytes of the caller's workspace (0 = compute the minimum).  The two per-call buffer sets sit
// at offsets that depend only on (B, n_rows, grid dims, capacity), never on V/H/W, so that consecutive
// pipelined calls of different V on one workspace cannot alias each other's buffers.
Layout make_layout(int B, int V, int H, int W, long long n_rows, int dimz, int dimy, int dimx, size_t capacity = 0)
{
    Layout l;
    size_t off = 0;
    l.nbx = (dimx + 3) / 4; l.nby = (dimy + 3) / 4; l.nbz = (dimz + 3) / 4;
    l.nblk = ((long long)l.nbx * l.nby * l.nbz + 15) & ~15ll;   // padded: per-batch tables stay 16-byte aligned
    // status words of set 0 come first: vp_workspace_status/counters read the head of the workspace
    l.status[0] = off;   off += align256(ST_WORDS * sizeof(int));
    l.status[1] = off;   off += align256(ST_WORDS * sizeof(int));
    l.cell_of_id = off;  off += align256(size_t(B) * size_t(n_rows) * sizeof(int));
    l.mask64 = off;      off += align256(size_t(B) * l.nblk * sizeof(unsigned long long));
    l.near2 = off;       off += align256(size_t(B) * l.nblk * 16);
    l.dist = off;        off += align256(size_t(B) * l.nblk);
    l.dist_tmp = off;    off += align256(size_t(B) * l.nblk);
    l.occ_copy = off;    off += align256(size_t(B) * size_t(dimz) * dimy * dimx * sizeof(int));
    for (int q = 0; q < 2; q++) {
        l.cnt_call[q] = off; off += align256(size_t(n_rows) * sizeof(int));
        l.heavy[q] = off;    off += align256(size_t(n_rows) * sizeof(int));
        l.work[q] = off;     off += align256(size_t(WORK_CLASSES) * size_t(n_rows) * sizeof(int));
    }
    const size_t per_set = align256(size_t(B) * V * sizeof(ViewEntry)) + align256(size_t(B) * V * H * W * sizeof(int));
    size_t half = per_set;
    if (capacity > off + 2 * per_set) half = ((capacity - off) / 2) & ~size_t(255);
    for (int q = 0; q < 2; q++) {
        l.viewtab[q] = off + q * half;
        l.hit[q] = l.viewtab[q] + align256(size_t(B) * V * sizeof(ViewEntry));
    }
    l.total = off + 2 * per_set;
    return l;
}

}  // namespace
