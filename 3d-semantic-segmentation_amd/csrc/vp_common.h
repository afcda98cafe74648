// vp_common.h -- host-side plumbing shared by every part of libvoxproj: error text, HIP call checking, optional
// per-kernel timing (HIP events), the per-workspace state record (streams/events of VP_FLAG_PIPELINE, table shape, options), kernel parameters and
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
// Per-workspace state.  Everything the library remembers about a workspace lives in ONE record: the side stream and
// events of VP_FLAG_PIPELINE, what the occupancy-derived tables in it were built for, the options set on it, where the
// last call left its first-hit image.  The record is found through the workspace's address, but it is not trusted on
// the address alone: every record has a GENERATION number that is also written into the workspace memory itself (the
// header words of status block 0, below), together with a key of the tables' shape, and every call's first kernel
// compares the two.  Memory that was freed without vp_workspace_release and handed out again -- or overwritten by
// anyone -- no longer carries the generation the record expects: a call that trusts the tables (VP_FLAG_REUSE_ACCEL)
// then does no work and raises the sticky ST_STICKY_STALE word (VP_EINVAL at the next status read), a call that
// rebuilds them re-initialises the header and the sticky words first (k_ws_open).
// ------------------------------------------------------------------------------------------------
struct PipeState {
    hipStream_t side = nullptr;    // phase 1
    hipEvent_t fh_done[2] = {nullptr, nullptr};      // phase 1 of buffer set q finished (side stream)
    hipEvent_t call_done[2] = {nullptr, nullptr};    // everything of the call that used set q finished (caller's stream)
    bool used[2] = {false, false};
    long long calls = 0;
    int last_q = 0;
    bool ok = false;
};
struct WsState {
    unsigned gen = 0;                 // this record's generation (also in the workspace header once a call has run)
    // occupancy-derived tables held by the workspace: the shape they were built for, whether the workspace also holds the
    // 32-bit copy of the grid they were built from (VP_FLAG_VERIFY_ACCEL), how often they have been built
    int B = 0, dimz = 0, dimy = 0, dimx = 0;
    long long n_rows = 0;
    bool copy_valid = false;
    long long builds = 0;
    bool opened = false;              // k_ws_open has run for this record (header initialised)
    // Sticky error words of the workspace (ST_STICKY_*): a page of pinned HOST memory that the kernels write through its
    // device mapping and vp_workspace_status reads after synchronising -- no device-to-host copy at the end of a blocking
    // call (round 4: ~8 us of a 0.3-ms drop-in call), and the words belong to the record, not to workspace memory that
    // someone else may have scribbled over
    int *sticky_host = nullptr, *sticky_dev = nullptr;
    // options (vp_workspace_set_option); -1 = the library's default
    long long opt_heavy_t = -1;
    long long opt_march_lds_kb = -1;
    long long opt_row_begin = -1, opt_row_end = -1;     // VP_OPT_ROW_BEGIN / _END: phase 2 gathers IDs in [begin, end)
    long long opt_one_view = -1;                        // VP_OPT_ONE_VIEW_GATHER: 0 = one-view calls through k_gather (A/B arm)
    long long opt_part_px = -1;                         // VP_OPT_PART_PIXELS: pixels per part of a split voxel (-1: default)
    long long opt_one_view_split = -1;                  // VP_OPT_ONE_VIEW_SPLIT: one-view calls cut voxels above this many pixels into parts (0: never)
    // arguments of the last vp_project_features call (VP_FLAG_GATHER_ONLY repeats its phase 2 on another row range)
    int last_B = 0, last_V = 0, last_H = 0, last_W = 0, last_C = 0, last_q = 0;
    bool last_f16 = false, last_ranged = false;
    int last_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // PlanArgs of the last call, as ints
    unsigned split_seq = 0;                             // sequence number of the last blocking one-view call that polled ST_HOST_NSPLIT
    const void *last_feats = nullptr, *last_out = nullptr, *last_count = nullptr, *last_vmi = nullptr;
    // first-hit image of the last call (vp_copy_hit_image)
    bool has_hit = false;
    size_t hit_off = 0;
    PipeState pipe;
};
std::mutex g_ws_mu;
std::unordered_map<const void *, WsState *> g_ws;
unsigned g_next_gen = 0;

// the record of a workspace (created on first use); records live until vp_workspace_release / vp_workspace_create
WsState *ws_state(const void *workspace, bool create)
{
    std::lock_guard<std::mutex> g(g_ws_mu);
    auto it = g_ws.find(workspace);
    if (it != g_ws.end()) return it->second;
    if (!create) return nullptr;
    if (g_next_gen == 0) g_next_gen = (unsigned)std::chrono::steady_clock::now().time_since_epoch().count() | 1u;
    WsState *st = new WsState();
    st->gen = g_next_gen;
    g_next_gen += 2;                  // stays odd: never 0
    g_ws.emplace(workspace, st);
    return st;
}

void pipe_destroy(PipeState &ps)
{
    if (!ps.ok) return;
    (void)hipStreamSynchronize(ps.side);
    (void)hipStreamDestroy(ps.side);
    for (int q = 0; q < 2; q++) {
        (void)hipEventDestroy(ps.fh_done[q]);
        (void)hipEventDestroy(ps.call_done[q]);
    }
    ps = PipeState();
}

void ws_forget(const void *workspace)
{
    WsState *st = nullptr;
    {
        std::lock_guard<std::mutex> g(g_ws_mu);
        auto it = g_ws.find(workspace);
        if (it == g_ws.end()) return;
        st = it->second;
        g_ws.erase(it);
    }
    pipe_destroy(st->pipe);
    if (st->sticky_host) (void)hipHostFree(st->sticky_host);
    delete st;
}

// compute units of the calling thread's current device (cached per device ordinal)
int device_cus()
{
    static std::mutex mu;
    static std::unordered_map<int, int> cus;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    std::lock_guard<std::mutex> g(mu);
    auto it = cus.find(dev);
    if (it != cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cus[dev] = n;
    return n;
}

// the record's page of sticky error words (pinned, mapped into every device's address space), allocated on first use
bool sticky_open(WsState &st)
{
    if (st.sticky_host) return true;
    void *h = nullptr, *d = nullptr;
    // (coherent = fine-grained: a blocking one-view call polls ST_HOST_NSPLIT while its gather runs)
    if (hipHostMalloc(&h, 256, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return false; }
    memset(h, 0, 256);      // ST_WORDS ints (the enum follows below)
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipHostFree(h); return false; }
    st.sticky_host = (int *)h;
    st.sticky_dev = (int *)d;
    return true;
}

bool pipe_open(PipeState &ps)
{
    if (ps.ok) return true;
    // The side stream is created with the device's highest priority -- not for the scheduling (measured: no effect on the
    // pipelined step time) but for its HARDWARE QUEUE.  The runtime multiplexes a process's streams onto a few HSA queues
    // per priority level (GPU_MAX_HW_QUEUES, 4 by default), handing a new stream the least-used one.  A process that has
    // initialised RCCL through torch.distributed first (a pool of 32 streams per level) gets, for a normal-priority side
    // stream, the very queue the caller's stream sits on: the march of call j+1 then queues up BEHIND the gather of call
    // j instead of running under it and a 300-view R2 pass takes 63 ms instead of 54 (profiles/r03_hw_queue_sharing.log;
    // GPU_MAX_HW_QUEUES=1 reproduces it without RCCL).  Queues of another priority level come from another pool.
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) { (void)hipGetLastError(); prio_greatest = 0; }
    bool ok = hipStreamCreateWithPriority(&ps.side, hipStreamNonBlocking, prio_greatest) == hipSuccess;
    for (int q = 0; q < 2 && ok; q++)
        ok = hipEventCreateWithFlags(&ps.fh_done[q], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&ps.call_done[q], hipEventDisableTiming) == hipSuccess;
    ps.ok = ok;
    return ok;
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
// on the set, read by vp_workspace_counters.  ST_STICKY_* index the record's page of pinned host memory (WsState::sticky_*),
// are raised together with their per-call twins, survive every later call on the workspace and are cleared by
// vp_workspace_status alone -- so an error raised by pipelined call j is still there when the job finally asks, however
// many calls later.
// ST_HDR_*: the workspace header (set 0 only): magic, the generation of the record that initialised this memory, the key
// of the tables it holds (0 while none are sealed) -- written by k_ws_open / k_ws_seal, compared by every call's k_zero_call.
enum { ST_BADID = 0, ST_BOXMISS = 1, ST_NHEAVY = 2, ST_ZERO = 3 /* never written */, ST_STUCK = 4, ST_OCCDIFF = 5, ST_STALE = 6,
       ST_HEAVY_T = 7,                         // per-call: the heavy threshold in force (after the part-slot bound), for the counters
       ST_NHIT = 8,                            // per-call: pixels whose ray hit a voxel (one-view calls that size their parts from it: k_worklist's sum of the march's per-wavefront counts)
       ST_PART_T = 9, ST_PART_PX = 10,         // per-call: pixels above which a voxel was cut into parts, pixels per part (k_worklist)
       ST_WORK0 = 16, WORK_CLASSES = 8,        // per-call: number of voxels in each size class of the gather's work list
       ST_NPARTS = 24, ST_NSPLIT = 25,         // per-call, next to the class counts: parts and split voxels planned by k_worklist
       ST_PLAN_WORDS = WORK_CLASSES + 2,       // ST_WORK0 .. ST_NSPLIT: what a work-list run (re)counts
       ST_CALL_WORDS = 32,
       ST_HDR_MAGIC = 56, ST_HDR_GEN = 57, ST_HDR_TABLES = 58,
       ST_HOST_NSPLIT = 60,                    // sticky page only: k_gather_one's note to a blocking one-view call (GatherArgs::host_word)
       ST_STICKY_STALE = 61, ST_STICKY_BADID = 62, ST_STICKY_STUCK = 63, ST_WORDS = 64 };
constexpr unsigned WS_MAGIC = 0x56585033u;   // "VXP3"
#ifdef VP_DIAG
enum { VP_FLAG_DIAG_EVALS = 1 << 20,       // diagnostic build only (make diag): the hit image receives per-ray evaluation counts
       VP_FLAG_DIAG_WAVES = 1 << 21 };     // ... per-wavefront clock stamps and iteration counts (tools/march_waves.py)
#endif

// key of the tables a workspace holds: shape they were built for + how many times this record has built them
inline unsigned tables_key(int B, int dimz, int dimy, int dimx, long long n_rows, long long builds)
{
    unsigned long long h = 0x9E3779B97F4A7C15ull;
    const long long v[6] = {B, dimz, dimy, dimx, n_rows, builds};
    for (long long x : v) { h ^= (unsigned long long)x + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); h *= 0xff51afd7ed558ccdull; }
    const unsigned k = (unsigned)(h ^ (h >> 32));
    return k ? k : 1u;     // 0 means "no tables"
}

// Near field of one 4x4x4 block (blocks within one block of geometry): for each of its 64 cells the Chebyshev distance in
// CELLS to the nearest occupied cell, capped at 7 ("7 or more"), as three bit planes (bit = z%4*16 + y%4*4 + x%4;
// nd = p0 | p1 << 1 | p2 << 2; nd == 0 <=> the cell is occupied), and a copy of the block's entry of the block distance
// field, so that the march fetches everything it needs about a block with ONE 32-byte read.  Round 3 kept two planes
// (distances up to 3): 8.4 of the 13.1 samples a ray evaluated lay within two blocks of the surface, where a bound of 2 or 3
// allows skips of one and three samples only.
struct __attribute__((aligned(32))) NearRec {
    unsigned long long p0, p1, p2;
    unsigned long long dist;      // the block's Chebyshev distance in blocks to the nearest non-empty block (0 .. 255)
};

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
    size_t parts[2], split[2], pmeta[2], prow[2];        // split voxels (vp_gather.h): part items, split list, per-part results
    size_t hitcnt[2];                                    // one-view calls: hit pixels per wavefront of the march (k_worklist adds them up)
    long long n_hitcnt;
    long long slot_cap;                                  // part slots per set
    size_t total;
    int nbx, nby, nbz;
    long long nblk;   // occupancy blocks (4x4x4 cells) per batch
};

inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// `capacity` = bytes of the caller's workspace (0 = compute the minimum).  The two per-call buffer sets sit
// at offsets that depend only on (B, n_rows, grid dims, capacity), never on V/H/W, so that consecutive
// pipelined calls of different V on one workspace cannot alias each other's buffers.
// Part slots of one buffer set (split voxels, vp_gather.h): a voxel above the heavy threshold is summed as P parts, each
// part's C-wide partial row in a slot.  The number of slots bounds how finely a call can be cut: with part_px >= 2*B*V*H*W /
// slots and heavy_t >= part_px the parts of a call can never outnumber the slots (project_impl raises both to that bound).
// 65536 slots -- parts of 2048 pixels for calls of up to 67 M pixels (126 views of 968x548: the 100-108 views a call of fp16 maps
// holds; round 5's 32768 slots forced parts of 3238-3373 pixels on those calls: R2T fp16 25.3 -> 24.9 ms, A1 fp16 14.9 -> 13.9 ms
// per pass, profiles/r06_f16_part_slots.log) --, fewer when the rows are wide (128 MiB of partial rows per set at most) or the
// call is small.
#ifndef VP_MAX_SLOTS
#define VP_MAX_SLOTS 65536
#endif
#ifndef VP_ONE_VIEW_SLOTS
#define VP_ONE_VIEW_SLOTS 8192
#endif
inline long long part_slot_cap(int B, int V, int H, int W, int C)
{
    const long long px2 = 2ll * B * V * (long long)H * W;
    const long long by_bytes = std::max<long long>(1024, ((long long)VP_MAX_SLOTS * 2048) / (std::max(C, 1) * 4ll));
    long long cap = std::max<long long>(64, std::min<long long>(VP_MAX_SLOTS, std::min(px2, by_bytes)));
    // a call of ONE view cuts voxels into parts of 256 pixels by default (128 at the least for a view of 524 k pixels): 8192 slots
    // (16 MiB at C = 512) -- the drop-in module's scratch buffer should not carry 2 x 128 MiB it never touches
    if ((long long)B * V == 1) cap = std::min<long long>(cap, VP_ONE_VIEW_SLOTS);
    return cap;
}

Layout make_layout(int B, int V, int H, int W, int C, long long n_rows, int dimz, int dimy, int dimx, size_t capacity = 0)
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
    l.near2 = off;       off += align256(size_t(B) * l.nblk * sizeof(NearRec));
    l.dist = off;        off += align256(size_t(B) * l.nblk);
    l.dist_tmp = off;    off += align256(size_t(B) * l.nblk);
    l.occ_copy = off;    off += align256(size_t(B) * size_t(dimz) * dimy * dimx * sizeof(int));
    for (int q = 0; q < 2; q++) {
        l.cnt_call[q] = off; off += align256(size_t(n_rows) * sizeof(int));
        l.heavy[q] = off;    off += align256(size_t(n_rows) * sizeof(int));
        l.work[q] = off;     off += align256(size_t(WORK_CLASSES) * size_t(n_rows) * sizeof(int));
    }
    l.slot_cap = part_slot_cap(B, V, H, W, C);
    const size_t sz_view = align256(size_t(B) * V * sizeof(ViewEntry)), sz_hit = align256(size_t(B) * V * H * W * sizeof(int));
    const size_t sz_i4 = align256(size_t(l.slot_cap) * 16), sz_prow = align256(size_t(l.slot_cap) * size_t(C) * sizeof(float));
    // one int per wavefront (8x8 pixel tile) of a ONE-VIEW march: the view's hit total sizes its parts (PlanArgs, vp_gather.h)
    l.n_hitcnt = (long long)B * V == 1 ? (long long)((W + 15) / 16) * ((H + 15) / 16) * 4 : 0;
    const size_t sz_hc = align256(size_t(l.n_hitcnt) * sizeof(int));
    const size_t per_set = sz_view + sz_hit + 3 * sz_i4 + sz_prow + sz_hc;
    size_t half = per_set;
    if (capacity > off + 2 * per_set) half = ((capacity - off) / 2) & ~size_t(255);
    for (int q = 0; q < 2; q++) {
        l.viewtab[q] = off + q * half;
        l.hit[q] = l.viewtab[q] + sz_view;
        l.parts[q] = l.hit[q] + sz_hit;
        l.split[q] = l.parts[q] + sz_i4;
        l.pmeta[q] = l.split[q] + sz_i4;
        l.prow[q] = l.pmeta[q] + sz_i4;
        l.hitcnt[q] = l.prow[q] + sz_prow;
    }
    l.total = off + 2 * per_set;
    return l;
}

}  // namespace
