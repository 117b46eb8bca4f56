// rt_api.hip — the C ABI of include/rt_abi.h: scene upload, the per-level wavefront pipeline, outputs.
//
// The product path is HIP only.  There is no CPU implementation behind these entry points: without a gfx950
// device every call that would compute fails with RT_ERR_NO_DEVICE / RT_ERR_HIP.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rt_internal.h"
#include "rt_kernels.hip.h"
#include "rt_general.hip.h"

using namespace rt;

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[1024] = "";
extern "C" void rt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char* rt_last_error(void) { return g_err; }

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            rt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return RT_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

// ------------------------------------------------------------------------------------------------ static scratch
// rt_canvas_to_ppm_noalloc (the one entry point that allocates nothing) keeps its only scratch — the per-row byte
// counts / offsets of the PPM text, rows + 1 values — in this file-scope device array instead of a grown buffer.  Every
// GPU the library runs on holds its own copy (128 KiB) until the process ends.  In host code the array's NAME is only a
// stand-in for hipGetSymbolAddress: kernels get the per-device address looked up below, never the name.
constexpr int NOALLOC_MAX_ROWS = 16384;
static __device__ __align__(256) long long scratch[NOALLOC_MAX_ROWS + 1];
constexpr int NOALLOC_MAX_DEVICES = 64;
static std::mutex g_scratch_mutex;
static long long* g_scratch_ptr[NOALLOC_MAX_DEVICES];  // [device] -> that device's copy of `scratch`, null until set up

// Set-up, once per device, with that device current (rt_scene_create calls it: a scene is what every caller of the
// entry point must hold, so the look-up — which may load the code object onto the device — never happens inside it).
static int scratch_setup(int device) {
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return RT_OK;  // such a device simply has no no-alloc entry point
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return RT_OK;
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    g_scratch_ptr[device] = static_cast<long long*>(ptr);
    return RT_OK;
}
static long long* scratch_of(int device) {
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    return g_scratch_ptr[device];
}

// ------------------------------------------------------------------------------------------------ scene
struct LevelStore {
    void* block = nullptr;
    long long cap = 0;
    Level L{};
};

struct GrowBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct MultiState;
struct rt_scene {
    int device = 0;
    hipStream_t stream = nullptr;
    DScene ds{};
    std::vector<void*> allocs;
    bool has_csg = false, has_smooth = false, any_transparent = false, has_tlas = false;
    std::vector<int32_t> leaf_kind;  // host copy of the node kinds (argument checks of rt_normal_at / rt_lighting)
    int meshes_built_on_gpu = 0;
    int meshes_shared = 0;
    int scene_class = 0;  // SC_CSG | SC_CULL: which traversal kernel variants this scene runs
    int n_lights = 0;
    std::vector<LevelStore> levels;
    // one block of CNT_TOTAL device counters per recursion level + one for the launch after the last level; the host
    // mirror (pinned) holds the CNT_COUNT plain counters of every block
    unsigned long long* d_counters = nullptr;
    unsigned long long* h_counters = nullptr;
    int counter_blocks = 0;
    long long* h_word = nullptr;  // one page-locked int64 made at creation: read-backs of a single device word (the entry
                                  // point that allocates nothing must not make the runtime stage a pageable copy)
    GrowBuf out_rgb, out_id, out_flags, frame_rgb, frame_id, frame_flags, in_a, in_b, misc_a, misc_b, misc_c;
    std::vector<hipEvent_t> events;  // timing pool of the rt_stats path (events on the launch stream)
    // Overflow fix-up (rays whose tree ran into a bounded per-lane list are traced again by the BIG kernel variants):
    // the HBM arena those variants keep their lists in, the small grid they run with, and a second set of level queues
    // and counters for the re-traced rays (parked here, swapped in by FixScope)
    GrowBuf arena, ovf_index;
    int arena_wgs = 0;
    std::vector<LevelStore> fix_levels;
    unsigned long long* fix_d_counters = nullptr;
    unsigned long long* fix_h_counters = nullptr;
    int fix_counter_blocks = 0;
    bool big = false;  // the launches being enqueued are the BIG variants
    // The enqueue of a whole frame (~35 launches) is captured once into a HIP graph and replayed with one call while
    // everything it bakes in stays the same (camera, shard, depth, output pointers, queue blocks): `graph_key` is the
    // byte image of those.
    // A few instantiated graphs are kept (least recently used goes first): callers that alternate between two output
    // buffers — rt_render_begin's two device frames, a one-process-per-GPU driver's two shard buffers — replay, not recapture.
    struct GraphEntry { std::string key; hipGraphExec_t exec = nullptr; unsigned long long used = 0; };
    std::vector<GraphEntry> graphs;
    unsigned long long graph_clock = 0;
    std::vector<std::string> graph_recent_keys;  // keys of the last few frames that were enqueued plainly (see render_shard_impl)
    bool use_graph = true;
    // rt_render_begin / rt_render_end: two device frames used in turn, the D2H copy of one on `copy_stream` while the
    // next is rendered on `stream`
    hipStream_t copy_stream = nullptr;
    static constexpr int PIPE_LANES = 3;  // render lanes at most: this scene and two twins (rt_scene_opts.render_lanes)
    static constexpr int PIPE_SLOTS = 2 * PIPE_LANES;  // frames in flight at most: one being rendered and one being copied per lane
                                                       // (slot % pipe_lanes = the render lane; 6 is a multiple of 1, 2 and 3)
    GrowBuf pipe_rgb[PIPE_SLOTS], pipe_id[PIPE_SLOTS], pipe_flags[PIPE_SLOTS];
    hipEvent_t pipe_rendered[PIPE_SLOTS] = {}, pipe_copied[PIPE_SLOTS] = {};
    int pipe_head = 0, pipe_pending = 0;  // next slot to use; frames begun and not yet retired (the most recent slots)
    // Round 3: the two frames in flight are RENDERED side by side as well.  Every traversal launch ends in a tail of a few
    // long rays, and a 1024-lane workgroup keeps its compute unit's LDS until its last wavefront is through, so a launch of
    // the same stream cannot start there; a launch of ANOTHER stream's frame can take the compute units that are already
    // free (measured: two frames at once 2.23 ms, one after the other 2.94).  Slot 0 renders on this scene, slot 1 on
    // `twin`: a shallow copy that shares the read-only scene tables and owns everything a frame writes (stream, level
    // queues, counters, stack spill area, arena, graphs); each slot has a host thread that runs the frame's (blocking)
    // render and the copy, rt_render_end waits for the oldest.
    // Both slots render on NON-BLOCKING streams (`pipe_stream` here, the twin's own): a slot's host thread may be capturing
    // its frame graph while the caller's thread uses the legacy default stream (torch does), which is an error for a
    // blocking stream under capture.  What a blocking stream gives implicitly — work the caller enqueued on the default
    // stream before the call is done before the frame is written — is kept by an event recorded on the default stream at
    // the begin (`pipe_ready`), which the slot's stream waits for.
    rt_scene* twins[PIPE_LANES - 1] = {};
    bool is_twin = false;
    int max_lanes = PIPE_LANES;  // rt_scene_opts.render_lanes
    int pipe_lanes = 0;          // lanes in use: fixed at the first begin (1 + the twins that could be made), 0 before
    struct Worker* pipe_worker[PIPE_LANES] = {};
    hipStream_t pipe_stream = nullptr;
    hipEvent_t pipe_ready[PIPE_SLOTS] = {};
    hipStream_t pipe_copy_stream[PIPE_SLOTS] = {};
    bool pipe_host[PIPE_SLOTS] = {};  // the slot's frame in flight goes to host buffers (rt_render_end waits for its copy)
    // a lane's host thread runs one job at a time: before it is given the next frame, the result of the one it has just
    // finished is put aside here for the rt_render_end that retires that frame
    bool pipe_harvested[PIPE_SLOTS] = {};
    int pipe_rc[PIPE_SLOTS] = {};
    std::string pipe_err[PIPE_SLOTS];
    bool fuse_output = true;   // graph path: level 0's k_shade writes the shard / frame (RT_AMD_FUSE_OUTPUT=0: A/B)
    bool fuse_spawn = true;    // k_trace makes the rays of levels 1.. as it fetches them (RT_AMD_FUSE_SPAWN=0: k_spawn launches, A/B)
    bool fuse_raygen = true;   // ... and level 0's from the camera (RT_AMD_FUSE_RAYGEN=0: a k_raygen launch writes them first, A/B)
    const CameraParams* level0_cam = nullptr;  // set while a Camera::render frame with fuse_raygen is being enqueued
    CameraParams* d_cam = nullptr;             // its device copy (k_set_camera), read by level 0's k_trace
    bool spawns = false;       // some leaf's material reflects or refracts: only then can a ray tree be deeper than one level
    // traversal kernels: dynamic LDS (staged BVH nodes + per-lane stacks) and the persistent grid each one gets
    size_t trav_lds_bytes = 0;
    int n_cus = 1;
    std::map<const void*, int> trav_grid;  // kernel -> workgroups that are resident at once on the whole device
    struct MultiState* multi = nullptr;    // rt_scene_create_multi: replicas on other devices, worker threads, RCCL
    // The 4-component path (rt_general.hip.h): `general` = some inverse of the scene is not affine, every call takes that
    // path; otherwise only frames whose camera has no inverse, or a non-affine one, do.  Its slab and counters are made
    // on first use.
    bool general = false;
    GScene gs{};
    GrowBuf g_slab, g_one;
    int g_lanes = 0;
    unsigned long long* g_h_counters = nullptr;  // pinned mirror of gs.counters
};

namespace {

// Timing events that live for one call.
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr, r0 = nullptr, r1 = nullptr;
    EventPair() = default;
    EventPair(const EventPair&) = delete;
    EventPair& operator=(const EventPair&) = delete;
    ~EventPair() {
        for (hipEvent_t e : {e0, e1, r0, r1})
            if (e) (void)hipEventDestroy(e);
    }
};
// A device allocation that lives for one call.
struct DeviceTemp {
    void* p = nullptr;
    DeviceTemp() = default;
    DeviceTemp(const DeviceTemp&) = delete;
    DeviceTemp& operator=(const DeviceTemp&) = delete;
    ~DeviceTemp() { if (p) (void)hipFree(p); }
};

int grow(GrowBuf& b, size_t bytes) {
    if (bytes <= b.bytes) return RT_OK;
    if (b.p) HIP_TRY(hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    size_t want = bytes + bytes / 4 + 256;
    HIP_TRY(hipMalloc(&b.p, want));
    b.bytes = want;
    return RT_OK;
}

template <class T>
int upload(rt_scene* s, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    size_t bytes = v.size() * sizeof(T);
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
    s->allocs.push_back(p);
    if (bytes) HIP_TRY(hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return RT_OK;
}

template <class T>
std::vector<T> vec_of(const T* p, size_t n) {
    return p && n ? std::vector<T>(p, p + n) : std::vector<T>();
}

// rows 0..2 of each 4x4
std::vector<double> rows3(const double* m16, int n) {
    std::vector<double> out(size_t(n) * 12);
    for (int i = 0; i < n; ++i) std::memcpy(&out[size_t(i) * 12], m16 + size_t(i) * 16, 12 * sizeof(double));
    return out;
}

int ensure_level(rt_scene* s, size_t idx, long long cap) {
    if (s->levels.size() <= idx) s->levels.resize(idx + 1);
    LevelStore& ls = s->levels[idx];
    if (ls.cap >= cap && ls.block) return RT_OK;
    if (ls.block) HIP_TRY(hipFree(ls.block));
    ls.block = nullptr;
    ls.cap = 0;
    long long c = cap + cap / 8 + 1024;
    c = (c + 63) & ~63ll;
    if (c > (1ll << 30)) {  // slots and parent tags (2 * ray + kind) are 32-bit
        rt_set_error("a level of %lld rays is beyond the 2^30 a queue can number", c);
        return RT_ERR_UNSUPPORTED;
    }
    const size_t nd = 18, ni = 6;
    const size_t nl = size_t(s->n_lights > 0 ? s->n_lights : 1);
    size_t bytes = size_t(c) * (nd * sizeof(double) + ni * sizeof(int32_t) + 1 + nl);
    HIP_TRY(hipMalloc(&ls.block, bytes));
    double* dp = static_cast<double*>(ls.block);
    double** dfields[nd] = {&ls.L.ox, &ls.L.oy, &ls.L.oz, &ls.L.dx, &ls.L.dy, &ls.L.dz, &ls.L.ht, &ls.L.hu, &ls.L.hv,
                            &ls.L.px, &ls.L.py, &ls.L.pz, &ls.L.nx, &ls.L.ny, &ls.L.nz, &ls.L.cr, &ls.L.cg, &ls.L.cb};
    for (size_t k = 0; k < nd; ++k) *dfields[k] = dp + size_t(c) * k;
    int32_t* ip = reinterpret_cast<int32_t*>(dp + size_t(c) * nd);
    ls.L.hkey = ip;
    ls.L.mode = ip + size_t(c);
    ls.L.child_refl = ip + size_t(c) * 2;
    ls.L.child_refr = ip + size_t(c) * 3;
    ls.L.shadow_q = ip + size_t(c) * 4;
    ls.L.cont_q = ip + size_t(c) * 5;
    ls.L.flags = reinterpret_cast<uint8_t*>(ip + size_t(c) * ni);
    ls.L.flags32 = reinterpret_cast<uint32_t*>(ls.L.flags);
    ls.L.shadowed = ls.L.flags + size_t(c);
    ls.cap = c;
    return RT_OK;
}

// Counter blocks for a frame of `blocks` levels (+ host mirror).
int ensure_counters(rt_scene* s, int blocks) {
    if (blocks <= s->counter_blocks) return RT_OK;
    const int want = std::max(blocks, 16);
    if (s->d_counters) HIP_TRY(hipFree(s->d_counters));
    if (s->h_counters) HIP_TRY(hipHostFree(s->h_counters));
    s->d_counters = nullptr;
    s->h_counters = nullptr;
    s->counter_blocks = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s->d_counters), size_t(want) * CNT_TOTAL * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_counters), size_t(want) * CNT_COUNT * sizeof(unsigned long long)));
    s->counter_blocks = want;
    return RT_OK;
}

inline unsigned blocks_for(long long n) { return unsigned((n + BLOCK - 1) / BLOCK); }

// Device time of bracketed launches by kernel class, with HIP events on the launch stream.  Nothing waits while the
// frame is being enqueued: the pairs are read after the caller has waited for the stream.
struct Timer {
    rt_scene* s;
    hipStream_t st;
    bool on;
    struct Pair { double* acc; size_t e0, e1; };
    std::vector<Pair> pairs;
    size_t used = 0;
    double* cur = nullptr;
    size_t cur_e0 = 0;
    size_t take() {
        if (used == s->events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) { on = false; return 0; }
            s->events.push_back(e);
        }
        return used++;
    }
    void begin(double* a) {
        if (!on) return;
        cur = a;
        cur_e0 = take();
        if (on) (void)hipEventRecord(s->events[cur_e0], st);
    }
    void end() {
        if (!on || !cur) return;
        size_t e1 = take();
        if (!on) return;
        (void)hipEventRecord(s->events[e1], st);
        pairs.push_back({cur, cur_e0, e1});
        cur = nullptr;
    }
    void collect() {  // after the stream has been waited for
        if (!on) return;
        for (const Pair& p : pairs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, s->events[p.e0], s->events[p.e1]) == hipSuccess) *p.acc += double(ms);
        }
        pairs.clear();
    }
};

// Launches a persistent traversal kernel over `n_chunks` chunks of 64 rays (an upper bound is fine: the kernel
// derives its own chunk count).  The grid is what the device can hold at once with this kernel's LDS footprint.
template <class K, class... A>
int launch_trav(rt_scene* s, hipStream_t st, K kernel, long long n_chunks, A... args) {
    const void* key = reinterpret_cast<const void*>(kernel);
    auto it = s->trav_grid.find(key);
    if (it == s->trav_grid.end()) {
        HIP_TRY(hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, int(s->trav_lds_bytes)));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, TBLOCK, s->trav_lds_bytes));
        if (per_cu < 1) { rt_set_error("traversal kernel does not fit a compute unit with %zu bytes of LDS", s->trav_lds_bytes); return RT_ERR_HIP; }
        it = s->trav_grid.emplace(key, per_cu * s->n_cus).first;
    }
    long long grid = std::min<long long>(n_chunks, it->second);
    if (s->big) grid = std::min<long long>(grid, s->arena_wgs);  // one arena slab per lane of this grid
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kernel, dim3(unsigned(grid)), dim3(TBLOCK), s->trav_lds_bytes, st, args...);
    return RT_OK;
}
inline long long chunks_for(long long n) { return (n + 63) / 64; }  // small launches spread over all CUs, a few chunks each

// Kernel variants are compiled per scene class (rt_kernels.hip.h: SC_CSG | SC_CULL).
#define RT_BY_SCENE_CLASS(sc, CALL) \
    switch (sc) {                   \
        case 0: CALL(0); break;     \
        case 1: CALL(1); break;     \
        case 2: CALL(2); break;     \
        case 3: CALL(3); break;     \
        case 10: CALL(10); break;   \
        default: CALL(11); break;   \
    }
// k_trace: the closest hits of level L (n rays, or n_dev's count clamped to n; n == 0: none) and the shadow rays of
// the hits of level P queued in `prev` (null: none); `prev_upper` bounds P's hit count for the grid size.
template <bool COUNT>
int launch_trace(rt_scene* s, hipStream_t st, const Level& L, long long n, const unsigned long long* n_dev, int remaining,
                 unsigned long long* ctr, const Level& P, unsigned long long* prev, long long prev_upper, int from_parent) {
    const long long chunks = chunks_for(n) + (prev ? chunks_for(prev_upper * (s->n_lights > 0 ? s->n_lights : 1)) : 0);
    int rc = RT_OK;
    const CameraParams* cam = s->d_cam;  // from_parent == 2: level 0's rays are made from the camera (k_set_camera) by the launch that traces them
    if (s->big) {  // overflow fix-up: one variant serves every scene class (rt_kernels.hip.h: SC_BIG)
        if (COUNT) { rt_set_error("internal: the BIG variants are not instrumented"); return RT_ERR_BAD_ARG; }
        return launch_trav(s, st, k_trace<SC_CSG | SC_CULL | SC_TLAS | SC_BIG, true, false>, chunks, s->ds, L, n, n_dev, remaining, ctr, P, prev, from_parent, cam);
    }
    if (from_parent == 2) {  // level 0 of a Camera::render frame, rays made from the camera: the RAYGEN variants (never instrumented)
        if (COUNT) { rt_set_error("internal: the RAYGEN variants are not instrumented"); return RT_ERR_BAD_ARG; }
#define RT_CALL(SC)                                                                                                                        \
    rc = s->has_smooth ? launch_trav(s, st, k_trace<SC, true, false, true>, chunks, s->ds, L, n, n_dev, remaining, ctr, P, prev, 0, cam)    \
                       : launch_trav(s, st, k_trace<SC, false, false, true>, chunks, s->ds, L, n, n_dev, remaining, ctr, P, prev, 0, cam)
        RT_BY_SCENE_CLASS(s->scene_class, RT_CALL)
#undef RT_CALL
        return rc;
    }
#define RT_CALL(SC)                                                                                                                      \
    rc = s->has_smooth ? launch_trav(s, st, k_trace<SC, true, COUNT>, chunks, s->ds, L, n, n_dev, remaining, ctr, P, prev, from_parent, cam)    \
                       : launch_trav(s, st, k_trace<SC, false, COUNT>, chunks, s->ds, L, n, n_dev, remaining, ctr, P, prev, from_parent, cam)
    RT_BY_SCENE_CLASS(s->scene_class, RT_CALL)
#undef RT_CALL
    return rc;
}
template <bool COUNT>
int launch_walk(rt_scene* s, hipStream_t st, const Level& L, long long n_upper, unsigned long long* ctr) {
    int rc = RT_OK;
    if (s->big) return launch_trav(s, st, k_walk<SC_CSG | SC_CULL | SC_TLAS | SC_BIG, false>, chunks_for(n_upper), s->ds, L, ctr);
#define RT_CALL(SC) rc = launch_trav(s, st, k_walk<SC, COUNT>, chunks_for(n_upper), s->ds, L, ctr)
    RT_BY_SCENE_CLASS(s->scene_class, RT_CALL)
#undef RT_CALL
    return rc;
}

// k_queue with the tile size that suits the level (n: its ray count, or the capacity that bounds it)
inline void launch_queue(hipStream_t st, const Level& L, long long n, const unsigned long long* n_dev, long long next_cap,
                         unsigned long long* ctr, int32_t* next_tag) {
    // (with a device-side count `n` is the queue's capacity: deep levels then take the large tiles too — measured in graph
    // mode, 20-27 us per launch against 24-36 us with small tiles, whose many empty workgroups cost more than they spread)
    if (n > 1500000) {
        constexpr long long T = (long long)TILE_ITERS * QBLOCK;
        hipLaunchKernelGGL(k_queue<TILE_ITERS>, dim3(unsigned((n + T - 1) / T)), dim3(QBLOCK), 0, st, L, n, n_dev, next_cap, ctr, next_tag);
    } else {
        constexpr long long T = (long long)TILE_ITERS_SMALL * QBLOCK;
        hipLaunchKernelGGL(k_queue<TILE_ITERS_SMALL>, dim3(unsigned((n + T - 1) / T)), dim3(QBLOCK), 0, st, L, n, n_dev, next_cap, ctr, next_tag);
    }
}

inline unsigned long long* counter_block(rt_scene* s, int d) { return s->d_counters + size_t(d) * CNT_TOTAL; }

// Do levels 0..max_depth have queues (so that a whole frame can be enqueued without a host round trip per level)?
bool levels_sized(rt_scene* s, int max_depth) {
    if (max_depth < 0 || s->levels.size() <= size_t(max_depth)) return false;
    for (int d = 0; d <= max_depth; ++d)
        if (!s->levels[size_t(d)].block) return false;
    return true;
}
// after the stream has been waited for: did every level fit its queue?
bool levels_overflowed(rt_scene* s, int max_depth) {
    for (int d = 0; d < max_depth; ++d)
        if ((long long)s->h_counters[size_t(d) * CNT_COUNT + CNT_NEXT] > s->levels[size_t(d) + 1].cap) return true;
    return false;
}

// color_at for every ray already sitting in level 0 (n0 slots): enqueues the whole ray tree and, bottom-up, its shading;
// resolved colours end up in level 0's cr/cg/cb.
//   sized == false (`sync` walk): a host round trip per level reads the level's ray count and sizes the next queue —
//     the first frame of a scene / view, and the fall-back when a queue overflowed.
//   sized == true: every level reads its count from the counter block of the level before it, grids are sized by queue
//     capacity, nothing waits; a level that spawned more rays than the next queue holds wrote nothing out of bounds, and
//     the caller — after waiting for the stream — sees it with levels_overflowed() and repeats the frame with the walk.
// Either way the plain counters of every block are copied to h_counters at the end (stream-ordered).
template <bool COUNT>
int enqueue_levels(rt_scene* s, hipStream_t st, long long n0, int D, bool sync, Timer& tm, rt_stats* stats, const OutArgs* out = nullptr) {
    HIP_TRY(hipMemsetAsync(s->d_counters, 0, size_t(D + 2) * CNT_TOTAL * sizeof(unsigned long long), st));
    const bool shadows = s->n_lights > 0;
    std::vector<long long> count(size_t(D) + 1, 0);  // sync walk: exact ray count of each level
    int last = D;
    long long n = n0;
    for (int d = 0; d <= D; ++d) {
        const int remaining = D - d;
        if (sync && remaining > 0) {
            int rc = ensure_level(s, size_t(d) + 1, 2 * n);
            if (rc != RT_OK) return rc;
        }
        const Level& L = s->levels[size_t(d)].L;
        const Level& N = remaining > 0 ? s->levels[size_t(d) + 1].L : L;
        const Level& P = d > 0 ? s->levels[size_t(d) - 1].L : L;
        const bool exact = d == 0 || sync;
        const long long n_arg = exact ? n : s->levels[size_t(d)].cap;
        const unsigned long long* n_dev = exact ? nullptr : counter_block(s, d - 1) + CNT_NEXT;
        const long long next_cap = remaining > 0 ? s->levels[size_t(d) + 1].cap : 0;
        unsigned long long* ctr = counter_block(s, d);
        unsigned long long* prev = d > 0 && shadows ? counter_block(s, d - 1) : nullptr;
        const long long prev_upper = d > 0 ? (sync ? count[size_t(d) - 1] : (d == 1 ? n0 : s->levels[size_t(d) - 1].cap)) : 0;
        count[size_t(d)] = n_arg;
        tm.begin(stats ? &stats->ms_trace : nullptr);
        // the rays of levels 1.. are made by the k_trace that traces them, from the parent hits (RT_AMD_FUSE_SPAWN=0: by
        // k_spawn after the level before)
        const int from_parent = d > 0 ? (s->fuse_spawn ? 1 : 0) : (s->level0_cam ? 2 : 0);
        { int rc = launch_trace<COUNT>(s, st, L, n_arg, n_dev, remaining, ctr, P, prev, prev_upper, from_parent); if (rc != RT_OK) return rc; }
        tm.end();
        if (stats) stats->n_trace_launches++;
        tm.begin(stats ? &stats->ms_queue : nullptr);
        launch_queue(st, L, n_arg, n_dev, next_cap, ctr, N.hkey);
        tm.end();
        if (s->any_transparent && (remaining > 0 || s->ds.walk_at_zero)) {
            tm.begin(stats ? &stats->ms_walk : nullptr);
            { int rc = launch_walk<COUNT>(s, st, L, n_arg, ctr); if (rc != RT_OK) return rc; }
            tm.end();
            if (stats) stats->n_walk_launches++;
        }
        if (remaining > 0 && !s->fuse_spawn) {  // one lane per slot of the next level: at most 2 per ray of this one
            const long long slots = std::min(2 * n_arg, next_cap);
            tm.begin(stats ? &stats->ms_queue : nullptr);
            hipLaunchKernelGGL(k_spawn, dim3(blocks_for(slots)), dim3(BLOCK), 0, st, L, N, slots, ctr + CNT_NEXT);
            tm.end();
        }
        HIP_TRY(hipGetLastError());
        if (sync) {
            HIP_TRY(hipMemcpyAsync(s->h_counters + size_t(d) * CNT_COUNT, ctr, CNT_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const long long next = (long long)s->h_counters[size_t(d) * CNT_COUNT + CNT_NEXT];
            if (next > next_cap) {
                rt_set_error("internal: level %d spawned %lld rays into a queue of %lld", d, next, next_cap);
                return RT_ERR_OVERFLOW;
            }
            n = next;
            if (n == 0) { last = d; break; }
        }
    }
    if (shadows) {  // the shadow rays of the last level that holds rays: a k_trace launch without a closest-hit part
        const Level& P = s->levels[size_t(last)].L;
        tm.begin(stats ? &stats->ms_trace : nullptr);
        { int rc = launch_trace<COUNT>(s, st, P, 0, nullptr, 0, counter_block(s, last + 1), P, counter_block(s, last), count[size_t(last)], 0); if (rc != RT_OK) return rc; }
        tm.end();
        if (stats) stats->n_trace_launches++;
    }
    tm.begin(stats ? &stats->ms_kshade : nullptr);
    for (int d = last; d >= 0; --d) {
        const Level& L = s->levels[size_t(d)].L;
        const Level& N = d < last ? s->levels[size_t(d) + 1].L : L;
        const bool exact = d == 0 || sync;
        const unsigned long long* n_dev = exact ? nullptr : counter_block(s, d - 1) + CNT_NEXT;
        unsigned long long* ovf = d == 0 ? counter_block(s, 0) + CNT_OVERFLOW : (unsigned long long*)nullptr;
        if (d == 0 && out)  // level 0 writes the shard / frame itself: no k_output launch, no second pass over the colours
            hipLaunchKernelGGL(k_shade<true>, dim3(blocks_for(count[size_t(d)])), dim3(BLOCK), 0, st, s->ds, L, N, count[size_t(d)], n_dev,
                               d < last ? 1 : 0, ovf, *out);
        else
            hipLaunchKernelGGL(k_shade<false>, dim3(blocks_for(count[size_t(d)])), dim3(BLOCK), 0, st, s->ds, L, N, count[size_t(d)], n_dev,
                               d < last ? 1 : 0, ovf, OutArgs{});
    }
    tm.end();
    HIP_TRY(hipGetLastError());
    // one strided copy: the CNT_COUNT plain counters of every block
    HIP_TRY(hipMemcpy2DAsync(s->h_counters, CNT_COUNT * sizeof(unsigned long long), s->d_counters, CNT_TOTAL * sizeof(unsigned long long),
                             CNT_COUNT * sizeof(unsigned long long), size_t(D + 2), hipMemcpyDeviceToHost, st));
    if (sync) {
        // give the levels this frame did not reach a (small) queue too, so that the next frame can be enqueued in one go
        for (int d = last + 1; d <= D; ++d) {
            int rc = ensure_level(s, size_t(d), 64);
            if (rc != RT_OK) return rc;
        }
    }
    return RT_OK;
}

// after the stream has been waited for: ray accounting and work counters out of the counter blocks
void fill_stats(rt_scene* s, int D, rt_stats* stats) {
    if (!stats) return;
    int64_t shaded = 0, walks = 0, refl = 0, refr = 0, tir = 0;
    int levels = 1;
    for (int d = 0; d <= D + 1; ++d) {
        const unsigned long long* c = s->h_counters + size_t(d) * CNT_COUNT;
        refl += (int64_t)c[CNT_REFLECT];
        refr += (int64_t)c[CNT_REFRACT];
        tir += (int64_t)c[CNT_TIR];
        shaded += (int64_t)c[CNT_SHADOW_Q];
        walks += (int64_t)c[CNT_CONTAINER_Q];
        if (d < D && c[CNT_NEXT] > 0) levels = d + 2;
        for (int k = 0; k < RT_WORK_COUNT; ++k) {
            stats->work_trace[k] += (int64_t)c[CNT_W_XFORMS + k];
            stats->work_shadow[k] += (int64_t)c[CNT_WS_XFORMS + k];
            stats->work_container[k] += (int64_t)c[CNT_WC_XFORMS + k];
        }
    }
    stats->rays_reflect = refl;
    stats->rays_refract = refr - tir;
    stats->shaded_nodes = shaded;
    stats->rays_shadow = shaded * s->n_lights;
    stats->rays_container = walks;
    stats->levels = levels;
}

// ---- overflow fix-up ---------------------------------------------------------------------------------------------
// Capacities of the BIG variants' lists by round (CSG pending intersections, n1/n2 / intersect_world list entries) and
// the grid they run with; the arena is lanes x (csg * 32 + list * 16) bytes, about 1.6 GB in either round.  CSG trees may nest
// as deep as the CSG list is long (one mark per open node, kept in the slab too).
struct ArenaRound { int csg_cap, list_cap, wgs; };
constexpr ArenaRound ARENA_ROUNDS[] = {{1024, 4096, 16}, {8192, 32768, 2}};
constexpr int N_ARENA_ROUNDS = int(sizeof ARENA_ROUNDS / sizeof ARENA_ROUNDS[0]);

int ensure_arena(rt_scene* s, int round) {
    const ArenaRound& a = ARENA_ROUNDS[round];
    const long long lane_bytes = (long long)a.csg_cap * 32 + (long long)a.list_cap * 16;
    int rc = grow(s->arena, size_t(lane_bytes) * size_t(a.wgs) * TBLOCK);
    if (rc != RT_OK) return rc;
    s->ds.arena = static_cast<char*>(s->arena.p);
    s->ds.arena_lane_bytes = lane_bytes;
    s->ds.arena_csg_cap = a.csg_cap;
    s->ds.arena_list_cap = a.list_cap;
    s->arena_wgs = a.wgs;
    return RT_OK;
}

// While alive, the fix-up pass's own queues and counters sit in the scene's fields (and the frame's are parked).
struct FixScope {
    rt_scene* s;
    const CameraParams* cam;  // (the fix-up pass's level 0 holds collected rays, whatever the frame's level 0 was made from)
    explicit FixScope(rt_scene* scene) : s(scene), cam(scene->level0_cam) { s->level0_cam = nullptr; swap(); }
    ~FixScope() { swap(); s->level0_cam = cam; }
    void swap() {
        std::swap(s->levels, s->fix_levels);
        std::swap(s->d_counters, s->fix_d_counters);
        std::swap(s->h_counters, s->fix_h_counters);
        std::swap(s->counter_blocks, s->fix_counter_blocks);
    }
};

int enqueue_levels_any(rt_scene* s, hipStream_t st, long long n0, int D, bool sync, bool counters, Timer& tm, rt_stats* stats,
                       const OutArgs* out = nullptr);

// Level 0 (n0 rays, resolved, the stream waited for) holds `n_fix` rays flagged RT_FLAG_LIST_OVERFLOW: trace those
// again with lists that are large enough and patch their colour, hit and flags back into level 0.  With `stats` the ray
// accounting is corrected too: minus what the bounded pass counted for these rays (found by repeating it on them
// alone), plus what the BIG pass counts.
int fix_overflowed_rays(rt_scene* s, hipStream_t st, long long n0, long long n_fix, int D, rt_stats* stats) {
    int rc = grow(s->ovf_index, size_t(n_fix) * sizeof(int32_t));
    if (rc != RT_OK) return rc;
    const Level main0 = s->levels[0].L;
    unsigned long long* cursor = counter_block(s, 0) + CNT_OVERFLOW;  // reused as the append cursor
    HIP_TRY(hipMemsetAsync(cursor, 0, sizeof(unsigned long long), st));
    long long still = n_fix;
    {
        FixScope scope(s);
        if ((rc = ensure_level(s, 0, n_fix)) != RT_OK || (rc = ensure_counters(s, D + 2)) != RT_OK) return rc;
        const Level fix0 = s->levels[0].L;
        hipLaunchKernelGGL(k_collect_overflow, dim3(blocks_for(n0)), dim3(BLOCK), 0, st, main0, n0, cursor, static_cast<int32_t*>(s->ovf_index.p), fix0);
        HIP_TRY(hipGetLastError());
        Timer off{s, st, false};
        rt_stats before, after;
        std::memset(&before, 0, sizeof before);
        if (stats) {  // what the bounded kernels did with these rays
            rc = enqueue_levels_any(s, st, n_fix, D, true, false, off, nullptr);
            if (rc != RT_OK) return rc;
            HIP_TRY(hipStreamSynchronize(st));
            fill_stats(s, D, &before);
        }
        for (int round = 0; round < N_ARENA_ROUNDS && still > 0; ++round) {
            if ((rc = ensure_arena(s, round)) != RT_OK) return rc;
            s->big = true;
            rc = enqueue_levels_any(s, st, n_fix, D, true, false, off, nullptr);
            s->big = false;
            if (rc != RT_OK) return rc;
            HIP_TRY(hipStreamSynchronize(st));
            still = (long long)s->h_counters[CNT_OVERFLOW];
        }
        if (stats) {
            std::memset(&after, 0, sizeof after);
            fill_stats(s, D, &after);
            stats->rays_reflect += after.rays_reflect - before.rays_reflect;
            stats->rays_refract += after.rays_refract - before.rays_refract;
            stats->shaded_nodes += after.shaded_nodes - before.shaded_nodes;
            stats->rays_shadow += after.rays_shadow - before.rays_shadow;
            stats->rays_container += after.rays_container - before.rays_container;
            stats->levels = std::max(stats->levels, after.levels);
        }
        hipLaunchKernelGGL(k_patch_overflow, dim3(blocks_for(n_fix)), dim3(BLOCK), 0, st, main0, s->levels[0].L, n_fix, static_cast<const int32_t*>(s->ovf_index.p));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
    }
    return RT_OK;
}

int enqueue_levels_any(rt_scene* s, hipStream_t st, long long n0, int D, bool sync, bool counters, Timer& tm, rt_stats* stats,
                       const OutArgs* out) {
    return counters ? enqueue_levels<true>(s, st, n0, D, sync, tm, stats, out) : enqueue_levels<false>(s, st, n0, D, sync, tm, stats, out);
}

// color_at for every ray already sitting in level 0 (n0 slots), complete when it returns: the whole tree enqueued at
// once when the queues exist, else (or when one overflowed) the level-by-level walk that sizes them.
// `out` / `wrote`: level 0's shading pass may write the shard / frame itself (as on the graph path); *wrote says whether
// it did — not when a queue overflowed or rays had to be traced again, then the caller's k_output does it.
int run_levels(rt_scene* s, hipStream_t st, long long n0, int D, bool counters, rt_stats* stats, const OutArgs* out = nullptr,
               bool* wrote = nullptr) {
    if (wrote) *wrote = false;
    if (D < 0) D = 0;  // color_at with remaining <= 0 still traces and shades the ray itself (src/world.rs:81-90)
    if (!s->spawns) D = 0;  // no material reflects or refracts: levels 1.. would be empty launches (a third of C2's frame)
    int rc = ensure_counters(s, D + 2);
    if (rc != RT_OK) return rc;
    Timer tm{s, st, stats != nullptr};
    if (levels_sized(s, D)) {
        rc = enqueue_levels_any(s, st, n0, D, false, counters, tm, stats, out);
        if (rc != RT_OK) return rc;
        HIP_TRY(hipStreamSynchronize(st));
        if (!levels_overflowed(s, D)) {
            tm.collect();
            fill_stats(s, D, stats);
            const long long n_fix = (long long)s->h_counters[CNT_OVERFLOW];
            if (n_fix > 0) return fix_overflowed_rays(s, st, n0, n_fix, D, stats);
            if (wrote && out) *wrote = true;
            return RT_OK;
        }
        if (stats) {  // start the accounting over
            const int64_t primary = stats->rays_primary;
            const double raygen = stats->ms_raygen;
            std::memset(stats, 0, sizeof *stats);
            stats->rays_primary = primary;
            stats->ms_raygen = raygen;
        }
        tm.pairs.clear();
        tm.used = 0;
    }
    rc = enqueue_levels_any(s, st, n0, D, true, counters, tm, stats);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    tm.collect();
    fill_stats(s, D, stats);
    const long long n_fix = (long long)s->h_counters[CNT_OVERFLOW];
    return n_fix > 0 ? fix_overflowed_rays(s, st, n0, n_fix, D, stats) : RT_OK;
}

// `general_cam`: callers that can take the 4-component path (or only need the tile geometry) pass a pointer and learn
// whether the camera needs it — no inverse (src/camera.rs:44-46) or a non-affine one; with a null pointer such a camera
// is an error.
int camera_params(const rt_camera_desc* cam, int rank, int n_ranks, CameraParams& c, long long& n_slots, long long& n_valid,
                  bool* general_cam = nullptr) {
    if (!cam) { rt_set_error("null camera"); return RT_ERR_BAD_ARG; }
    if (cam->hsize < 1 || cam->vsize < 1) { rt_set_error("camera size must be positive (the reference underflows at 0, src/camera.rs:65-66)"); return RT_ERR_BAD_ARG; }
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) { rt_set_error("bad shard %d of %d", rank, n_ranks); return RT_ERR_BAD_ARG; }
    const bool affine = cam->inverse[12] == 0. && cam->inverse[13] == 0. && cam->inverse[14] == 0. && cam->inverse[15] == 1.;
    if (general_cam) *general_cam = !cam->invertible || !affine;
    else if (!cam->invertible) {
        rt_set_error("camera transform is not invertible: the reference then traces Ray(ORIGIN, ORIGIN) for every pixel "
                     "(src/camera.rs:44-46), a point used as a direction, which three components cannot carry");
        return RT_ERR_UNSUPPORTED;
    } else if (!affine) {
        rt_set_error("camera inverse is not affine: its rays have origins with w != 1, which three components cannot carry");
        return RT_ERR_UNSUPPORTED;
    }
    std::memcpy(c.inv, cam->inverse, sizeof c.inv);
    c.half_width = cam->half_width;
    c.half_height = cam->half_height;
    c.pixel_size = cam->pixel_size;
    c.hsize = cam->hsize;
    c.vsize = cam->vsize;
    int W = cam->hsize - 1, H = cam->vsize - 1;
    c.tiles_x = (W + 7) / 8;
    c.tiles_y = (H + 7) / 8;
    c.rank = rank;
    c.n_ranks = n_ranks;
    long long tiles = (long long)c.tiles_x * c.tiles_y;
    long long local_tiles = tiles > rank ? (tiles - rank + n_ranks - 1) / n_ranks : 0;
    n_slots = local_tiles * 64;
    n_valid = 0;
    if (n_ranks == 1) n_valid = (long long)W * H;
    else {
        for (long long t = rank; t < tiles; t += n_ranks) {
            int ty = int(t / c.tiles_x), tx = int(t % c.tiles_x);
            int w = std::min(8, W - tx * 8), h = std::min(8, H - ty * 8);
            n_valid += (long long)w * h;
        }
    }
    return RT_OK;
}

// ---- the 4-component path (rt_general.hip.h)
// Slab sizes: lists of 4096 intersections, 96 nested groups / CSG nodes, ray trees 62 deep: 150 KiB per lane.  The slab
// grows with the largest batch a scene has been asked for — one workgroup of 64 lanes per 64 rays, up to 256 workgroups
// (16384 lanes, 2.5 GB) — and is made the first time a scene takes this path.
constexpr int G_LIST_CAP = 4096, G_DEPTH_CAP = 96, G_FRAME_CAP = 64, G_WGS_MAX = 256;
int general_grid(long long n) { return int(std::max<long long>(1, std::min<long long>(G_WGS_MAX, (n + GBLOCK - 1) / GBLOCK))); }
int ensure_general(rt_scene* s, long long n) {
    int rc;
    const int lanes = general_grid(n) * GBLOCK;
    const long long lane_bytes = (g_lane_bytes(G_LIST_CAP, G_DEPTH_CAP, G_FRAME_CAP) + 255) / 256 * 256;
    if (!s->gs.counters) {
        if ((rc = grow(s->g_one, 8 * sizeof(double))) != RT_OK) return rc;
        void* ctr = nullptr;
        HIP_TRY(hipMalloc(&ctr, G_COUNT * sizeof(unsigned long long)));
        s->allocs.push_back(ctr);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->g_h_counters), G_COUNT * sizeof(unsigned long long), hipHostMallocDefault));
        s->gs.counters = static_cast<unsigned long long*>(ctr);
        s->gs.lane_bytes = lane_bytes;
        s->gs.list_cap = G_LIST_CAP;
        s->gs.depth_cap = G_DEPTH_CAP;
        s->gs.frame_cap = G_FRAME_CAP;
    }
    if (lanes > s->g_lanes) {
        if (s->stream) HIP_TRY(hipStreamSynchronize(s->stream));  // nothing of an earlier call may still be using the old slab
        if ((rc = grow(s->g_slab, size_t(lane_bytes) * size_t(lanes))) != RT_OK) { s->g_lanes = 0; s->gs.slab = nullptr; return rc; }
        s->gs.slab = static_cast<char*>(s->g_slab.p);
        s->g_lanes = lanes;
    }
    return RT_OK;
}
int general_counters_begin(rt_scene* s, hipStream_t st) {
    HIP_TRY(hipMemsetAsync(s->gs.counters, 0, G_COUNT * sizeof(unsigned long long), st));
    return RT_OK;
}
// after the stream has been waited for
int general_counters_end(rt_scene* s, hipStream_t st, rt_stats* stats, long long times) {
    HIP_TRY(hipMemcpyAsync(s->g_h_counters, s->gs.counters, G_COUNT * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (stats) {
        stats->rays_reflect = (long long)s->g_h_counters[G_REFLECT] * times;
        stats->rays_refract = (long long)s->g_h_counters[G_REFRACT] * times;
        stats->rays_shadow = (long long)s->g_h_counters[G_SHADOW] * times;
        stats->shaded_nodes = (long long)s->g_h_counters[G_SHADED] * times;
    }
    return RT_OK;
}
GCamera general_camera(const rt_camera_desc* cam) {
    GCamera g;
    std::memcpy(g.inv, cam->inverse, sizeof g.inv);
    g.half_width = cam->half_width;
    g.half_height = cam->half_height;
    g.pixel_size = cam->pixel_size;
    g.invertible = cam->invertible ? 1 : 0;
    return g;
}
// Camera::render for one shard on the 4-component path; same outputs and stats as render_shard_impl below.
int general_render(rt_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, hipStream_t st, double* d_rgb, int32_t* d_id,
                   uint8_t* d_flags, rt_stats* stats, const CameraParams& c, long long n_slots, long long n_valid, bool frame) {
    int rc;
    if (opts->max_depth + 2 > G_FRAME_CAP) { rt_set_error("the 4-component path holds ray trees up to %d deep", G_FRAME_CAP - 2); return RT_ERR_UNSUPPORTED; }
    if ((rc = ensure_general(s, cam->invertible ? n_slots : 1)) != RT_OK || (rc = general_counters_begin(s, st)) != RT_OK) return rc;
    const int depth = opts->max_depth < 0 ? 0 : opts->max_depth;
    const GCamera g = general_camera(cam);
    const auto t0 = std::chrono::steady_clock::now();
    if (!g.invertible) {  // every pixel's ray is Ray(ORIGIN, ORIGIN): traced once
        double* one = static_cast<double*>(s->g_one.p);
        hipLaunchKernelGGL(k_general_one, dim3(1), dim3(GBLOCK), 0, st, s->ds, s->gs, g, depth, one);
        hipLaunchKernelGGL(k_general_fill, dim3(blocks_for(n_slots)), dim3(BLOCK), 0, st, c, n_slots, static_cast<const double*>(one), d_rgb, d_id, d_flags, frame ? 1 : 0);
    } else {
        hipLaunchKernelGGL(k_general_render, dim3(general_grid(n_slots)), dim3(GBLOCK), 0, st, s->ds, s->gs, g, c, n_slots, depth, d_rgb, d_id, d_flags, frame ? 1 : 0);
    }
    HIP_TRY(hipGetLastError());
    if ((rc = general_counters_end(s, st, stats, g.invertible ? 1 : n_valid)) != RT_OK) return rc;
    if (stats) stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

// Set by the host threads of rt_render_begin's two slots: their frames are enqueued launch by launch (with no host wait
// between launches that is worth ~1 % of a frame), never captured — two threads capturing and instantiating graphs at the
// same time while the caller's thread uses the default stream is more than the frames in flight need.
thread_local bool t_plain_enqueue = false;

// Camera::render for one shard, all work ordered on `st`, complete (the host has waited) when it returns.
int render_shard_impl(rt_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, hipStream_t st, double* d_rgb,
                      int32_t* d_id, uint8_t* d_flags, rt_stats* stats, CameraParams& c, long long& n_slots, bool frame = false) {
    long long n_valid = 0;
    bool general_cam = false;
    int rc = camera_params(cam, opts->rank, opts->n_ranks, c, n_slots, n_valid, &general_cam);
    if (rc != RT_OK) return rc;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->rays_primary = n_valid;
    }
    if (n_slots == 0) return RT_OK;
    if (s->general || general_cam) return general_render(s, cam, opts, st, d_rgb, d_id, d_flags, stats, c, n_slots, n_valid, frame);
    const int D = opts->max_depth < 0 || !s->spawns ? 0 : opts->max_depth;  // see run_levels
    if ((rc = ensure_level(s, 0, n_slots)) != RT_OK || (rc = ensure_counters(s, D + 2)) != RT_OK) return rc;
    const bool fused_raygen = s->fuse_raygen && !opts->want_counters;  // (the instrumented variants take their rays from k_raygen)
    if (fused_raygen && !s->d_cam) {
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, sizeof(CameraParams)));
        s->allocs.push_back(p);
        s->d_cam = static_cast<CameraParams*>(p);
    }
    auto raygen = [&]() {
        if (fused_raygen) hipLaunchKernelGGL(k_set_camera, dim3(1), dim3(64), 0, st, c, s->d_cam);
        else hipLaunchKernelGGL(k_raygen, dim3(blocks_for(n_slots)), dim3(BLOCK), 0, st, c, s->levels[0].L, n_slots);
    };
    struct Level0FromCamera {  // level 0's k_trace makes its rays from `c` while this frame is being enqueued
        rt_scene* s;
        Level0FromCamera(rt_scene* scene, const CameraParams* cam, bool on) : s(scene) { s->level0_cam = on ? cam : nullptr; }
        ~Level0FromCamera() { s->level0_cam = nullptr; }
    } level0_from_camera(s, &c, fused_raygen);
    auto output = [&]() {
        hipLaunchKernelGGL(k_output, dim3(blocks_for(n_slots)), dim3(BLOCK), 0, st, s->ds, s->levels[0].L, n_slots, d_rgb, d_id, d_flags, 1, 0, c, frame ? 1 : 0);
    };
    // ---- the plain path (no statistics): the whole frame replayed as one HIP graph while nothing it bakes in changes
    if (!stats && !opts->want_counters && s->use_graph && !t_plain_enqueue && levels_sized(s, D)) {
        std::string key;
        auto put = [&key](const void* p, size_t n) { key.append(static_cast<const char*>(p), n); };
        // the camera by its named fields (`c` is derived from exactly those; the struct's padding bytes are not part of
        // the key: a descriptor rebuilt in uninitialised storage must not force a new capture).  No captured kernel reads
        // a DScene field that changes after creation (arena*, which only the BIG variants read, and those never run captured).
        put(c.inv, sizeof c.inv); put(&c.half_width, sizeof(double)); put(&c.half_height, sizeof(double)); put(&c.pixel_size, sizeof(double));
        put(&c.hsize, sizeof(int32_t)); put(&c.vsize, sizeof(int32_t));
        put(&opts->rank, sizeof opts->rank); put(&opts->n_ranks, sizeof opts->n_ranks);
        put(&D, sizeof D); put(&d_rgb, sizeof d_rgb); put(&d_id, sizeof d_id); put(&d_flags, sizeof d_flags);
        put(&st, sizeof st); put(&frame, sizeof frame); put(&s->d_counters, sizeof s->d_counters);
        for (int d = 0; d <= D; ++d) { put(&s->levels[size_t(d)].block, sizeof(void*)); put(&s->levels[size_t(d)].cap, sizeof(long long)); }
        constexpr size_t GRAPH_SLOTS = 4;
        hipGraphExec_t graph_exec = nullptr;
        for (rt_scene::GraphEntry& g : s->graphs)
            if (g.key == key) { graph_exec = g.exec; g.used = ++s->graph_clock; }
        // A key is captured the SECOND time it is seen: a camera that moves every frame (an animation) never repeats one, and
        // capture + instantiation cost such a frame 5 % for nothing (1.67 against 1.59 ms on the headline scene)
        // (the last few plainly enqueued keys are remembered, not just one: callers alternate between two output buffers)
        bool seen_before = graph_exec != nullptr;
        for (const std::string& k : s->graph_recent_keys) seen_before = seen_before || k == key;
        if (!seen_before) {
            if (s->graph_recent_keys.size() >= GRAPH_SLOTS) s->graph_recent_keys.erase(s->graph_recent_keys.begin());
            s->graph_recent_keys.push_back(key);
        }
        if (seen_before && !graph_exec) {
            if (s->graphs.size() >= GRAPH_SLOTS) {  // drop the least recently used one
                size_t lru = 0;
                for (size_t k = 1; k < s->graphs.size(); ++k)
                    if (s->graphs[k].used < s->graphs[lru].used) lru = k;
                (void)hipGraphExecDestroy(s->graphs[lru].exec);
                s->graphs.erase(s->graphs.begin() + long(lru));
            }
            hipGraph_t graph = nullptr;
            // a stream that cannot capture (it is being captured by the caller, or capture is unavailable) is not an
            // error: the frame is simply enqueued launch by launch
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                Timer off{s, st, false};
                raygen();
                const OutArgs fused{d_rgb, d_id, d_flags, c, frame ? 1 : 0};
                int rcc = enqueue_levels<false>(s, st, n_slots, D, false, off, nullptr, s->fuse_output ? &fused : nullptr);
                if (!s->fuse_output) output();
                hipError_t ee = hipStreamEndCapture(st, &graph);
                if (rcc != RT_OK || ee != hipSuccess || !graph) {
                    if (graph) (void)hipGraphDestroy(graph);
                    (void)hipGetLastError();
                    s->use_graph = false;
                } else {
                    hipError_t ei = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(graph);
                    if (ei != hipSuccess) { graph_exec = nullptr; (void)hipGetLastError(); s->use_graph = false; }
                    else s->graphs.push_back(rt_scene::GraphEntry{key, graph_exec, ++s->graph_clock});
                }
            } else {
                (void)hipGetLastError();
                s->use_graph = false;
            }
        }
        if (graph_exec) {
            HIP_TRY(hipGraphLaunch(graph_exec, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (!levels_overflowed(s, D)) {
                const long long n_fix = (long long)s->h_counters[CNT_OVERFLOW];
                if (n_fix == 0) return RT_OK;
                // rays that ran into a bounded list: traced again with the BIG variants, then the frame is written once more
                if ((rc = fix_overflowed_rays(s, st, n_slots, n_fix, D, nullptr)) != RT_OK) return rc;
                output();
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(st));
                return RT_OK;
            }
            // the view changed and a queue overflowed: the level-by-level walk below grows the queues (and the next
            // frame captures a new graph, its key holds the queue blocks)
            raygen();
            Timer off{s, st, false};
            rc = enqueue_levels<false>(s, st, n_slots, D, true, off, nullptr);
            if (rc != RT_OK) return rc;
            HIP_TRY(hipStreamSynchronize(st));
            if (const long long n_fix = (long long)s->h_counters[CNT_OVERFLOW])
                if ((rc = fix_overflowed_rays(s, st, n_slots, n_fix, D, nullptr)) != RT_OK) return rc;
            output();
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));
            return RT_OK;
        }
    }
    EventPair whole;
    if (stats) {
        HIP_TRY(hipEventCreate(&whole.e0));
        HIP_TRY(hipEventCreate(&whole.e1));
        HIP_TRY(hipEventRecord(whole.e0, st));
        HIP_TRY(hipEventCreate(&whole.r0));
        HIP_TRY(hipEventCreate(&whole.r1));
        HIP_TRY(hipEventRecord(whole.r0, st));
    }
    raygen();
    if (stats) HIP_TRY(hipEventRecord(whole.r1, st));
    // (frames in flight are enqueued plainly: level 0's shading pass writes the frame there as well)
    const OutArgs fused_plain{d_rgb, d_id, d_flags, c, frame ? 1 : 0};
    bool wrote = false;
    rc = run_levels(s, st, n_slots, D, opts->want_counters != 0, stats, t_plain_enqueue && s->fuse_output && !stats ? &fused_plain : nullptr, &wrote);
    if (rc != RT_OK) return rc;
    if (!wrote) output();
    HIP_TRY(hipGetLastError());
    if (stats) {
        HIP_TRY(hipEventRecord(whole.e1, st));
        HIP_TRY(hipEventSynchronize(whole.e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, whole.e0, whole.e1));
        stats->ms_total = ms;
        HIP_TRY(hipEventElapsedTime(&ms, whole.r0, whole.r1));
        stats->ms_raygen = ms;
    } else {
        HIP_TRY(hipStreamSynchronize(st));
    }
    return RT_OK;
}

void wait_for_frames_in_flight(rt_scene* s);
// `pipe_call`: rt_render_begin / rt_render_frame_begin / rt_render_end themselves.  Any other call on a scene with frames in
// flight first waits for their renders (they use the scene's queues); the frames still have to be retired by rt_render_end.
bool check_scene(rt_scene* s, bool pipe_call = false) {
    if (!s) { rt_set_error("null scene handle"); return false; }
    hipError_t e = hipSetDevice(s->device);
    if (e != hipSuccess) { rt_set_error("hipSetDevice(%d): %s", s->device, hipGetErrorString(e)); return false; }
    if (!pipe_call && s->pipe_pending > 0) wait_for_frames_in_flight(s);
    return true;
}


}  // namespace

// ------------------------------------------------------------------------------------------------ multi-GPU
// One process, one host thread per device: the scene is replicated, the image is cut into 8x8
// tiles dealt round-robin to the devices, every device renders its shard into its own HBM, and ONE gather brings the
// (equally sized, padded) shard buffers to device 0 over xGMI — ncclGather of RCCL, nothing is reduced — where
// k_assemble scatters them into the row-major frame.  RCCL is loaded with dlopen the first time a multi-GPU scene is
// created (a single-GPU host never loads it, and a process that already holds a copy — PyTorch bundles one — shares
// it); when it cannot be used (not loadable, or the device list names one device twice, which the tests on a one-GPU
// box do) the gather falls back to hipMemcpyPeerAsync, one copy per peer on the peer's stream.
struct RcclApi {
    typedef int (*comm_init_all_t)(void**, int, const int*);
    typedef int (*comm_destroy_t)(void*);
    typedef int (*group_t)();
    typedef int (*gather_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef const char* (*errstr_t)(int);
    void* lib = nullptr;
    comm_init_all_t comm_init_all = nullptr;
    comm_destroy_t comm_destroy = nullptr;
    group_t group_start = nullptr, group_end = nullptr;
    gather_t gather = nullptr;
    errstr_t errstr = nullptr;
    std::string why;  // why it is not usable
    bool ok() const { return lib != nullptr; }
};
constexpr int NCCL_UINT8 = 1, NCCL_INT32 = 2, NCCL_FLOAT64 = 8;  // ncclDataType_t, rccl.h:459-467

// One entry per value of RT_AMD_RCCL_LIB seen at scene creation ("" = unset): the variable names the one file to try
// instead of the search below (a host that ships RCCL elsewhere; the tests point it at a file that does not exist to
// take the "RCCL cannot be loaded" branch).  A multi-GPU scene keeps the entry it was created with (MultiState::rccl).
static RcclApi& rccl_api() {
    static std::mutex m;
    static std::map<std::string, std::unique_ptr<RcclApi>> apis;
    const char* env = std::getenv("RT_AMD_RCCL_LIB");
    const std::string which = env ? env : "";
    std::lock_guard<std::mutex> lock(m);
    std::unique_ptr<RcclApi>& slot = apis[which];
    if (slot) return *slot;
    slot.reset(new RcclApi([&which] {
        RcclApi a;
        const char* only = which.c_str();
        if (only && *only) {
            a.lib = dlopen(only, RTLD_NOW | RTLD_LOCAL);
        } else {
            const char* names[] = {"librccl.so", "librccl.so.1"};
            for (const char* n : names)  // a copy the process already holds (PyTorch's bundled one has no soname)
                if (!a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (!a.lib) a.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!a.lib) a.lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        }
        if (!a.lib) {
            const char* e = dlerror();  // ONE call: dlerror() hands the message over and clears it
            a.why = std::string(only && *only ? only : "librccl.so.1") + " cannot be loaded: " + (e ? e : "?");
            return a;
        }
        a.comm_init_all = reinterpret_cast<RcclApi::comm_init_all_t>(dlsym(a.lib, "ncclCommInitAll"));
        a.comm_destroy = reinterpret_cast<RcclApi::comm_destroy_t>(dlsym(a.lib, "ncclCommDestroy"));
        a.group_start = reinterpret_cast<RcclApi::group_t>(dlsym(a.lib, "ncclGroupStart"));
        a.group_end = reinterpret_cast<RcclApi::group_t>(dlsym(a.lib, "ncclGroupEnd"));
        a.gather = reinterpret_cast<RcclApi::gather_t>(dlsym(a.lib, "ncclGather"));
        a.errstr = reinterpret_cast<RcclApi::errstr_t>(dlsym(a.lib, "ncclGetErrorString"));
        if (!a.comm_init_all || !a.comm_destroy || !a.group_start || !a.group_end || !a.gather || !a.errstr) {
            a.why = "librccl lacks ncclCommInitAll / ncclGather / ...";
            a.lib = nullptr;
        }
        return a;
    }()));
    return *slot;
}

// A host thread bound to one replica: runs one job at a time, the caller waits for it.
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int rc = RT_OK;
    std::string err;
    Worker() {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(m);
            while (true) {
                cv.wait(lk, [this] { return has_job || quit; });
                if (quit) return;
                has_job = false;
                std::function<int()> f = std::move(job);
                lk.unlock();
                rt_set_error("%s", "");
                const int r = f();
                const std::string e = rt_last_error();
                lk.lock();
                rc = r;
                err = e;
                done = true;
                cv.notify_all();
            }
        });
    }
    void start(std::function<int()> f) {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(f);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return done; });
        return rc;
    }
    ~Worker() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

namespace {
void wait_for_frames_in_flight(rt_scene* s) {
    for (int k = 0; k < rt_scene::PIPE_LANES; ++k)
        if (s->pipe_worker[k]) (void)s->pipe_worker[k]->wait();
}
}  // namespace
static void destroy_pipe_workers(rt_scene* s) {
    for (int k = 0; k < rt_scene::PIPE_LANES; ++k) {
        delete s->pipe_worker[k];
        s->pipe_worker[k] = nullptr;
    }
}

// The scene that renders the second frame in flight (rt_scene::twin): shares every read-only table of `s` by pointer — the
// tables are never written after creation — and owns what a frame writes.  Nothing here is undone on failure except the
// twin itself: the caller then renders both slots on `s`, one after the other, as before.
static int make_twin(rt_scene* s, int index) {
    std::unique_ptr<rt_scene, void (*)(rt_scene*)> t(new rt_scene(), [](rt_scene* x) { rt_scene_destroy(x); });
    t->is_twin = true;
    t->device = s->device;
    t->ds = s->ds;
    t->ds.spill = nullptr;
    t->ds.arena = nullptr;
    t->ds.arena_lane_bytes = 0;
    t->ds.arena_csg_cap = t->ds.arena_list_cap = 0;
    t->gs = s->gs;          // the 4-component path's tables; its counters and slab are made on first use, per scene
    t->gs.counters = nullptr;
    t->gs.slab = nullptr;
    t->has_csg = s->has_csg; t->has_smooth = s->has_smooth; t->any_transparent = s->any_transparent; t->has_tlas = s->has_tlas;
    t->leaf_kind = s->leaf_kind;
    t->meshes_built_on_gpu = s->meshes_built_on_gpu; t->meshes_shared = s->meshes_shared;
    t->scene_class = s->scene_class;
    t->n_lights = s->n_lights;
    t->use_graph = s->use_graph; t->fuse_output = s->fuse_output; t->fuse_spawn = s->fuse_spawn; t->spawns = s->spawns;
    t->fuse_raygen = s->fuse_raygen;
    t->trav_lds_bytes = s->trav_lds_bytes;
    t->n_cus = s->n_cus;
    t->general = s->general;
    if (s->ds.spill) {  // the traversal stacks' spill rows are written by every traversal launch: one area per stream
        size_t bytes = 0;
        void* base = nullptr;
        HIP_TRY(hipMemGetAddressRange(&base, &bytes, s->ds.spill));
        void* sp = nullptr;
        HIP_TRY(hipMalloc(&sp, bytes));
        t->allocs.push_back(sp);
        t->ds.spill = static_cast<int32_t*>(sp);
    }
    HIP_TRY(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    int rc = ensure_counters(t.get(), 16);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&t->h_word), sizeof(long long), hipHostMallocDefault));
    s->twins[index] = t.release();
    return RT_OK;
}

struct MultiState {
    std::vector<rt_scene*> replicas;  // [0] is the scene that owns this state; the others are whole scenes of their own
    std::vector<int> devices;
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<void*> comms;         // ncclComm_t per replica when the gather is RCCL's
    std::vector<hipEvent_t> sent;     // peer-copy gather: replica r's copy has been enqueued
    std::vector<GrowBuf> shard_q;     // per replica: quantised shard (u8 gather)
    GrowBuf g_rgb, g_q, g_id, g_flags, frame_q;  // on the root: gathered shards, u8 frame
    int gather_kind = 0;              // 0: one device, nothing to gather; 1: ncclGather; 2: hipMemcpyPeerAsync
    RcclApi* rccl = nullptr;          // gather_kind == 1: the library the communicators came from
    std::string gather_note;
};

// Runs f(r) on every replica's worker thread (device r current) and waits for all; first error wins.
template <class F>
int on_all_replicas(MultiState* ms, F f) {
    const size_t n = ms->replicas.size();
    for (size_t r = 0; r < n; ++r)
        ms->workers[r]->start([ms, r, f]() -> int {
            hipError_t e = hipSetDevice(ms->devices[r]);
            if (e != hipSuccess) { rt_set_error("hipSetDevice(%d): %s", ms->devices[r], hipGetErrorString(e)); return RT_ERR_HIP; }
            return f(int(r));
        });
    int rc = RT_OK;
    for (size_t r = 0; r < n; ++r) {
        const int rr = ms->workers[r]->wait();
        if (rr != RT_OK && rc == RT_OK) {
            rc = rr;
            rt_set_error("device %d (replica %zu): %s", ms->devices[r], r, ms->workers[r]->err.c_str());
        }
    }
    return rc;
}


// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

namespace {
// rt_scene_opts with every -1 resolved: the field, else its RT_AMD_* variable, else the default
struct Knobs {
    bool cull = true, passive = true, graph = true, device_bvh = false, verbose = false;
    bool share_meshes = true, fuse_output = true, fuse_spawn = true, tlas = true, tlas_instances = true, general = false, walk_bound = true;
    bool fuse_raygen = true;
    int gather = -1;  // -1 automatic, 1 RCCL also for one device, 2 peer copies
    int lanes = rt_scene::PIPE_LANES;  // render lanes of rt_render_begin and its siblings
};
void resolve_knobs(const rt_scene_opts* o, Knobs& k) {
    rt_scene_opts v;
    std::memset(&v, 0xff, sizeof v);  // all -1
    if (o) std::memcpy(&v, o, std::min(sizeof v, size_t(o->struct_size > 0 ? o->struct_size : 0)));
    auto pick = [](int32_t field, const char* env, bool dflt) {
        if (field >= 0) return field != 0;
        if (const char* e = std::getenv(env)) return std::atoi(e) != 0;
        return dflt;
    };
    k.cull = pick(v.cull_ranges, "RT_AMD_CULL", true);
    k.passive = pick(v.passive_triangles, "RT_AMD_PASSIVE_TRIS", true);
    k.graph = pick(v.frame_graph, "RT_AMD_GRAPH", true);
    k.device_bvh = pick(v.device_bvh, "RT_AMD_DEVICE_BVH", false);
    k.verbose = pick(v.verbose, "RT_AMD_VERBOSE", false);
    k.share_meshes = pick(v.share_meshes, "RT_AMD_SHARE_MESHES", true);
    k.fuse_output = pick(v.fuse_output, "RT_AMD_FUSE_OUTPUT", true);
    k.fuse_spawn = pick(v.fuse_spawn, "RT_AMD_FUSE_SPAWN", true);
    k.tlas = pick(v.top_level_tree, "RT_AMD_TLAS", true);
    k.general = pick(v.four_components, "RT_AMD_GENERAL", false);
    k.walk_bound = pick(v.walk_bound, "RT_AMD_WALK_BOUND", true);
    k.fuse_raygen = pick(v.fuse_raygen, "RT_AMD_FUSE_RAYGEN", true);
    {   // 2: the tree, but every object left to the wavefront (no per-lane mesh instances)
        const char* e = std::getenv("RT_AMD_TLAS");
        k.tlas_instances = !(v.top_level_tree == 2 || (v.top_level_tree < 0 && e && std::atoi(e) == 2));
    }
    {
        int n = v.render_lanes;
        if (n < 1)
            if (const char* e = std::getenv("RT_AMD_LANES")) n = std::atoi(e);
        if (n < 1) {
            // The library's choice: a lane only pays while its stream has a hardware queue to itself.  The HIP runtime spreads a
            // process's streams over GPU_MAX_HW_QUEUES queues (4 unless the variable says otherwise) and streams that share one
            // run one after the other: the caller's default stream, the scene's own and two lanes fill four; measured with a
            // third lane, rank 0's shard of an 8-way split: 0.52 ms per shard at 4 queues (two lanes: 0.42), 0.32 at 8.
            const char* q = std::getenv("GPU_MAX_HW_QUEUES");
            n = (q && std::atoi(q) >= 6) ? rt_scene::PIPE_LANES : 2;
        }
        k.lanes = std::min(n, int(rt_scene::PIPE_LANES));
    }
    k.gather = v.gather == 1 || v.gather == 2 ? v.gather : -1;
    if (v.gather < 0)
        if (const char* e = std::getenv("RT_AMD_GATHER")) k.gather = std::strcmp(e, "rccl") == 0 ? 1 : std::strcmp(e, "peer") == 0 ? 2 : -1;
}
int scene_create_one(const rt_scene_desc* desc, int device, const Knobs& knobs, rt_scene** out);
}  // namespace

int rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out) {
    if (!desc || !out) { rt_set_error("null argument"); return RT_ERR_BAD_ARG; }
    Knobs knobs;
    resolve_knobs(nullptr, knobs);
    return scene_create_one(desc, device, knobs, out);
}

namespace {
int scene_create_one(const rt_scene_desc* desc, int device, const Knobs& knobs, rt_scene** out) {
    *out = nullptr;
    Program prog;
    std::string err;
    const int ndev0 = rt_device_count();
    ProgramOpts popts;
    popts.build_device = device >= 0 && device < ndev0 ? device : -1;
    popts.device_bvh = knobs.device_bvh;
    popts.cull_ranges = knobs.cull;
    popts.share_meshes = knobs.share_meshes;
    popts.top_level_tree = knobs.tlas;
    popts.tlas_instances = knobs.tlas_instances;
    if (const char* e = std::getenv("RT_AMD_TLAS_MIN")) popts.tlas_min_objects = std::atoi(e);  // measurement aid only
    popts.verbose = knobs.verbose;
    const auto t_create0 = std::chrono::steady_clock::now();
    int rc = compile_program(*desc, prog, err, popts);
    if (rc != RT_OK) { rt_set_error("%s", err.c_str()); return rc; }
    int ndev = rt_device_count();
    if (ndev <= 0) { rt_set_error("no HIP device: this library has no CPU path"); return RT_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { rt_set_error("device %d out of range (have %d)", device, ndev); return RT_ERR_NO_DEVICE; }
    HIP_TRY(hipSetDevice(device));
    if ((rc = scratch_setup(device)) != RT_OK) return rc;
    rt_scene* s = new rt_scene();
    s->device = device;
    const rt_scene_desc& d = *desc;
    auto fail = [&](int code) { rt_scene_destroy(s); return code; };
#define UP(expr) do { int rc_ = (expr); if (rc_ != RT_OK) return fail(rc_); } while (0)
    UP(upload(s, vec_of(d.node_kind, size_t(d.n_nodes)), &s->ds.node_kind));
    UP(upload(s, vec_of(d.node_object_id, size_t(d.n_nodes)), &s->ds.node_object_id));
    UP(upload(s, vec_of(d.node_material, size_t(d.n_nodes)), &s->ds.node_material));
    UP(upload(s, vec_of(d.node_eq_class, size_t(d.n_nodes)), &s->ds.node_eq_class));
    UP(upload(s, vec_of(d.node_param, size_t(d.n_nodes)), &s->ds.node_param));
    UP(upload(s, prog.node_nchain, &s->ds.node_nchain));
    {
        std::vector<NodeRec> recs(size_t(d.n_nodes));
        auto near_zero = [](double v) { return std::fabs(v - 0.) < EPS; };  // near_eq(v, 0.) as the kernels spell it
        for (size_t n = 0; n < recs.size(); ++n) {
            int bits = 0;
            const int32_t mi = d.node_material[n];
            if (mi >= 0 && d.node_kind[n] <= RT_SMOOTH_TRIANGLE) {
                const double rf = d.material_params[size_t(mi) * RT_MAT_STRIDE + RT_MAT_REFLECTIVE];
                const double tr = d.material_params[size_t(mi) * RT_MAT_STRIDE + RT_MAT_TRANSPARENCY];
                if (!near_zero(rf)) bits |= NODE_REFLECTS;
                if (!near_zero(tr)) bits |= NODE_REFRACTS;
                if (!near_zero(tr) || (rf > 0. && tr > 0.)) bits |= NODE_WALKS;
            }
            if (bits) s->spawns = true;  // any of NODE_REFLECTS, NODE_REFRACTS, NODE_WALKS
            recs[n] = NodeRec{d.node_kind[n] | (bits << 8), d.node_param[n], prog.node_nchain[n], d.node_material[n]};
        }
        UP(upload(s, recs, &s->ds.node_rec));
    }
    UP(upload(s, vec_of(d.node_casts_shadow, size_t(d.n_nodes)), &s->ds.node_casts_shadow));
    UP(upload(s, rows3(d.xform_inverse, d.n_xforms), &s->ds.xf));
    UP(upload(s, vec_of(d.group_box, size_t(d.n_groups) * 6), &s->ds.group_box));
    {
        std::vector<double> conic(size_t(d.n_conics) * 2);
        for (int i = 0; i < d.n_conics; ++i) { conic[2 * size_t(i)] = d.conic_minimum[i]; conic[2 * size_t(i) + 1] = d.conic_maximum[i]; }
        UP(upload(s, conic, &s->ds.conic));
        UP(upload(s, vec_of(d.conic_closed, size_t(d.n_conics)), &s->ds.conic_closed));
    }
    {
        std::vector<double> tri(size_t(d.n_triangles) * 12);
        for (int i = 0; i < d.n_triangles; ++i)
            for (int k = 0; k < 3; ++k) {
                tri[size_t(i) * 12 + k] = d.tri_p1[3 * size_t(i) + k];
                tri[size_t(i) * 12 + 3 + k] = d.tri_e1[3 * size_t(i) + k];
                tri[size_t(i) * 12 + 6 + k] = d.tri_e2[3 * size_t(i) + k];
                tri[size_t(i) * 12 + 9 + k] = d.tri_normal[3 * size_t(i) + k];
            }
        UP(upload(s, tri, &s->ds.tri));
        UP(upload(s, vec_of(d.tri_smooth, size_t(d.n_triangles)), &s->ds.tri_smooth));
        std::vector<double> sm(size_t(d.n_smooth) * 9);
        for (int i = 0; i < d.n_smooth; ++i)
            for (int k = 0; k < 3; ++k) {
                sm[size_t(i) * 9 + k] = d.smooth_n1[3 * size_t(i) + k];
                sm[size_t(i) * 9 + 3 + k] = d.smooth_n2[3 * size_t(i) + k];
                sm[size_t(i) * 9 + 6 + k] = d.smooth_n3[3 * size_t(i) + k];
            }
        UP(upload(s, sm, &s->ds.smooth));
    }
    UP(upload(s, vec_of(d.material_params, size_t(d.n_materials) * RT_MAT_STRIDE), &s->ds.mat));
    UP(upload(s, vec_of(d.material_pattern, size_t(d.n_materials)), &s->ds.mat_pattern));
    UP(upload(s, vec_of(d.pattern_kind, size_t(d.n_patterns)), &s->ds.pat_kind));
    {
        std::vector<double> ab(size_t(d.n_patterns) * 6);
        for (int i = 0; i < d.n_patterns; ++i)
            for (int k = 0; k < 3; ++k) { ab[size_t(i) * 6 + k] = d.pattern_a[3 * size_t(i) + k]; ab[size_t(i) * 6 + 3 + k] = d.pattern_b[3 * size_t(i) + k]; }
        UP(upload(s, ab, &s->ds.pat_ab));
        UP(upload(s, rows3(d.pattern_inverse, d.n_patterns), &s->ds.pat_inv));
        std::vector<double> li(size_t(d.n_lights) * 6);
        for (int i = 0; i < d.n_lights; ++i)
            for (int k = 0; k < 3; ++k) { li[size_t(i) * 6 + k] = d.light_position[3 * size_t(i) + k]; li[size_t(i) * 6 + 3 + k] = d.light_intensity[3 * size_t(i) + k]; }
        UP(upload(s, li, &s->ds.light));
    }
    // what the 4-component path reads (rt_general.hip.h): the descriptor's tree tables and full inverses as they came in
    UP(upload(s, vec_of(d.xform_inverse, size_t(d.n_xforms) * 16), &s->gs.xf16));
    UP(upload(s, vec_of(d.pattern_inverse, size_t(d.n_patterns) * 16), &s->gs.pat16));
    UP(upload(s, vec_of(d.node_xform, size_t(d.n_nodes)), &s->gs.node_xform));
    UP(upload(s, vec_of(d.node_subtree_end, size_t(d.n_nodes)), &s->gs.node_end));
    UP(upload(s, vec_of(d.node_ancestor, size_t(d.n_nodes)), &s->gs.node_ancestor));
    UP(upload(s, vec_of(d.ancestor_parent, size_t(d.n_ancestors)), &s->gs.anc_parent));
    UP(upload(s, vec_of(d.ancestor_xform, size_t(d.n_ancestors)), &s->gs.anc_xform));
    UP(upload(s, vec_of(d.csg_op, size_t(d.n_csg)), &s->gs.csg_op));
    s->general = prog.general || knobs.general;
    UP(upload(s, prog.items, &s->ds.items));
    UP(upload(s, prog.chain_pool, &s->ds.chain_pool));
    UP(upload(s, prog.bvh, &s->ds.bvh));
    UP(upload(s, prog.mesh_tris, &s->ds.mesh_tris));
    UP(upload(s, prog.mesh_dups, &s->ds.mesh_dups));
    UP(upload(s, prog.mesh_box, &s->ds.mesh_box));
#undef UP
    s->ds.n_lights = d.n_lights;
    s->ds.n_nodes = d.n_nodes;
    s->leaf_kind = vec_of(d.node_kind, size_t(d.n_nodes));
    s->ds.n_items = int32_t(prog.items.size());
    s->ds.all_cast_shadow = prog.all_cast_shadow ? 1 : 0;
    s->ds.cull_reach = prog.cull_reach;
    s->ds.cull_min_dir2 = prog.cull_min_dir2;
    s->ds.far_a = prog.far_a;
    s->ds.far_b = prog.far_b;
    s->ds.far_extent = prog.far_extent;
    s->ds.tlas_root = prog.tlas_root;
    s->ds.n_linear = prog.n_linear;
    s->ds.n_lds = int32_t(std::min<size_t>(prog.bvh.size(), size_t(LDS_NODE_CAP)));
    s->trav_lds_bytes = prog.bvh.empty() ? 0 : size_t(s->ds.n_lds) * sizeof(BvhNode) + size_t(LDS_STACK) * TBLOCK * sizeof(int);
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) != hipSuccess) { rt_set_error("hipGetDeviceProperties failed"); return fail(RT_ERR_HIP); }
        s->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
        // stack rows beyond LDS_STACK, one column per lane that can be resident (2048 lanes per CU at most)
        s->ds.spill_lanes = (long long)s->n_cus * 2048;
        void* sp = nullptr;
        // as many rows as this scene's deepest tree can ask for (3 pending siblings per BVH4 level; host-built trees: <= 31)
        const int spill_rows = std::max(1, 3 * int(prog.bvh_max_depth + prog.tlas_levels) + 1 - LDS_STACK);
        if (hipMalloc(&sp, size_t(spill_rows) * size_t(s->ds.spill_lanes) * sizeof(int32_t)) != hipSuccess) {
            rt_set_error("could not allocate the traversal stack spill area");
            return fail(RT_ERR_HIP);
        }
        s->allocs.push_back(sp);
        s->ds.spill = static_cast<int32_t*>(sp);
    }
    for (int m = 0; m < d.n_materials; ++m) {  // see needs_walk (rt_kernels.hip.h)
        double ri = d.material_params[size_t(m) * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
        if (!(std::isfinite(ri) && ri > 0.)) s->ds.walk_at_zero = 1;
    }
    // F14: a leaf that is not equal to itself (an unbounded cylinder or cone: near_eq(inf, inf) is false) is never found
    // by the n1/n2 walk, which leaves n1 = n2 = 0 and makes schlick 0/0 = NaN — and NaN * BLACK is NaN, not BLACK
    for (int n = 0; n < d.n_nodes; ++n)
        if (d.node_kind[n] != RT_GROUP && d.node_kind[n] != RT_CSG && d.node_eq_class[n] < 0) s->ds.walk_at_zero = 1;
    {   // WalkSink's shortcut (rt_kernels.hip.h) needs every mesh triangle to be equal to no other node of the scene: its
        // equality class (>= 0) is carried by that node alone, or it is not even equal to itself (-1)
        std::map<int32_t, int> uses;
        for (int n = 0; n < d.n_nodes; ++n)
            if (d.node_kind[n] <= RT_SMOOTH_TRIANGLE && d.node_eq_class[n] >= 0) uses[d.node_eq_class[n]]++;
        bool unique = true;
        for (int32_t n : prog.mesh_nodes) {  // every triangle of every mesh, shared trees' instances and repeats included
            const int32_t c = d.node_eq_class[n];
            if (c >= 0 && uses[c] != 1) unique = false;
        }
        s->ds.passive_mesh_tris = unique && knobs.passive ? 1 : 0;
        // ... and the walk may bound its mesh searches (WalkSink::before_mesh) when, besides, every mesh triangle has the same
        // refractive index, bit for bit
        s->ds.walk_bound_item = -1;
        s->ds.walk_passive_ior = 1.;
        if (s->ds.passive_mesh_tris && knobs.walk_bound && prog.walk_bound_item >= 0 && !prog.mesh_nodes.empty()) {
            bool same = true;
            uint64_t bits0 = 0;
            for (size_t q = 0; q < prog.mesh_nodes.size() && same; ++q) {
                const double ior = d.material_params[size_t(d.node_material[prog.mesh_nodes[q]]) * RT_MAT_STRIDE + RT_MAT_REFRACTIVE_INDEX];
                uint64_t bits;
                std::memcpy(&bits, &ior, 8);
                if (q == 0) { bits0 = bits; s->ds.walk_passive_ior = ior; }
                same = bits == bits0;
            }
            if (same) s->ds.walk_bound_item = prog.walk_bound_item;
        }
    }
    if (knobs.verbose)
        std::fprintf(stderr, "rt_scene_create: %.3f s so far (compile + uploads)\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_create0).count());
    if (knobs.verbose)
        std::fprintf(stderr, "rt_scene_create: %d nodes, %zu program items (%d walked by every ray, top-level tree: %s, %d levels), %zu BVH4 nodes (%d in LDS, %zu bytes of LDS per workgroup of %d lanes), %zu mesh triangles, scene class %d\n",
                     d.n_nodes, prog.items.size(), prog.n_linear, prog.tlas_root >= 0 ? "yes" : "no", prog.tlas_levels, prog.bvh.size(), s->ds.n_lds,
                     s->trav_lds_bytes, TBLOCK, prog.mesh_tris.size(), (prog.has_csg ? SC_CSG : 0) | (prog.has_cull ? SC_CULL : 0));
    s->meshes_built_on_gpu = prog.meshes_built_on_gpu;
    s->meshes_shared = prog.meshes_shared;
    s->has_csg = prog.has_csg;
    s->scene_class = (prog.has_csg ? SC_CSG : 0) | (prog.has_cull ? SC_CULL : 0) | (prog.tlas_root >= 0 ? SC_TLAS : 0);
    s->has_tlas = prog.tlas_root >= 0;
    s->has_smooth = prog.has_smooth;
    s->n_lights = d.n_lights;
    for (int m = 0; m < d.n_materials; ++m) {  // does any material ever need the n1/n2 walk?
        const double* mp = d.material_params + size_t(m) * RT_MAT_STRIDE;
        double tr = mp[RT_MAT_TRANSPARENCY], rf = mp[RT_MAT_REFLECTIVE];
        if (!(std::fabs(tr - 0.) < EPS) || (rf > 0. && tr > 0.)) s->any_transparent = true;
    }
    if (hipStreamCreate(&s->stream) != hipSuccess) {
        rt_set_error("could not create the scene's stream: %s", hipGetErrorString(hipGetLastError()));
        return fail(RT_ERR_HIP);
    }
    if (ensure_counters(s, 16) != RT_OK) return fail(RT_ERR_HIP);
    if (hipHostMalloc(reinterpret_cast<void**>(&s->h_word), sizeof(long long), hipHostMallocDefault) != hipSuccess) {
        rt_set_error("could not allocate the scene's page-locked read-back word");
        return fail(RT_ERR_HIP);
    }
    s->use_graph = knobs.graph;
    s->max_lanes = knobs.lanes;
    s->fuse_raygen = knobs.fuse_raygen;
    s->fuse_output = knobs.fuse_output;
    s->fuse_spawn = knobs.fuse_spawn;
    *out = s;
    return RT_OK;
}
}  // namespace

int rt_host_alloc(size_t bytes, void** out) {
    if (!out) { rt_set_error("null argument"); return RT_ERR_BAD_ARG; }
    *out = nullptr;
    if (rt_device_count() <= 0) { rt_set_error("no HIP device: this library has no CPU path"); return RT_ERR_NO_DEVICE; }
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return RT_OK;
}

void rt_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    if (MultiState* ms = s->multi) {  // the owner of a multi-GPU scene takes its replicas with it
        s->multi = nullptr;
        ms->workers.clear();  // joins the threads
        if (ms->gather_kind == 1)
            for (void* c : ms->comms)
                if (c && ms->rccl) (void)ms->rccl->comm_destroy(c);
        for (size_t r = 0; r < ms->replicas.size(); ++r) {
            (void)hipSetDevice(ms->devices[r]);
            if (r < ms->sent.size() && ms->sent[r]) (void)hipEventDestroy(ms->sent[r]);
            if (r < ms->shard_q.size() && ms->shard_q[r].p) (void)hipFree(ms->shard_q[r].p);
            if (r > 0) rt_scene_destroy(ms->replicas[r]);
        }
        (void)hipSetDevice(s->device);
        for (GrowBuf* b : {&ms->g_rgb, &ms->g_q, &ms->g_id, &ms->g_flags, &ms->frame_q})
            if (b->p) (void)hipFree(b->p);
        delete ms;
    }
    (void)hipSetDevice(s->device);
    destroy_pipe_workers(s);  // joins the threads: nothing of a frame in flight is left running
    for (rt_scene*& t : s->twins)
        if (t) { rt_scene_destroy(t); t = nullptr; }
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (void* p : s->allocs) (void)hipFree(p);
    for (LevelStore& l : s->levels)
        if (l.block) (void)hipFree(l.block);
    for (rt_scene::GraphEntry& g : s->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    for (int k = 0; k < rt_scene::PIPE_SLOTS; ++k) {
        if (s->pipe_copied[k]) { (void)hipEventSynchronize(s->pipe_copied[k]); (void)hipEventDestroy(s->pipe_copied[k]); }
        if (s->pipe_rendered[k]) (void)hipEventDestroy(s->pipe_rendered[k]);
        for (GrowBuf* b : {&s->pipe_rgb[k], &s->pipe_id[k], &s->pipe_flags[k]})
            if (b->p) (void)hipFree(b->p);
    }
    if (s->copy_stream) { (void)hipStreamSynchronize(s->copy_stream); (void)hipStreamDestroy(s->copy_stream); }
    if (s->pipe_stream) { (void)hipStreamSynchronize(s->pipe_stream); (void)hipStreamDestroy(s->pipe_stream); }
    for (int k = 0; k < rt_scene::PIPE_SLOTS; ++k) {
        if (s->pipe_ready[k]) (void)hipEventDestroy(s->pipe_ready[k]);
        if (s->pipe_copy_stream[k]) { (void)hipStreamSynchronize(s->pipe_copy_stream[k]); (void)hipStreamDestroy(s->pipe_copy_stream[k]); }
    }
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    for (LevelStore& l : s->fix_levels)
        if (l.block) (void)hipFree(l.block);
    if (s->fix_d_counters) (void)hipFree(s->fix_d_counters);
    if (s->fix_h_counters) (void)hipHostFree(s->fix_h_counters);
    if (s->arena.p) (void)hipFree(s->arena.p);
    if (s->g_slab.p) (void)hipFree(s->g_slab.p);
    if (s->g_one.p) (void)hipFree(s->g_one.p);
    if (s->g_h_counters) (void)hipHostFree(s->g_h_counters);
    if (s->ovf_index.p) (void)hipFree(s->ovf_index.p);
    for (GrowBuf* b : {&s->out_rgb, &s->out_id, &s->out_flags, &s->frame_rgb, &s->frame_id, &s->frame_flags, &s->in_a, &s->in_b,
                       &s->misc_a, &s->misc_b, &s->misc_c})
        if (b->p) (void)hipFree(b->p);
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->h_counters) (void)hipHostFree(s->h_counters);
    if (s->h_word) (void)hipHostFree(s->h_word);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int64_t rt_shard_pixel_count(const rt_camera_desc* cam, int32_t rank, int32_t n_ranks) {
    if (!cam || cam->hsize < 1 || cam->vsize < 1 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return RT_ERR_BAD_ARG;
    long long tiles = (long long)((cam->hsize - 1 + 7) / 8) * ((cam->vsize - 1 + 7) / 8);
    long long local_tiles = tiles > rank ? (tiles - rank + n_ranks - 1) / n_ranks : 0;
    return local_tiles * 64;
}

int rt_render_shard(rt_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, double* d_rgb_out, int32_t* d_hit_id_out,
                    uint8_t* d_flags_out, rt_stats* stats) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (!opts || !d_rgb_out) { rt_set_error("null argument"); return RT_ERR_BAD_ARG; }
    hipStream_t st = opts->stream ? static_cast<hipStream_t>(opts->stream) : s->stream;
    CameraParams c;
    long long n_slots = 0;
    int rc = render_shard_impl(s, cam, opts, st, d_rgb_out, d_hit_id_out, d_flags_out, stats, c, n_slots);
    if (rc != RT_OK) return rc;
    if (!opts->stream) HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int32_t rt_scene_meshes_built_on_gpu(const rt_scene* s) { return s ? s->meshes_built_on_gpu : 0; }
int32_t rt_scene_meshes_shared(const rt_scene* s) { return s ? s->meshes_shared : 0; }

// Canvas::new is all BLACK and Camera::render never writes the last row and column (F1): those are the only pixels of
// a frame no shard owns, so they are the only ones a whole-frame render has to clear.
static int clear_frame_border(hipStream_t st, const rt_camera_desc* cam, double* d_rgb_frame, int32_t* d_hit_id_frame, uint8_t* d_flags_frame) {
    const size_t W = size_t(cam->hsize), H = size_t(cam->vsize);
    HIP_TRY(hipMemsetAsync(d_rgb_frame + (H - 1) * W * 3, 0, W * 3 * sizeof(double), st));
    HIP_TRY(hipMemset2DAsync(d_rgb_frame + (W - 1) * 3, W * 3 * sizeof(double), 0, 3 * sizeof(double), H, st));
    if (d_hit_id_frame) {
        HIP_TRY(hipMemsetAsync(d_hit_id_frame + (H - 1) * W, 0, W * sizeof(int32_t), st));
        HIP_TRY(hipMemset2DAsync(d_hit_id_frame + (W - 1), W * sizeof(int32_t), 0, sizeof(int32_t), H, st));
    }
    if (d_flags_frame) {
        HIP_TRY(hipMemsetAsync(d_flags_frame + (H - 1) * W, 0, W, st));
        HIP_TRY(hipMemset2DAsync(d_flags_frame + (W - 1), W, 0, 1, H, st));
    }
    return RT_OK;
}

int rt_render_frame(rt_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, double* d_rgb_frame, int32_t* d_hit_id_frame,
                    uint8_t* d_flags_frame, rt_stats* stats) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (!opts || !d_rgb_frame || !cam) { rt_set_error("null argument"); return RT_ERR_BAD_ARG; }
    if (cam->hsize < 1 || cam->vsize < 1) { rt_set_error("camera size must be positive"); return RT_ERR_BAD_ARG; }
    hipStream_t st = opts->stream ? static_cast<hipStream_t>(opts->stream) : s->stream;
    { int rc0 = clear_frame_border(st, cam, d_rgb_frame, d_hit_id_frame, d_flags_frame); if (rc0 != RT_OK) return rc0; }
    CameraParams c;
    long long n_slots = 0;
    int rc = render_shard_impl(s, cam, opts, st, d_rgb_frame, d_hit_id_frame, d_flags_frame, stats, c, n_slots, true);
    if (rc != RT_OK) return rc;
    if (!opts->stream) HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_assemble(rt_scene* s, const rt_camera_desc* cam, int32_t n_ranks, const double* d_rgb_shards, const int32_t* d_id_shards,
                double* d_rgb_frame, int32_t* d_id_frame, void* stream) {
    return rt_assemble_strided(s, cam, n_ranks, d_rgb_shards, d_id_shards, 0, d_rgb_frame, d_id_frame, stream);
}

int rt_assemble_strided(rt_scene* s, const rt_camera_desc* cam, int32_t n_ranks, const double* d_rgb_shards, const int32_t* d_id_shards,
                        int64_t shard_stride, double* d_rgb_frame, int32_t* d_id_frame, void* stream) {
    // (on a caller's stream this touches nothing a frame in flight uses — no queue, no counter, not the scene's stream — so it
    // does not wait for the renders in flight: a one-process-per-GPU driver scatters frame k while its shards of k + 1, k + 2 render)
    if (!check_scene(s, stream != nullptr)) return RT_ERR_BAD_ARG;
    if (!cam || !d_rgb_shards || !d_rgb_frame) { rt_set_error("null argument"); return RT_ERR_BAD_ARG; }
    if (shard_stride < 0) { rt_set_error("negative shard stride"); return RT_ERR_BAD_ARG; }
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s->stream;
    size_t npx = size_t(cam->hsize) * size_t(cam->vsize);
    HIP_TRY(hipMemsetAsync(d_rgb_frame, 0, npx * 3 * sizeof(double), st));
    if (d_id_frame) HIP_TRY(hipMemsetAsync(d_id_frame, 0, npx * sizeof(int32_t), st));
    long long off = 0;
    for (int r = 0; r < n_ranks; ++r) {
        CameraParams c;
        long long n_slots = 0, n_valid = 0;
        bool any_cam = false;
        int rc = camera_params(cam, r, n_ranks, c, n_slots, n_valid, &any_cam);
        if (rc != RT_OK) return rc;
        if (shard_stride > 0 && n_slots > shard_stride) { rt_set_error("shard %d holds %lld pixels, the stride is %lld", r, n_slots, (long long)shard_stride); return RT_ERR_BAD_ARG; }
        if (n_slots)
            hipLaunchKernelGGL(k_assemble, dim3(blocks_for(n_slots)), dim3(BLOCK), 0, st, c, n_slots, d_rgb_shards + 3 * off,
                               d_id_shards ? d_id_shards + off : nullptr, (const uint8_t*)nullptr, d_rgb_frame, d_id_frame,
                               (uint8_t*)nullptr);
        off += shard_stride > 0 ? (long long)shard_stride : n_slots;
    }
    HIP_TRY(hipGetLastError());
    if (!stream) HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

// Canvas::scale_color (src/canvas.rs:86-98) for n_values f64 channel values in device memory -> u8, on `stream`.
int rt_quantise(rt_scene* s, const double* d_rgb, int64_t n_values, uint8_t* d_quantised_out, void* stream) {
    if (!check_scene(s, stream != nullptr)) return RT_ERR_BAD_ARG;  // (as rt_assemble_strided)
    if (n_values < 0 || (n_values > 0 && (!d_rgb || !d_quantised_out))) { rt_set_error("null argument or negative count"); return RT_ERR_BAD_ARG; }
    if (n_values == 0) return RT_OK;
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s->stream;
    hipLaunchKernelGGL(k_quantise, dim3(blocks_for(n_values)), dim3(BLOCK), 0, st, d_rgb, (long long)n_values, d_quantised_out);
    HIP_TRY(hipGetLastError());
    if (!stream) HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

// rt_assemble_strided for quantised (u8 RGB) shards: what a gather of rt_quantise'd shard buffers leaves on the root.
int rt_assemble_quantised(rt_scene* s, const rt_camera_desc* cam, int32_t n_ranks, const uint8_t* d_q_shards, int64_t shard_stride,
                          uint8_t* d_q_frame, void* stream) {
    if (!check_scene(s, stream != nullptr)) return RT_ERR_BAD_ARG;  // (as rt_assemble_strided)
    if (!cam || !d_q_shards || !d_q_frame) { rt_set_error("null argument"); return RT_ERR_BAD_ARG; }
    if (shard_stride < 0) { rt_set_error("negative shard stride"); return RT_ERR_BAD_ARG; }
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s->stream;
    const size_t npx = size_t(cam->hsize) * size_t(cam->vsize);
    HIP_TRY(hipMemsetAsync(d_q_frame, 0, npx * 3, st));
    long long off = 0;
    for (int r = 0; r < n_ranks; ++r) {
        CameraParams c;
        long long n_slots = 0, n_valid = 0;
        bool any_cam = false;
        int rc = camera_params(cam, r, n_ranks, c, n_slots, n_valid, &any_cam);
        if (rc != RT_OK) return rc;
        if (shard_stride > 0 && n_slots > shard_stride) { rt_set_error("shard %d holds %lld pixels, the stride is %lld", r, n_slots, (long long)shard_stride); return RT_ERR_BAD_ARG; }
        if (n_slots)
            hipLaunchKernelGGL(k_assemble_all, dim3(blocks_for(n_slots)), dim3(BLOCK), 0, st, c, n_slots, (const double*)nullptr, d_q_shards + 3 * off,
                               (const int32_t*)nullptr, (const uint8_t*)nullptr, (double*)nullptr, d_q_frame, (int32_t*)nullptr, (uint8_t*)nullptr);
        off += shard_stride > 0 ? (long long)shard_stride : n_slots;
    }
    HIP_TRY(hipGetLastError());
    if (!stream) HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_render(rt_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, double* rgb_out, int32_t* hit_id_out,
              uint8_t* flags_out, rt_stats* stats) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (!opts || !rgb_out || !cam) { rt_set_error("null argument"); return RT_ERR_BAD_ARG; }
    hipStream_t st = opts->stream ? static_cast<hipStream_t>(opts->stream) : s->stream;
    long long n_slots = rt_shard_pixel_count(cam, opts->rank, opts->n_ranks);
    if (n_slots < 0) { rt_set_error("bad camera size or shard"); return RT_ERR_BAD_ARG; }
    size_t npx = size_t(cam->hsize) * size_t(cam->vsize);
    int rc;
    if ((rc = grow(s->frame_rgb, npx * 3 * sizeof(double))) != RT_OK) return rc;
    if ((rc = grow(s->frame_id, npx * sizeof(int32_t))) != RT_OK) return rc;
    if ((rc = grow(s->frame_flags, npx)) != RT_OK) return rc;
    if (opts->n_ranks == 1) {  // every pixel but the F1 border is written by the render itself
        if ((rc = clear_frame_border(st, cam, static_cast<double*>(s->frame_rgb.p), static_cast<int32_t*>(s->frame_id.p),
                                     static_cast<uint8_t*>(s->frame_flags.p))) != RT_OK) return rc;
    } else {
        HIP_TRY(hipMemsetAsync(s->frame_rgb.p, 0, npx * 3 * sizeof(double), st));  // Canvas::new: all BLACK
        HIP_TRY(hipMemsetAsync(s->frame_id.p, 0, npx * sizeof(int32_t), st));
        HIP_TRY(hipMemsetAsync(s->frame_flags.p, 0, npx, st));
    }
    CameraParams c;
    rc = render_shard_impl(s, cam, opts, st, static_cast<double*>(s->frame_rgb.p), static_cast<int32_t*>(s->frame_id.p),
                           static_cast<uint8_t*>(s->frame_flags.p), stats, c, n_slots, true);  // straight into the frame
    if (rc != RT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(rgb_out, s->frame_rgb.p, npx * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (hit_id_out) HIP_TRY(hipMemcpyAsync(hit_id_out, s->frame_id.p, npx * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (flags_out) HIP_TRY(hipMemcpyAsync(flags_out, s->frame_flags.p, npx, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

// Camera::render (src/camera.rs:62-74) frame after frame, two frames in flight: see include/rt_abi.h and rt_scene::twin.
// `host`: the frame goes to HOST buffers (rendered into one of the scene's two device frames, copied on the copy stream);
// otherwise it is left in the caller's DEVICE frame.
// `shard` (device output only): the output is shard opts->rank of opts->n_ranks in shard-local pixel order, as rt_render_shard
// writes it — what a one-process-per-GPU driver hands to its gather.
static int pipe_begin(rt_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, bool host, bool shard, double* rgb_out,
                      int32_t* hit_id_out, uint8_t* flags_out, const char* who) {
    if (!check_scene(s, true)) return RT_ERR_BAD_ARG;
    if (!opts || !rgb_out || !cam) { rt_set_error("null argument"); return RT_ERR_BAD_ARG; }
    if (!shard && (opts->n_ranks != 1 || opts->rank != 0)) { rt_set_error("%s renders whole frames (rank 0 of 1)", who); return RT_ERR_BAD_ARG; }
    if (cam->hsize < 1 || cam->vsize < 1) { rt_set_error("camera size must be positive"); return RT_ERR_BAD_ARG; }
    if (shard && rt_shard_pixel_count(cam, opts->rank, opts->n_ranks) < 0) { rt_set_error("%s: bad camera size or shard", who); return RT_ERR_BAD_ARG; }
    if (s->pipe_pending >= rt_scene::PIPE_SLOTS) { rt_set_error("%d frames are in flight already: call rt_render_end first", rt_scene::PIPE_SLOTS); return RT_ERR_BAD_ARG; }
    const int k = s->pipe_head;
    const size_t npx = size_t(cam->hsize) * size_t(cam->vsize);
    int rc;
    double* d_rgb = rgb_out;
    int32_t* d_id = hit_id_out;
    uint8_t* d_fl = flags_out;
    if (host) {
        if (!s->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
        if (!s->pipe_rendered[k]) HIP_TRY(hipEventCreateWithFlags(&s->pipe_rendered[k], hipEventDisableTiming));
        if (!s->pipe_copied[k]) HIP_TRY(hipEventCreateWithFlags(&s->pipe_copied[k], hipEventDisableTiming));
        // (this device frame's previous copy was waited for by the job that made it, before that slot was retired)
        if ((rc = grow(s->pipe_rgb[k], npx * 3 * sizeof(double))) != RT_OK) return rc;
        if (hit_id_out && (rc = grow(s->pipe_id[k], npx * sizeof(int32_t))) != RT_OK) return rc;
        if (flags_out && (rc = grow(s->pipe_flags[k], npx)) != RT_OK) return rc;
        d_rgb = static_cast<double*>(s->pipe_rgb[k].p);
        d_id = hit_id_out ? static_cast<int32_t*>(s->pipe_id[k].p) : nullptr;
        d_fl = flags_out ? static_cast<uint8_t*>(s->pipe_flags[k].p) : nullptr;
    }
    if (!s->pipe_stream) HIP_TRY(hipStreamCreateWithFlags(&s->pipe_stream, hipStreamNonBlocking));
    if (!s->pipe_ready[k]) HIP_TRY(hipEventCreateWithFlags(&s->pipe_ready[k], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(s->pipe_ready[k], nullptr));  // what the caller has enqueued on the default stream so far
    if (s->pipe_lanes == 0) {  // first begin on this scene: as many lanes as twins can be made (none: one lane, frames one after the other)
        s->pipe_lanes = 1;
        while (s->pipe_lanes < s->max_lanes && make_twin(s, s->pipe_lanes - 1) == RT_OK) s->pipe_lanes++;
    }
    // frames are dealt to the render lanes in turn (with one lane the frames are rendered one after the other; their copies
    // still overlap the renders)
    const int lanes = s->pipe_lanes, lane = k % lanes;
    rt_scene* rs = lane == 0 ? s : s->twins[lane - 1];
    if (!s->pipe_worker[lane]) s->pipe_worker[lane] = new Worker();
    Worker* w = s->pipe_worker[lane];
    {   // the lane's previous frame, if it is still to be retired: wait for its render and put its result aside
        const int step = lanes, prev = (k - step + rt_scene::PIPE_SLOTS) % rt_scene::PIPE_SLOTS;
        if (s->pipe_pending >= step && !s->pipe_harvested[prev]) {
            s->pipe_rc[prev] = w->wait();
            s->pipe_err[prev] = w->err;
            s->pipe_harvested[prev] = true;
        }
    }
    const rt_camera_desc cam_v = *cam;
    rt_render_opts opts_v = *opts;
    opts_v.stream = nullptr;  // (a caller stream would have to be ordered against the other slot's by the caller)
    const int device = s->device;
    if (host && !s->pipe_copy_stream[k]) HIP_TRY(hipStreamCreateWithFlags(&s->pipe_copy_stream[k], hipStreamNonBlocking));
    hipStream_t copy_stream = host ? s->pipe_copy_stream[k] : nullptr;  // a copy stream per slot: copies of two frames may run side by side
    hipEvent_t rendered = host ? s->pipe_rendered[k] : nullptr, copied = host ? s->pipe_copied[k] : nullptr, ready = s->pipe_ready[k];
    hipStream_t st = rs == s ? s->pipe_stream : rs->stream;
    w->start([=]() -> int {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { rt_set_error("hipSetDevice(%d): %s", device, hipGetErrorString(e)); return RT_ERR_HIP; }
        t_plain_enqueue = true;
        HIP_TRY(hipStreamWaitEvent(st, ready, 0));
        int r = shard ? RT_OK : clear_frame_border(st, &cam_v, d_rgb, d_id, d_fl);
        if (r != RT_OK) return r;
        CameraParams c;
        long long n_slots = 0;
        r = render_shard_impl(rs, &cam_v, &opts_v, st, d_rgb, d_id, d_fl, nullptr, c, n_slots, !shard);
        if (r != RT_OK) return r;
        if (!host) {
            HIP_TRY(hipStreamSynchronize(st));
            return RT_OK;
        }
        HIP_TRY(hipEventRecord(rendered, st));
        HIP_TRY(hipStreamWaitEvent(copy_stream, rendered, 0));
        HIP_TRY(hipMemcpyAsync(rgb_out, d_rgb, npx * 3 * sizeof(double), hipMemcpyDeviceToHost, copy_stream));
        if (hit_id_out) HIP_TRY(hipMemcpyAsync(hit_id_out, d_id, npx * sizeof(int32_t), hipMemcpyDeviceToHost, copy_stream));
        if (flags_out) HIP_TRY(hipMemcpyAsync(flags_out, d_fl, npx, hipMemcpyDeviceToHost, copy_stream));
        HIP_TRY(hipEventRecord(copied, copy_stream));
        return RT_OK;  // (rt_render_end waits for the copy: this slot's stream and queues are free for the next frame now)
    });
    s->pipe_host[k] = host;
    s->pipe_harvested[k] = false;
    s->pipe_head = (k + 1) % rt_scene::PIPE_SLOTS;
    s->pipe_pending++;
    return RT_OK;
}

int rt_render_begin(rt_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, double* rgb_out, int32_t* hit_id_out,
                    uint8_t* flags_out) {
    return pipe_begin(s, cam, opts, true, false, rgb_out, hit_id_out, flags_out, "rt_render_begin");
}

int rt_render_frame_begin(rt_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, double* d_rgb_frame, int32_t* d_hit_id_frame,
                          uint8_t* d_flags_frame) {
    return pipe_begin(s, cam, opts, false, false, d_rgb_frame, d_hit_id_frame, d_flags_frame, "rt_render_frame_begin");
}

int rt_render_shard_begin(rt_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, double* d_rgb_out, int32_t* d_hit_id_out,
                          uint8_t* d_flags_out) {
    return pipe_begin(s, cam, opts, false, true, d_rgb_out, d_hit_id_out, d_flags_out, "rt_render_shard_begin");
}

int rt_render_end(rt_scene* s) {
    if (!check_scene(s, true)) return RT_ERR_BAD_ARG;
    if (s->pipe_pending == 0) return RT_OK;  // nothing in flight
    const int k = (s->pipe_head - s->pipe_pending + rt_scene::PIPE_SLOTS) % rt_scene::PIPE_SLOTS;  // the oldest frame begun and not yet retired
    Worker* w = s->pipe_worker[k % std::max(s->pipe_lanes, 1)];
    int rc = RT_OK;
    if (s->pipe_harvested[k]) {  // its lane has been given a later frame since
        rc = s->pipe_rc[k];
        s->pipe_harvested[k] = false;
        if (rc != RT_OK) rt_set_error("%s", s->pipe_err[k].c_str());
    } else if (w) {
        rc = w->wait();
        if (rc != RT_OK) rt_set_error("%s", w->err.c_str());
    }
    s->pipe_pending--;
    if (rc == RT_OK && s->pipe_host[k] && s->pipe_copied[k] && hipEventSynchronize(s->pipe_copied[k]) != hipSuccess) {
        rt_set_error("the copy of a frame to the host failed: %s", hipGetErrorString(hipGetLastError()));
        return RT_ERR_HIP;
    }
    return rc;
}

int rt_scene_create_multi(const rt_scene_desc* desc, int32_t n_gpus, const int32_t* devices, rt_scene** out) {
    return rt_scene_create_opts(desc, n_gpus, devices, nullptr, out);
}

int rt_scene_create_opts(const rt_scene_desc* desc, int32_t n_gpus, const int32_t* devices, const rt_scene_opts* opts, rt_scene** out) {
    if (!desc || !out) { rt_set_error("null argument"); return RT_ERR_BAD_ARG; }
    *out = nullptr;
    if (opts && opts->struct_size < int32_t(2 * sizeof(int32_t))) { rt_set_error("rt_scene_opts.struct_size is not set"); return RT_ERR_BAD_ARG; }
    Knobs knobs;
    resolve_knobs(opts, knobs);

    if (n_gpus < 1 || n_gpus > 64) { rt_set_error("n_gpus must be 1..64 (got %d)", n_gpus); return RT_ERR_BAD_ARG; }
    const int ndev = rt_device_count();
    if (ndev <= 0) { rt_set_error("no HIP device: this library has no CPU path"); return RT_ERR_NO_DEVICE; }
    std::unique_ptr<MultiState> ms(new MultiState());
    bool distinct = true;
    for (int r = 0; r < n_gpus; ++r) {
        const int dev = devices ? devices[r] : r;
        if (dev < 0 || dev >= ndev) { rt_set_error("device %d out of range (have %d)", dev, ndev); return RT_ERR_NO_DEVICE; }
        for (int q : ms->devices) distinct = distinct && q != dev;
        ms->devices.push_back(dev);
    }
    ms->replicas.assign(size_t(n_gpus), nullptr);
    ms->sent.assign(size_t(n_gpus), nullptr);
    ms->shard_q.resize(size_t(n_gpus));
    for (int r = 0; r < n_gpus; ++r) ms->workers.emplace_back(new Worker());
    // every device compiles and uploads its own copy of the tables, in parallel (scene tables are small next to HBM:
    // the teapot is ~1.3 MB, a 1 M-triangle grid ~0.3 GB)
    MultiState* m = ms.get();
    int rc = on_all_replicas(m, [m, desc, &knobs](int r) -> int {
        int rc2 = scene_create_one(desc, m->devices[size_t(r)], knobs, &m->replicas[size_t(r)]);
        if (rc2 != RT_OK) return rc2;
        HIP_TRY(hipEventCreateWithFlags(&m->sent[size_t(r)], hipEventDisableTiming));
        return RT_OK;
    });
    if (rc != RT_OK) {
        const std::string keep = rt_last_error();
        m->workers.clear();
        for (int r = 0; r < n_gpus; ++r) {
            (void)hipSetDevice(m->devices[size_t(r)]);
            if (m->sent[size_t(r)]) (void)hipEventDestroy(m->sent[size_t(r)]);
            if (m->replicas[size_t(r)]) rt_scene_destroy(m->replicas[size_t(r)]);
        }
        rt_set_error("%s", keep.c_str());
        return rc;
    }
    // ---- how the shards will travel
    const bool force_rccl = knobs.gather == 1, no_rccl = knobs.gather == 2;
    if (n_gpus > 1 || force_rccl) {
        m->gather_kind = 2;
        if (no_rccl) m->gather_note = "peer copies asked for (rt_scene_opts.gather / RT_AMD_GATHER)";
        else if (!distinct) m->gather_note = "the device list names a device twice (RCCL wants one rank per GPU)";
        else if (RcclApi& R = rccl_api(); !R.ok()) m->gather_note = R.why;
        else {
            m->comms.assign(size_t(n_gpus), nullptr);
            const int r2 = R.comm_init_all(m->comms.data(), n_gpus, m->devices.data());
            if (r2 == 0) { m->gather_kind = 1; m->rccl = &R; m->gather_note = "ncclGather (RCCL)"; }
            else { m->gather_note = std::string("ncclCommInitAll failed: ") + R.errstr(r2); m->comms.clear(); (void)hipGetLastError(); }
        }
        if (m->gather_kind == 2) m->gather_note = "hipMemcpyPeerAsync: " + m->gather_note;
    }
    if (knobs.verbose) std::fprintf(stderr, "rt_scene_create_multi: %d replicas, gather: %s\n", n_gpus, m->gather_kind ? m->gather_note.c_str() : "none");
    (void)hipSetDevice(m->devices[0]);
    rt_scene* root = m->replicas[0];
    root->multi = ms.release();
    *out = root;
    return RT_OK;
}

int32_t rt_scene_gpu_count(const rt_scene* s) { return s ? (s->multi ? int32_t(s->multi->replicas.size()) : 1) : 0; }

int32_t rt_scene_gather_kind(const rt_scene* s) { return s && s->multi ? s->multi->gather_kind : 0; }

int rt_render_multi(rt_scene* s, const rt_camera_desc* cam, const rt_render_opts* opts, double* rgb_out, int32_t* hit_id_out,
                    uint8_t* flags_out, uint8_t* quantised_out, rt_stats* stats_per_gpu) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (!opts || !cam || (!rgb_out && !quantised_out)) { rt_set_error("null argument (one of rgb_out / quantised_out is required)"); return RT_ERR_BAD_ARG; }
    MultiState* ms = s->multi;
    if (!ms) { rt_set_error("rt_render_multi needs a scene from rt_scene_create_multi"); return RT_ERR_BAD_ARG; }
    const int N = int(ms->replicas.size());
    const long long n_max = rt_shard_pixel_count(cam, 0, N);  // shard 0 is never smaller than another
    if (n_max < 0) { rt_set_error("bad camera size"); return RT_ERR_BAD_ARG; }
    const bool want_rgb = rgb_out != nullptr, want_q = quantised_out != nullptr, want_id = hit_id_out != nullptr, want_fl = flags_out != nullptr;
    const size_t npx = size_t(cam->hsize) * size_t(cam->vsize);
    // ---- every device: its shard into its own HBM (padded to n_max pixels so that one ncclGather moves them)
    int rc = on_all_replicas(ms, [=](int r) -> int {
        rt_scene* q = ms->replicas[size_t(r)];
        int rc2;
        if ((rc2 = grow(q->out_rgb, size_t(n_max) * 3 * sizeof(double))) != RT_OK) return rc2;
        if (want_id && (rc2 = grow(q->out_id, size_t(n_max) * sizeof(int32_t))) != RT_OK) return rc2;
        if (want_fl && (rc2 = grow(q->out_flags, size_t(n_max))) != RT_OK) return rc2;
        if (want_q && (rc2 = grow(ms->shard_q[size_t(r)], size_t(n_max) * 3)) != RT_OK) return rc2;
        rt_render_opts o = *opts;
        o.rank = r;
        o.n_ranks = N;
        o.stream = nullptr;
        CameraParams c;
        long long n_slots = 0;
        rc2 = render_shard_impl(q, cam, &o, q->stream, static_cast<double*>(q->out_rgb.p), want_id ? static_cast<int32_t*>(q->out_id.p) : nullptr,
                                want_fl ? static_cast<uint8_t*>(q->out_flags.p) : nullptr, stats_per_gpu ? stats_per_gpu + r : nullptr, c, n_slots);
        if (rc2 != RT_OK) return rc2;
        if (want_q && n_slots > 0) {  // scale_color on the shard: the u8 shard is 8x smaller on the wire than the f64 one
            hipLaunchKernelGGL(k_quantise, dim3(blocks_for(n_slots * 3)), dim3(BLOCK), 0, q->stream, static_cast<const double*>(q->out_rgb.p),
                               n_slots * 3, static_cast<uint8_t*>(ms->shard_q[size_t(r)].p));
            HIP_TRY(hipGetLastError());
        }
        return RT_OK;
    });
    if (rc != RT_OK) return rc;
    // ---- gather to the root (device of replica 0), stream-ordered behind each replica's render
    HIP_TRY(hipSetDevice(s->device));
    const bool gathered = N > 1 || ms->gather_kind == 1;
    if (gathered) {
        if (want_rgb && (rc = grow(ms->g_rgb, size_t(N) * size_t(n_max) * 3 * sizeof(double))) != RT_OK) return rc;
        if (want_q && (rc = grow(ms->g_q, size_t(N) * size_t(n_max) * 3)) != RT_OK) return rc;
        if (want_id && (rc = grow(ms->g_id, size_t(N) * size_t(n_max) * sizeof(int32_t))) != RT_OK) return rc;
        if (want_fl && (rc = grow(ms->g_flags, size_t(N) * size_t(n_max))) != RT_OK) return rc;
    }
    struct Part { bool on; GrowBuf* dst; size_t elem; int nccl_type; };
    const Part parts[4] = {{want_rgb, &ms->g_rgb, 3 * sizeof(double), NCCL_FLOAT64}, {want_q, &ms->g_q, 3, NCCL_UINT8},
                           {want_id, &ms->g_id, sizeof(int32_t), NCCL_INT32}, {want_fl, &ms->g_flags, 1, NCCL_UINT8}};
    auto shard_ptr = [&](int r, int part) -> void* {
        rt_scene* q = ms->replicas[size_t(r)];
        return part == 0 ? q->out_rgb.p : part == 1 ? ms->shard_q[size_t(r)].p : part == 2 ? q->out_id.p : q->out_flags.p;
    };
    if (gathered && ms->gather_kind == 1) {
        RcclApi& R = *ms->rccl;
        for (int part = 0; part < 4; ++part) {
            if (!parts[part].on) continue;
            const size_t count = size_t(n_max) * (part == 0 ? 3 : part == 1 ? 3 : 1);  // elements of the part's datatype per rank
            int e = R.group_start();
            for (int r = 0; r < N && e == 0; ++r)
                e = R.gather(shard_ptr(r, part), r == 0 ? parts[part].dst->p : nullptr, count, parts[part].nccl_type, 0, ms->comms[size_t(r)],
                             ms->replicas[size_t(r)]->stream);
            const int e2 = R.group_end();
            if (e != 0 || e2 != 0) { rt_set_error("ncclGather failed: %s", R.errstr(e != 0 ? e : e2)); return RT_ERR_HIP; }
        }
    } else if (gathered) {
        for (int r = 0; r < N; ++r) {
            HIP_TRY(hipSetDevice(ms->devices[size_t(r)]));
            hipStream_t sr = ms->replicas[size_t(r)]->stream;
            for (int part = 0; part < 4; ++part) {
                if (!parts[part].on) continue;
                const size_t bytes = size_t(n_max) * parts[part].elem;
                HIP_TRY(hipMemcpyPeerAsync(static_cast<char*>(parts[part].dst->p) + size_t(r) * bytes, s->device, shard_ptr(r, part),
                                           ms->devices[size_t(r)], bytes, sr));
            }
            HIP_TRY(hipEventRecord(ms->sent[size_t(r)], sr));
        }
        HIP_TRY(hipSetDevice(s->device));
        for (int r = 1; r < N; ++r) HIP_TRY(hipStreamWaitEvent(s->stream, ms->sent[size_t(r)], 0));
    }
    // ---- root: scatter the shards into the row-major frame(s), copy to the host
    hipStream_t st = s->stream;
    if (want_rgb) { if ((rc = grow(s->frame_rgb, npx * 3 * sizeof(double))) != RT_OK) return rc; HIP_TRY(hipMemsetAsync(s->frame_rgb.p, 0, npx * 3 * sizeof(double), st)); }
    if (want_q) { if ((rc = grow(ms->frame_q, npx * 3)) != RT_OK) return rc; HIP_TRY(hipMemsetAsync(ms->frame_q.p, 0, npx * 3, st)); }
    if (want_id) { if ((rc = grow(s->frame_id, npx * sizeof(int32_t))) != RT_OK) return rc; HIP_TRY(hipMemsetAsync(s->frame_id.p, 0, npx * sizeof(int32_t), st)); }
    if (want_fl) { if ((rc = grow(s->frame_flags, npx)) != RT_OK) return rc; HIP_TRY(hipMemsetAsync(s->frame_flags.p, 0, npx, st)); }
    for (int r = 0; r < N; ++r) {
        CameraParams c;
        long long n_slots = 0, n_valid = 0;
        bool any_cam = false;
        if ((rc = camera_params(cam, r, N, c, n_slots, n_valid, &any_cam)) != RT_OK) return rc;
        if (n_slots == 0) continue;
        const size_t off = gathered ? size_t(r) * size_t(n_max) : 0;
        const double* rgb_in = want_rgb ? (gathered ? static_cast<const double*>(ms->g_rgb.p) : static_cast<const double*>(s->out_rgb.p)) + off * 3 : nullptr;
        const uint8_t* q_in = want_q ? (gathered ? static_cast<const uint8_t*>(ms->g_q.p) : static_cast<const uint8_t*>(ms->shard_q[0].p)) + off * 3 : nullptr;
        const int32_t* id_in = want_id ? (gathered ? static_cast<const int32_t*>(ms->g_id.p) : static_cast<const int32_t*>(s->out_id.p)) + off : nullptr;
        const uint8_t* fl_in = want_fl ? (gathered ? static_cast<const uint8_t*>(ms->g_flags.p) : static_cast<const uint8_t*>(s->out_flags.p)) + off : nullptr;
        hipLaunchKernelGGL(k_assemble_all, dim3(blocks_for(n_slots)), dim3(BLOCK), 0, st, c, n_slots, rgb_in, q_in, id_in, fl_in,
                           want_rgb ? static_cast<double*>(s->frame_rgb.p) : nullptr, want_q ? static_cast<uint8_t*>(ms->frame_q.p) : nullptr,
                           want_id ? static_cast<int32_t*>(s->frame_id.p) : nullptr, want_fl ? static_cast<uint8_t*>(s->frame_flags.p) : nullptr);
    }
    HIP_TRY(hipGetLastError());
    if (want_rgb) HIP_TRY(hipMemcpyAsync(rgb_out, s->frame_rgb.p, npx * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (want_q) HIP_TRY(hipMemcpyAsync(quantised_out, ms->frame_q.p, npx * 3, hipMemcpyDeviceToHost, st));
    if (want_id) HIP_TRY(hipMemcpyAsync(hit_id_out, s->frame_id.p, npx * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (want_fl) HIP_TRY(hipMemcpyAsync(flags_out, s->frame_flags.p, npx, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (gathered && ms->gather_kind == 2)  // the peers' streams: their copies are done (the root waited on them); nothing pending
        for (int r = 1; r < N; ++r) { HIP_TRY(hipSetDevice(ms->devices[size_t(r)])); HIP_TRY(hipStreamSynchronize(ms->replicas[size_t(r)]->stream)); }
    if (gathered && ms->gather_kind == 1)  // RCCL's work on the peers' streams must be over before their buffers are rewritten
        for (int r = 1; r < N; ++r) { HIP_TRY(hipSetDevice(ms->devices[size_t(r)])); HIP_TRY(hipStreamSynchronize(ms->replicas[size_t(r)]->stream)); }
    HIP_TRY(hipSetDevice(s->device));
    return RT_OK;
}

int64_t rt_canvas_to_ppm(rt_scene* s, const double* d_rgb_frame, int32_t hsize, int32_t vsize, uint8_t* d_quantised_out, char* ppm_out,
                         int64_t cap) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (!d_rgb_frame || hsize < 1 || vsize < 1 || cap < 0 || (cap > 0 && !ppm_out)) { rt_set_error("bad argument"); return RT_ERR_BAD_ARG; }
    hipStream_t st = s->stream;
    const long long n = (long long)hsize * vsize * 3;
    const int vpr = hsize * 3;
    int rc;
    uint8_t* q = d_quantised_out;
    if (!q) {
        if ((rc = grow(s->misc_a, size_t(n))) != RT_OK) return rc;
        q = static_cast<uint8_t*>(s->misc_a.p);
    }
    if ((rc = grow(s->misc_b, size_t(vsize + 1) * sizeof(long long))) != RT_OK) return rc;
    long long* rows = static_cast<long long*>(s->misc_b.p);
    hipLaunchKernelGGL(k_quantise, dim3(blocks_for(n)), dim3(BLOCK), 0, st, d_rgb_frame, n, q);
    hipLaunchKernelGGL(k_ppm_row_bytes, dim3(blocks_for(vsize)), dim3(BLOCK), 0, st, q, vpr, (int)vsize, rows);
    hipLaunchKernelGGL(k_ppm_scan, dim3(1), dim3(BLOCK), 0, st, rows, (int)vsize);
    HIP_TRY(hipGetLastError());
    long long body = 0;
    HIP_TRY(hipMemcpyAsync(&body, rows + vsize, sizeof body, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    char header[64];
    const int hl = std::snprintf(header, sizeof header, "P3\n%d %d\n255\n", hsize, vsize);
    const int64_t total = hl + body;
    if (!ppm_out || cap < total) return total;  // the length alone (call again with a buffer that holds it)
    if ((rc = grow(s->misc_c, size_t(body))) != RT_OK) return rc;
    hipLaunchKernelGGL(k_ppm_write, dim3(blocks_for(vsize)), dim3(BLOCK), 0, st, q, vpr, (int)vsize, rows, static_cast<char*>(s->misc_c.p), -1ll);
    HIP_TRY(hipGetLastError());
    std::memcpy(ppm_out, header, size_t(hl));
    HIP_TRY(hipMemcpyAsync(ppm_out + hl, s->misc_c.p, size_t(body), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return total;
}

int64_t rt_canvas_to_ppm_noalloc(rt_scene* s, const double* d_rgb_frame, int32_t hsize, int32_t vsize, uint8_t* d_quantised_out,
                                 char* d_ppm_out, int64_t cap, int64_t* d_len_out, void* stream) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (!d_rgb_frame || !d_quantised_out || hsize < 1 || vsize < 1 || cap < 0 || (cap > 0 && !d_ppm_out)) {
        rt_set_error("bad argument (the quantised frame is a required output here: it is this entry point's staging)");
        return RT_ERR_BAD_ARG;
    }
    if (vsize > NOALLOC_MAX_ROWS) {
        rt_set_error("rt_canvas_to_ppm_noalloc handles canvases of at most %d rows (got %d): its row-offset scratch is a fixed device array; "
                     "use rt_canvas_to_ppm", NOALLOC_MAX_ROWS, vsize);
        return RT_ERR_UNSUPPORTED;
    }
    long long* rows = scratch_of(s->device);  // this device's copy of `scratch` (never the array's name)
    if (!rows) { rt_set_error("no static scratch on device %d", s->device); return RT_ERR_UNSUPPORTED; }
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s->stream;
    const long long n = (long long)hsize * vsize * 3;
    const int vpr = hsize * 3;
    char header[64];
    const int hl = std::snprintf(header, sizeof header, "P3\n%d %d\n255\n", hsize, vsize);
    // Every word of rows[0 .. vsize] is rewritten by k_ppm_row_bytes / k_ppm_scan before anything reads it, so nothing
    // of the previous call's contents has to be cleared; words past vsize are never read.
    hipLaunchKernelGGL(k_quantise, dim3(blocks_for(n)), dim3(BLOCK), 0, st, d_rgb_frame, n, d_quantised_out);
    hipLaunchKernelGGL(k_ppm_row_bytes, dim3(blocks_for(vsize)), dim3(BLOCK), 0, st, d_quantised_out, vpr, (int)vsize, rows);
    hipLaunchKernelGGL(k_ppm_scan, dim3(1), dim3(BLOCK), 0, st, rows, (int)vsize);
    hipLaunchKernelGGL(k_ppm_header, dim3(1), dim3(64), 0, st, (int)hsize, (int)vsize, rows, (int)vsize, d_ppm_out, (long long)cap,
                       reinterpret_cast<long long*>(d_len_out));
    if (cap > hl)
        hipLaunchKernelGGL(k_ppm_write, dim3(blocks_for(vsize)), dim3(BLOCK), 0, st, d_quantised_out, vpr, (int)vsize, rows, d_ppm_out + hl,
                           (long long)cap - hl);
    HIP_TRY(hipGetLastError());
    if (d_len_out) return 0;  // the length arrives in *d_len_out in stream order; no host wait here
    // into the page-locked word the scene has held since its creation: a copy into pageable memory is staged by the
    // runtime through pinned memory it may allocate on first use
    HIP_TRY(hipMemcpyAsync(s->h_word, rows + vsize, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return hl + *s->h_word;
}

int rt_color_at(rt_scene* s, int64_t n_rays, const double* origins, const double* directions, int32_t remaining, double* rgb_out,
                int32_t* hit_id_out, uint8_t* flags_out, rt_stats* stats) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (n_rays < 0 || !origins || !directions || !rgb_out) { rt_set_error("null argument or negative count"); return RT_ERR_BAD_ARG; }
    if (stats) { std::memset(stats, 0, sizeof *stats); stats->rays_primary = n_rays; }
    if (n_rays == 0) return RT_OK;
    hipStream_t st = s->stream;
    int rc;
    size_t rb = size_t(n_rays) * 3 * sizeof(double);
    if ((rc = grow(s->in_a, rb)) != RT_OK || (rc = grow(s->in_b, rb)) != RT_OK || (rc = grow(s->out_rgb, rb)) != RT_OK ||
        (rc = grow(s->out_id, size_t(n_rays) * sizeof(int32_t))) != RT_OK || (rc = grow(s->out_flags, size_t(n_rays))) != RT_OK)
        return rc;
    if (s->general) {  // some inverse of the scene is not affine: the 4-component path (rt_general.hip.h)
        if (remaining + 2 > G_FRAME_CAP) { rt_set_error("the 4-component path holds ray trees up to %d deep", G_FRAME_CAP - 2); return RT_ERR_UNSUPPORTED; }
        if ((rc = ensure_general(s, n_rays)) != RT_OK || (rc = general_counters_begin(s, st)) != RT_OK) return rc;
        HIP_TRY(hipMemcpyAsync(s->in_a.p, origins, rb, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->in_b.p, directions, rb, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_general_color_at, dim3(general_grid(n_rays)), dim3(GBLOCK), 0, st, s->ds, s->gs, (long long)n_rays,
                           static_cast<const double*>(s->in_a.p), static_cast<const double*>(s->in_b.p), int(remaining < 0 ? 0 : remaining),
                           static_cast<double*>(s->out_rgb.p), static_cast<int32_t*>(s->out_id.p), static_cast<uint8_t*>(s->out_flags.p));
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(rgb_out, s->out_rgb.p, rb, hipMemcpyDeviceToHost, st));
        if (hit_id_out) HIP_TRY(hipMemcpyAsync(hit_id_out, s->out_id.p, size_t(n_rays) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (flags_out) HIP_TRY(hipMemcpyAsync(flags_out, s->out_flags.p, size_t(n_rays), hipMemcpyDeviceToHost, st));
        return general_counters_end(s, st, stats, 1);
    }
    if ((rc = ensure_level(s, 0, n_rays)) != RT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(s->in_a.p, origins, rb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->in_b.p, directions, rb, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_load_rays, dim3(blocks_for(n_rays)), dim3(BLOCK), 0, st, s->levels[0].L, (long long)n_rays,
                       static_cast<const double*>(s->in_a.p), static_cast<const double*>(s->in_b.p));
    rc = run_levels(s, st, n_rays, remaining, false, stats);
    if (rc != RT_OK) return rc;
    hipLaunchKernelGGL(k_output, dim3(blocks_for(n_rays)), dim3(BLOCK), 0, st, s->ds, s->levels[0].L, (long long)n_rays,
                       static_cast<double*>(s->out_rgb.p), static_cast<int32_t*>(s->out_id.p), static_cast<uint8_t*>(s->out_flags.p), 1, 0,
                       CameraParams{}, 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgb_out, s->out_rgb.p, rb, hipMemcpyDeviceToHost, st));
    if (hit_id_out) HIP_TRY(hipMemcpyAsync(hit_id_out, s->out_id.p, size_t(n_rays) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (flags_out) HIP_TRY(hipMemcpyAsync(flags_out, s->out_flags.p, size_t(n_rays), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

// One pass of k_list over n rays given on the host; big: the variant whose lists live in the arena (ensure_arena first).
static int list_pass(rt_scene* s, bool big, int64_t n, const double* origins, const double* directions, int32_t max_hits, int32_t* count_out,
                     double* t_out, int32_t* object_id_out, uint8_t* flags_host) {
    hipStream_t st = s->stream;
    int rc;
    const size_t rb = size_t(n) * 3 * sizeof(double), lb = size_t(n) * size_t(max_hits);
    if ((rc = grow(s->in_a, rb)) != RT_OK || (rc = grow(s->in_b, rb)) != RT_OK || (rc = grow(s->misc_a, size_t(n) * sizeof(int32_t))) != RT_OK ||
        (rc = grow(s->misc_b, lb * sizeof(double))) != RT_OK || (rc = grow(s->misc_c, lb * sizeof(int32_t))) != RT_OK ||
        (rc = grow(s->out_flags, size_t(n))) != RT_OK)
        return rc;
    HIP_TRY(hipMemcpyAsync(s->in_a.p, origins, rb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->in_b.p, directions, rb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(s->misc_b.p, 0, lb * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(s->misc_c.p, 0, lb * sizeof(int32_t), st));
    const double* o = static_cast<const double*>(s->in_a.p);
    const double* d = static_cast<const double*>(s->in_b.p);
    int32_t* cnt = static_cast<int32_t*>(s->misc_a.p);
    double* tt = static_cast<double*>(s->misc_b.p);
    int32_t* ids = static_cast<int32_t*>(s->misc_c.p);
    uint8_t* fl = static_cast<uint8_t*>(s->out_flags.p);
    s->big = big;
    if (s->general) {
        if ((rc = ensure_general(s, n)) != RT_OK) return rc;
        hipLaunchKernelGGL(k_general_list, dim3(general_grid(n)), dim3(GBLOCK), 0, st, s->ds, s->gs, (long long)n, o, d, (int)max_hits, cnt, tt, ids, fl);
    } else if (big) rc = launch_trav(s, st, k_list<SC_CSG | SC_CULL | SC_TLAS | SC_BIG>, chunks_for(n), s->ds, (long long)n, o, d, (int)max_hits, cnt, tt, ids, fl);
    else if (s->has_csg || s->has_tlas) rc = launch_trav(s, st, k_list<SC_CSG | SC_CULL | SC_TLAS>, chunks_for(n), s->ds, (long long)n, o, d, (int)max_hits, cnt, tt, ids, fl);
    else rc = launch_trav(s, st, k_list<SC_CULL>, chunks_for(n), s->ds, (long long)n, o, d, (int)max_hits, cnt, tt, ids, fl);
    s->big = false;
    if (rc != RT_OK) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(count_out, cnt, size_t(n) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(t_out, tt, lb * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(object_id_out, ids, lb * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(flags_host, fl, size_t(n), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_intersect_world(rt_scene* s, int64_t n_rays, const double* origins, const double* directions, int32_t max_hits,
                       int32_t* count_out, double* t_out, int32_t* object_id_out) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (n_rays < 0 || !origins || !directions || !count_out || !t_out || !object_id_out || max_hits < 1) {
        rt_set_error("null argument, negative count or max_hits < 1");
        return RT_ERR_BAD_ARG;
    }
    if (n_rays == 0) return RT_OK;
    std::vector<uint8_t> flags((size_t)n_rays);
    int rc = list_pass(s, false, n_rays, origins, directions, max_hits, count_out, t_out, object_id_out, flags.data());
    if (rc != RT_OK) return rc;
    // rays whose list (or a CSG subtree's pending list) did not fit the per-lane arrays: again, with the lists in HBM
    int cap = s->general ? G_LIST_CAP : LIST_CAP;  // (the 4-component path has one list size and no second round)
    std::vector<int64_t> todo;
    for (int64_t i = 0; i < n_rays; ++i)
        if (count_out[i] > cap || (flags[size_t(i)] & RT_FLAG_LIST_OVERFLOW)) todo.push_back(i);
    for (int round = 0; round < N_ARENA_ROUNDS && !todo.empty() && !s->general; ++round) {
        if ((rc = ensure_arena(s, round)) != RT_OK) return rc;
        cap = ARENA_ROUNDS[round].list_cap;
        const size_t m = todo.size();
        std::vector<double> o(m * 3), d(m * 3), tt(m * size_t(max_hits));
        std::vector<int32_t> cnt(m), ids(m * size_t(max_hits));
        std::vector<uint8_t> fl(m);
        for (size_t j = 0; j < m; ++j)
            for (int k = 0; k < 3; ++k) { o[3 * j + k] = origins[3 * todo[j] + k]; d[3 * j + k] = directions[3 * todo[j] + k]; }
        if ((rc = list_pass(s, true, int64_t(m), o.data(), d.data(), max_hits, cnt.data(), tt.data(), ids.data(), fl.data())) != RT_OK) return rc;
        std::vector<int64_t> still;
        for (size_t j = 0; j < m; ++j) {
            const int64_t i = todo[j];
            count_out[i] = cnt[j];
            std::memcpy(t_out + size_t(i) * max_hits, tt.data() + j * size_t(max_hits), size_t(max_hits) * sizeof(double));
            std::memcpy(object_id_out + size_t(i) * max_hits, ids.data() + j * size_t(max_hits), size_t(max_hits) * sizeof(int32_t));
            flags[size_t(i)] = fl[j];
            if (cnt[j] > cap || (fl[j] & RT_FLAG_LIST_OVERFLOW)) still.push_back(i);
        }
        todo.swap(still);
    }
    if (!todo.empty()) {
        rt_set_error("ray %lld has %d intersections (or more than %d pending inside one CSG subtree); the largest device list holds %d",
                     (long long)todo[0], count_out[todo[0]], ARENA_ROUNDS[N_ARENA_ROUNDS - 1].csg_cap, cap);
        return RT_ERR_OVERFLOW;
    }
    return RT_OK;
}

static int shadow_pass(rt_scene* s, bool big, int64_t n, const double* points, uint8_t* out, uint8_t* flags_host) {
    hipStream_t st = s->stream;
    int rc;
    const size_t rb = size_t(n) * 3 * sizeof(double), ob = size_t(n) * size_t(s->n_lights);
    if ((rc = grow(s->in_a, rb)) != RT_OK || (rc = grow(s->misc_a, ob)) != RT_OK || (rc = grow(s->out_flags, size_t(n))) != RT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(s->in_a.p, points, rb, hipMemcpyHostToDevice, st));
    const double* p = static_cast<const double*>(s->in_a.p);
    uint8_t* o = static_cast<uint8_t*>(s->misc_a.p);
    uint8_t* fl = static_cast<uint8_t*>(s->out_flags.p);
    s->big = big;
    if (s->general) {
        if ((rc = ensure_general(s, n)) != RT_OK) return rc;
        hipLaunchKernelGGL(k_general_shadow, dim3(general_grid(n)), dim3(GBLOCK), 0, st, s->ds, s->gs, (long long)n, p, o, fl);
    } else if (big) rc = launch_trav(s, st, k_shadow_points<SC_CSG | SC_CULL | SC_TLAS | SC_BIG>, chunks_for(n), s->ds, (long long)n, p, o, fl);
    else if (s->has_csg || s->has_tlas) rc = launch_trav(s, st, k_shadow_points<SC_CSG | SC_CULL | SC_TLAS>, chunks_for(n), s->ds, (long long)n, p, o, fl);
    else rc = launch_trav(s, st, k_shadow_points<SC_CULL>, chunks_for(n), s->ds, (long long)n, p, o, fl);
    s->big = false;
    if (rc != RT_OK) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, o, ob, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(flags_host, fl, size_t(n), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_is_shadowed(rt_scene* s, int64_t n_points, const double* points, uint8_t* out) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (n_points < 0 || !points || !out) { rt_set_error("null argument or negative count"); return RT_ERR_BAD_ARG; }
    if (n_points == 0 || s->n_lights == 0) return RT_OK;
    std::vector<uint8_t> flags((size_t)n_points);
    int rc = shadow_pass(s, false, n_points, points, out, flags.data());
    if (rc != RT_OK) return rc;
    std::vector<int64_t> todo;  // shadow rays through a CSG subtree with more pending intersections than the per-lane array holds
    for (int64_t i = 0; i < n_points; ++i)
        if (flags[size_t(i)] & RT_FLAG_LIST_OVERFLOW) todo.push_back(i);
    const size_t nl = size_t(s->n_lights);
    for (int round = 0; round < N_ARENA_ROUNDS && !todo.empty() && !s->general; ++round) {
        if ((rc = ensure_arena(s, round)) != RT_OK) return rc;
        const size_t m = todo.size();
        std::vector<double> p(m * 3);
        std::vector<uint8_t> o(m * nl), fl(m);
        for (size_t j = 0; j < m; ++j)
            for (int k = 0; k < 3; ++k) p[3 * j + k] = points[3 * todo[j] + k];
        if ((rc = shadow_pass(s, true, int64_t(m), p.data(), o.data(), fl.data())) != RT_OK) return rc;
        std::vector<int64_t> still;
        for (size_t j = 0; j < m; ++j) {
            std::memcpy(out + size_t(todo[j]) * nl, o.data() + j * nl, nl);
            if (fl[j] & RT_FLAG_LIST_OVERFLOW) still.push_back(todo[j]);
        }
        todo.swap(still);
    }
    if (!todo.empty()) {
        rt_set_error("shadow ray of point %lld has more than %d intersections pending inside one CSG subtree", (long long)todo[0],
                     ARENA_ROUNDS[N_ARENA_ROUNDS - 1].csg_cap);
        return RT_ERR_OVERFLOW;
    }
    return RT_OK;
}

int rt_prepare_computations(rt_scene* s, int64_t n_rays, const double* origins, const double* directions, int32_t* hit_id_out,
                            double* comps_out, uint8_t* flags_out) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (n_rays < 0 || !origins || !directions || !hit_id_out || !comps_out) { rt_set_error("null argument or negative count"); return RT_ERR_BAD_ARG; }
    if (n_rays == 0) return RT_OK;
    hipStream_t st = s->stream;
    int rc;
    const size_t rb = size_t(n_rays) * 3 * sizeof(double), cb = size_t(n_rays) * RT_COMP_COUNT * sizeof(double);
    if ((rc = grow(s->in_a, rb)) != RT_OK || (rc = grow(s->in_b, rb)) != RT_OK || (rc = grow(s->misc_a, size_t(n_rays) * sizeof(int32_t))) != RT_OK ||
        (rc = grow(s->misc_b, cb)) != RT_OK || (rc = grow(s->out_flags, size_t(n_rays))) != RT_OK)
        return rc;
    HIP_TRY(hipMemcpyAsync(s->in_a.p, origins, rb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->in_b.p, directions, rb, hipMemcpyHostToDevice, st));
    const double* o = static_cast<const double*>(s->in_a.p);
    const double* d = static_cast<const double*>(s->in_b.p);
    int32_t* ids = static_cast<int32_t*>(s->misc_a.p);
    double* comps = static_cast<double*>(s->misc_b.p);
    uint8_t* fl = static_cast<uint8_t*>(s->out_flags.p);
    if (s->general) {  // the 4-component path; x, y, z of every tuple are handed back
        if ((rc = ensure_general(s, n_rays)) != RT_OK) return rc;
        hipLaunchKernelGGL(k_general_comps, dim3(general_grid(n_rays)), dim3(GBLOCK), 0, st, s->ds, s->gs, (long long)n_rays, o, d, ids, comps, fl);
    } else if (s->has_csg || s->has_tlas) rc = launch_trav(s, st, k_comps<SC_CSG | SC_CULL | SC_TLAS>, chunks_for(n_rays), s->ds, (long long)n_rays, o, d, ids, comps, fl);
    else rc = launch_trav(s, st, k_comps<SC_CULL>, chunks_for(n_rays), s->ds, (long long)n_rays, o, d, ids, comps, fl);
    if (rc != RT_OK) return rc;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(hit_id_out, ids, size_t(n_rays) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(comps_out, comps, cb, hipMemcpyDeviceToHost, st));
    if (flags_out) HIP_TRY(hipMemcpyAsync(flags_out, fl, size_t(n_rays), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

static int leaf_node_arg(rt_scene* s, int32_t node) {
    if (node < 0 || node >= s->ds.n_nodes || s->leaf_kind.empty() || s->leaf_kind[size_t(node)] > RT_SMOOTH_TRIANGLE) {
        rt_set_error("node %d is not a leaf (primitive) node of this scene", node);
        return RT_ERR_BAD_ARG;
    }
    return RT_OK;
}

int rt_normal_at(rt_scene* s, int32_t node, int64_t n, const double* points, const double* uv, double* normals_out) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (n < 0 || !points || !normals_out) { rt_set_error("null argument or negative count"); return RT_ERR_BAD_ARG; }
    int rc = leaf_node_arg(s, node);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    hipStream_t st = s->stream;
    const size_t rb = size_t(n) * 3 * sizeof(double);
    if ((rc = grow(s->in_a, rb)) != RT_OK || (rc = grow(s->in_b, size_t(n) * 2 * sizeof(double))) != RT_OK || (rc = grow(s->out_rgb, rb)) != RT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(s->in_a.p, points, rb, hipMemcpyHostToDevice, st));
    if (uv) HIP_TRY(hipMemcpyAsync(s->in_b.p, uv, size_t(n) * 2 * sizeof(double), hipMemcpyHostToDevice, st));
    if (s->general)
        hipLaunchKernelGGL(k_general_normal_at, dim3(blocks_for(n)), dim3(BLOCK), 0, st, s->ds, s->gs, (int)node, (long long)n,
                           static_cast<const double*>(s->in_a.p), uv ? static_cast<const double*>(s->in_b.p) : nullptr, static_cast<double*>(s->out_rgb.p));
    else
        hipLaunchKernelGGL(k_normal_at, dim3(blocks_for(n)), dim3(BLOCK), 0, st, s->ds, (int)node, (long long)n, static_cast<const double*>(s->in_a.p),
                           uv ? static_cast<const double*>(s->in_b.p) : nullptr, static_cast<double*>(s->out_rgb.p));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(normals_out, s->out_rgb.p, rb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_lighting(rt_scene* s, int32_t node, int32_t light, int64_t n, const double* points, const double* eyes, const double* normals,
                const uint8_t* in_shadow, double* rgb_out) {
    if (!check_scene(s)) return RT_ERR_BAD_ARG;
    if (n < 0 || !points || !eyes || !normals || !in_shadow || !rgb_out) { rt_set_error("null argument or negative count"); return RT_ERR_BAD_ARG; }
    int rc = leaf_node_arg(s, node);
    if (rc != RT_OK) return rc;
    if (light < 0 || light >= s->n_lights) { rt_set_error("light %d out of range (the world has %d)", light, s->n_lights); return RT_ERR_BAD_ARG; }
    if (n == 0) return RT_OK;
    hipStream_t st = s->stream;
    const size_t rb = size_t(n) * 3 * sizeof(double);
    if ((rc = grow(s->in_a, rb)) != RT_OK || (rc = grow(s->in_b, rb)) != RT_OK || (rc = grow(s->misc_b, rb)) != RT_OK ||
        (rc = grow(s->misc_a, size_t(n))) != RT_OK || (rc = grow(s->out_rgb, rb)) != RT_OK)
        return rc;
    HIP_TRY(hipMemcpyAsync(s->in_a.p, points, rb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->in_b.p, eyes, rb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->misc_b.p, normals, rb, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(s->misc_a.p, in_shadow, size_t(n), hipMemcpyHostToDevice, st));
    if (s->general)
        hipLaunchKernelGGL(k_general_lighting, dim3(blocks_for(n)), dim3(BLOCK), 0, st, s->ds, s->gs, (int)node, (int)light, (long long)n,
                           static_cast<const double*>(s->in_a.p), static_cast<const double*>(s->in_b.p), static_cast<const double*>(s->misc_b.p),
                           static_cast<const uint8_t*>(s->misc_a.p), static_cast<double*>(s->out_rgb.p));
    else
        hipLaunchKernelGGL(k_lighting, dim3(blocks_for(n)), dim3(BLOCK), 0, st, s->ds, (int)node, (int)light, (long long)n,
                           static_cast<const double*>(s->in_a.p), static_cast<const double*>(s->in_b.p), static_cast<const double*>(s->misc_b.p),
                           static_cast<const uint8_t*>(s->misc_a.p), static_cast<double*>(s->out_rgb.p));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgb_out, s->out_rgb.p, rb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RT_OK;
}

int rt_ray_for_pixel(const rt_camera_desc* cam, int64_t n, const int32_t* px, const int32_t* py, double* origins_out,
                     double* directions_out) {
    if (n < 0 || !px || !py || !origins_out || !directions_out) { rt_set_error("null argument or negative count"); return RT_ERR_BAD_ARG; }
    CameraParams c;
    long long n_slots = 0, n_valid = 0;
    int rc = camera_params(cam, 0, 1, c, n_slots, n_valid);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    if (rt_device_count() <= 0) { rt_set_error("no HIP device: this library has no CPU path"); return RT_ERR_NO_DEVICE; }
    HIP_TRY(hipSetDevice(0));
    size_t ib = size_t(n) * sizeof(int32_t), db = size_t(n) * 3 * sizeof(double);
    DeviceTemp dpx, dpy, dor, ddi;  // freed on every way out
    HIP_TRY(hipMalloc(&dpx.p, ib));
    HIP_TRY(hipMalloc(&dpy.p, ib));
    HIP_TRY(hipMalloc(&dor.p, db));
    HIP_TRY(hipMalloc(&ddi.p, db));
    HIP_TRY(hipMemcpy(dpx.p, px, ib, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dpy.p, py, ib, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_rays_for_pixels, dim3(blocks_for(n)), dim3(BLOCK), 0, nullptr, c, (long long)n, static_cast<const int32_t*>(dpx.p),
                       static_cast<const int32_t*>(dpy.p), static_cast<double*>(dor.p), static_cast<double*>(ddi.p));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(origins_out, dor.p, db, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(directions_out, ddi.p, db, hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // extern "C"
