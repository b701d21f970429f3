// librt_hip.so -- gfx950 kernels + the C ABI of include/rt_abi.h.
//
// Pipeline (one "pass" = every pixel x samples_per_pass camera samples in flight):
//   k_raygen  -> [ k_trace(path rays) -> k_shade -> k_trace(shadow rays) ] x max_depth -> k_unwind -> k_accumulate
// * k_trace is a WAVE-COHERENT walk of the trace stream (the threaded pre-order BVH re-packed into filter records): the 64
//   rays of a wave advance through it together; the cursor is wave-uniform, so records arrive by scalar loads (one fetch
//   per wave).  Each lane keeps only "the next record I need" (its private skip target).  Every test is first classified
//   by an fp32 filter with a proven error bound and only undecided lanes run the reference's f64 arithmetic; subtrees that
//   provably hold no triangle / sphere the reference would ACCEPT are skipped (strict pruning).  A lane tests a subset of
//   the primitives the reference tests, in the reference's order (bvh.rs:91-106) with the reference's running t_max, and
//   never skips an accepted one: the closest hit (primitive, t) is bit-identical (DESIGN.md section 5 has the argument).
// * k_shade evaluates one ray_colour level (integrator.rs) per live path, writes a bounce record,
//   the next ray and (MIS) the shadow ray, and compacts survivors into the next queue with a
//   wave64 ballot + popcount prefix and ONE atomic per block and queue.
// * k_unwind applies the records last-to-first (the MIS weight gate depends on the returned
//   sub-path radiance, integrator.rs:119-136, so a running throughput cannot express it).
#include "rt_scene.h"
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <limits>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>

using namespace rt;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIP_OK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess)                                                                                     \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------------
RT_DEV uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o < v ? o : v;
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
RT_DEV uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
RT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

#include "rt_trace.hip.h"
#include "rt_walk.hip.h"
#include "rt_shade.hip.h"
// ------------------------------------------------------------------------------------------------
// host side of the ABI
// ------------------------------------------------------------------------------------------------
// Scratch of rt_trace_closest_device_noalloc (the entry point for callers that cannot allocate): the statistics
// counters k_trace adds to, then one 64-byte ray block per ray (the layout k_trace reads).  Every GPU the library runs
// on holds its own copy of this array until the process ends.  In host code the NAME is only a stand-in: the device
// address is taken once per device with hipGetSymbolAddress (rt_context_create) and kept in rt_context::scratch.
#define RT_NOALLOC_MAX_RAYS (1u << 18)
#define RT_NOALLOC_STATS_BYTES 256u
static_assert(RT_NOALLOC_MAX_RAYS == RT_NOALLOC_MAX_RAYS_ABI, "the limit rt_abi.h documents is the size of the scratch array");
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_STATS_BYTES + (size_t)RT_NOALLOC_MAX_RAYS * RT_RAY_STRIDE * sizeof(double)];

// Timing events of one render lane: created once, reused by every call on that lane (creating and destroying two events per
// span cost a few microseconds each inside the timed region, and error paths leaked them), destroyed with the scene.
struct EventPool {
    std::vector<hipEvent_t> ev; size_t used = 0;
    hipEvent_t get() {
        if (used == ev.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            ev.push_back(e);
        }
        return ev[used++];
    }
    void reset() { used = 0; }
    void destroy() { for (auto e : ev) (void)hipEventDestroy(e); ev.clear(); used = 0; }
};
// Events of one call that are not timing spans (pass-order events, lane joins): destroyed on every way out of the call.
struct EventScope {
    std::vector<hipEvent_t> ev;
    ~EventScope() { for (auto e : ev) (void)hipEventDestroy(e); }
    hipError_t make(hipEvent_t* out, unsigned flags) {
        hipError_t r = hipEventCreateWithFlags(out, flags);
        if (r == hipSuccess) ev.push_back(*out);
        return r;
    }
};
struct EventTimer {
    struct Span { hipEvent_t a, b; int cat; hipStream_t st; };
    std::vector<Span> spans;
    hipStream_t s;
    bool enabled;
    EventPool* pool;
    EventTimer(hipStream_t st, bool en, EventPool* p) : s(st), enabled(en && p != nullptr), pool(p) {}
    ~EventTimer() { if (pool) pool->reset(); }   // the events go back to the pool whichever way the call ends
    void begin(int cat, hipStream_t on = nullptr) {   // `on`: the stream the span's launches go to (default: the timer's)
        if (!enabled) return;
        Span sp; sp.cat = cat; sp.a = pool->get(); sp.b = pool->get(); sp.st = on ? on : s;
        if (sp.a) (void)hipEventRecord(sp.a, sp.st);
        spans.push_back(sp);
    }
    void end() { if (enabled && spans.back().b) (void)hipEventRecord(spans.back().b, spans.back().st); }
    void collect(double* ms_by_cat, int ncat, uint64_t* count_by_cat) {
        for (auto& sp : spans) {
            float ms = 0.f;
            if (!sp.a || !sp.b || hipEventElapsedTime(&ms, sp.a, sp.b) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (sp.cat < ncat) { ms_by_cat[sp.cat] += ms; count_by_cat[sp.cat]++; }
        }
        spans.clear();
    }
};
struct rt_context {
    int device;
    hipStream_t stream;
    bool own_stream;
    unsigned char* scratch;  // this device's copy of `scratch` (device address)
    int live_scenes = 0;     // rt_scene handles that still point at this context (rt_context_destroy refuses while > 0)
};

#define RT_N_STATS 24          /* u64 counters per workspace: [0] rays, [1..3] box / triangle / sphere tests, [4..7] k_walk, [13] violations, [14] uncertified, [15] wave steps, [16] shadow rays */
#define RT_STAT_SHADOW_RAYS 16
#define RT_LAG 4u              /* run_paths, small queues: the host runs at most this many bounces ahead of the live counts it has seen */
#ifndef RT_SMALL_QUEUE
#define RT_SMALL_QUEUE (192u << 10)   /* rays: below this a traversal launch is shorter than a blocking read-back costs (see run_paths) */
#endif
#define RT_SMALL_QUEUE_TINY_SCENE (1u << 20)   /* the same for scenes of at most 4096 nodes */
#define RT_DEFAULT_PASS_BUDGET (80ull << 30)   /* two lanes + scene + certificate tables fit the 288 GB */
struct Workspace {
    PathBuffers pb{};
    uint32_t* queue[2] = {nullptr, nullptr};
    uint32_t* shadow_queue = nullptr;
    uint32_t* counts = nullptr;  // ring, zeroed once per pass: [2b] path rays entering bounce b, [2b + 1] shadow rays of bounce b
    hipEvent_t lag_ev[RT_LAG] = {};  // small queues: count copies in flight (run_paths)
    hipStream_t side = nullptr;      // shadow-ray traversals run here, beside the next bounce's path-ray traversal (run_paths)
    hipEvent_t ev_shaded = nullptr, ev_shadowed = nullptr;
    unsigned long long* stats = nullptr;  // rays, box, tri, sph
    uint32_t* in_pixel = nullptr; uint32_t* in_sample = nullptr;
    uint32_t* raw_queue = nullptr; uint32_t* raw_keys = nullptr; uint32_t* raw_squeue = nullptr; uint32_t* raw_skeys = nullptr; uint32_t* sorted_keys = nullptr;  // ray sorting
    void* sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
    uint32_t cap = 0; int32_t depth = 0;
    uint32_t rec_stride = 0;        // doubles per bounce record the records buffer was sized for (REC_STRIDE, or CREC_STRIDE for calls that can only write compact records)
    uint32_t* order = nullptr;      // material-binned shading order of the current queue (k_bin_scatter)
    uint8_t* cls = nullptr;         // class byte per queue slot (k_bin_count)
    uint32_t* bins = nullptr;       // [class][unit] counts, then their exclusive prefix (k_bin_count / k_bin_scan)
    uint4* walk_state = nullptr;    // k_walk: [RT_WALK_STATE_VEC][cap] per-ray filter state (k_walk_setup)
    uint32_t* walk_odd = nullptr;   // queue slots k_walk leaves to k_trace (rays that need the literal box test)
    uint32_t* walk_ctr = nullptr;   // [0] next queue slot to hand out, [1] entries of walk_odd
    uint32_t* h_counts = nullptr;  // pinned host copy of the counts ring (the live ray count of the next bounce sizes its sort and launches)
    std::vector<void*> allocs;
};

struct rt_scene {
    rt_context* ctx;
    void* dblob = nullptr;
    uint8_t* d_mat_uses_uv = nullptr;
    uint8_t* d_stream = nullptr;
    uint32_t* d_node_off = nullptr;
    rt_node* d_nodes_ext = nullptr;   // node array of a re-bracketed production stream (blob's nodes + the new scene-level range)
    uint8_t* d_stream_full = nullptr;
    uint32_t* d_node_off_full = nullptr;
    uint32_t* d_cert_cells = nullptr;
    CertEntry* d_cert_entries = nullptr;
    double* d_cert_normals = nullptr;
    double* d_cert_planes = nullptr;
    uint32_t* d_cert_plane_off = nullptr;
    uint32_t* d_cert_plane_leaf = nullptr;
    InstCull* d_inst_cull = nullptr; InstPlane* d_inst_planes = nullptr;
    SceneDev dev{};
    rt_blob_header hdr{};
    std::vector<rt_prim> host_prims;
    bool has_instances = false;
    bool reg_ray = false;   // most leaves are not triangles: k_trace variant with the f64 ray in registers
    bool sortable = false;  // scene has a finite box to quantise ray origins against
    bool walk_ok = false;   // k_walk (per-lane walk) may trace this scene's secondary and shadow rays
    bool bin_materials = false;  // more than one kind of scattering material: shade in material-binned order
    int n_cu = 256;
    Workspace ws;
    Workspace ws2;                  // second render lane (rt_render runs two passes at a time, each on its own stream)
    hipStream_t stream2 = nullptr;
    uint64_t ws2_failed_cap = ~0ull;  // smallest second-lane size (paths) that could not be allocated
    // rt_render_begin / rt_render_end: at most one job per lane in flight; jobs add to the accumulator in the order they were begun
    struct rt_job* lane_job[2] = {nullptr, nullptr};
    uint64_t job_seq = 0;               // jobs begun so far
    hipEvent_t job_done[4] = {nullptr, nullptr, nullptr, nullptr};   // job k records job_done[k % 4] after its last accumulate
    std::mutex job_mu; std::condition_variable job_cv; uint64_t jobs_recorded = 0;   // jobs whose done-event has been recorded (host side)
    std::thread worker[2];              // one host thread per lane, started by the first rt_render_begin on that lane (a fresh thread per job costs HIP's per-thread set-up every time)
    bool workers_quit = false;
    EventPool ev_pool[2];               // timing events per lane (EventTimer)
    uint64_t pass_budget = RT_DEFAULT_PASS_BUDGET;   // bytes of path state per lane that samples_per_pass == 0 may use (rt_scene_set_pass_budget; halved when an allocation fails)
};

struct rt_job {
    rt_scene* sc = nullptr; rt_render_params p{}; float* accum = nullptr; int lane = 0; uint64_t seq = 0; uint32_t spass = 1;
    bool started = false, finished = false;   // under rt_scene::job_mu
    int rc = RT_OK; std::string err; rt_stats st{};
};
static bool scene_busy(rt_scene* sc) { return sc->lane_job[0] != nullptr || sc->lane_job[1] != nullptr; }
#define RT_NOT_BUSY(sc, who) do { if (scene_busy(sc)) return fail(RT_ERR_BAD_ARG, who ": render jobs begun with rt_render_begin are still in flight on this scene (rt_render_end them first)"); } while (0)

static void ws_free(Workspace& ws) {
    for (void* p : ws.allocs) (void)hipFree(p);
    if (ws.h_counts) (void)hipHostFree(ws.h_counts);
    for (auto e : ws.lag_ev) if (e) (void)hipEventDestroy(e);
    if (ws.ev_shaded) (void)hipEventDestroy(ws.ev_shaded);
    if (ws.ev_shadowed) (void)hipEventDestroy(ws.ev_shadowed);
    if (ws.side) (void)hipStreamDestroy(ws.side);
    ws = Workspace();
}
template <class T> static int ws_alloc(Workspace& ws, T*& out, size_t count) {
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    ws.allocs.push_back(p);
    out = (T*)p;
    return RT_OK;
}
// compact: the call can only write 32-byte bounce records (no light sample can occur: run_paths) -- its records buffer is a third of the full one, so a
// pass of such a call may hold more samples in the same memory (bench.py sizes its jobs by that)
static int ws_ensure(rt_scene* sc, Workspace& ws, uint32_t cap, int32_t depth, bool compact = false);
static int ws_ensure(rt_scene* sc, uint32_t cap, int32_t depth) { return ws_ensure(sc, sc->ws, cap, depth); }
static int ws_ensure(rt_scene* sc, Workspace& ws, uint32_t cap, int32_t depth, bool compact) {
    const uint32_t rec_stride = compact ? (uint32_t)CREC_STRIDE : (uint32_t)REC_STRIDE;
    if (ws.cap >= cap && ws.depth >= depth && ws.rec_stride >= rec_stride) return RT_OK;
    ws_free(ws);
    cap = (cap + 255u) & ~255u;
    int rc;
#define WS(ptr, count) if ((rc = ws_alloc(ws, ptr, (size_t)(count))) != RT_OK) { ws_free(ws); return rc; }
    WS(ws.pb.rays, RT_RAY_STRIDE * (size_t)cap) WS(ws.pb.shadow_rays, RT_RAY_STRIDE * (size_t)cap)
    WS(ws.pb.hit_prim, cap) WS(ws.pb.hit_inst, cap) WS(ws.pb.hit_t, cap)
    WS(ws.pb.hit, cap) WS(ws.pb.info, cap)
    WS(ws.pb.terminal, 4 * (size_t)cap) WS(ws.pb.radiance, 3 * (size_t)cap)
    WS(ws.pb.records, (size_t)std::max(depth, 1) * rec_stride * cap)
    WS(ws.queue[0], cap) WS(ws.queue[1], cap) WS(ws.shadow_queue, cap) WS(ws.counts, 2 * ((size_t)std::max(depth, 1) + 2)) WS(ws.stats, RT_N_STATS)
    WS(ws.in_pixel, cap) WS(ws.in_sample, cap)
    WS(ws.raw_queue, cap) WS(ws.raw_keys, cap) WS(ws.raw_squeue, cap) WS(ws.raw_skeys, cap) WS(ws.sorted_keys, cap)
    {
        size_t bytes = 0;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, ws.raw_keys, ws.sorted_keys, ws.raw_queue, ws.queue[0], (size_t)cap, 0u, RT_SORT_BITS, sc->ctx->stream);
        if (e != hipSuccess) { ws_free(ws); return fail(RT_ERR_HIP, std::string("rocprim size query: ") + hipGetErrorString(e)); }
        uint8_t* tmp = nullptr;
        WS(tmp, bytes)
        ws.sort_tmp = tmp; ws.sort_tmp_bytes = bytes;
    }
#undef WS
    if (hipHostMalloc((void**)&ws.h_counts, 2 * ((size_t)std::max(depth, 1) + 2) * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { ws_free(ws); return fail(RT_ERR_OOM, "pinned host allocation failed"); }
    for (auto& e : ws.lag_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; ws_free(ws); return fail(RT_ERR_HIP, "event creation failed"); }
    if (hipStreamCreateWithFlags(&ws.side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ws.ev_shaded, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ws.ev_shadowed, hipEventDisableTiming) != hipSuccess) { ws_free(ws); return fail(RT_ERR_HIP, "side stream creation failed"); }
    ws.pb.cap = cap; ws.cap = cap; ws.depth = depth; ws.rec_stride = rec_stride;
    return RT_OK;
}

static inline dim3 grid_for(uint32_t n) { return dim3((n + 255u) / 256u); }
static int tune(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }
// Rays per wave of a traversal launch over a SMALL queue.  A wave walks the UNION of its rays' records and a launch lasts as long as its
// slowest wave; once a queue is so small that 64 rays per wave leave most of the machine's wave slots empty, fewer rays per wave make
// every walk shorter at no cost in throughput (large launches: 32 / 16 / 8 rays per wave measured 20-150 % slower).
// The kernel picks the number from the queue's true count (RT_LANES_AUTO); the host, which only knows a bound, sizes the grid for the worst case.
static uint32_t small_target_waves() { static const uint32_t v = (uint32_t)std::min(std::max(tune("RT_TUNE_SMALL_WAVES", 8192), 64), 1 << 20); return v; }
static uint32_t small_least_lanes() { static const uint32_t v = (uint32_t)std::min(std::max(tune("RT_TUNE_MIN_LANES", 1), 1), 64); return v; }
static uint32_t lanes_auto() { return RT_LANES_AUTO | (small_least_lanes() << 24) | small_target_waves(); }
static uint32_t waves_auto(uint32_t bound) {   // most waves any count <= bound needs: n / 64 while 64 rays per wave are kept, else about the target
    const uint64_t full = ((uint64_t)bound + 63u) / 64u, few = std::min<uint64_t>(((uint64_t)bound + small_least_lanes() - 1u) / small_least_lanes(), (uint64_t)small_target_waves() + 1u);
    return (uint32_t)std::max(full, few);
}

enum { CAT_TRACE = 0, CAT_SHADE = 1, CAT_OTHER = 2, CAT_SHADOW = 3, CAT_TAIL = 4, CAT_N = 5 };  // CAT_TAIL: all launches of the small-queue bounces of a pass, one span  // CAT_SHADOW: traversal launches of MIS shadow rays (integrator.rs:82)

static void launch_trace(rt_scene* sc, unsigned long long* lstats, uint32_t lcap, hipStream_t st, bool count, const uint32_t* queue, const uint32_t* n_ptr, uint32_t n_bound, const double* rays, double t_min,
                         double t_max, uint32_t* out_prim, uint32_t* out_inst, double* out_t, uint32_t lanes = 64u, uint32_t t_stride = 1u, PathHit* out_hit = nullptr, bool shadow = false,
                         const uint32_t* remap = nullptr, bool count_steps = false) {
    if (n_bound == 0) return;
    SceneDev dv = sc->dev;   // this launch's copy of the kernel argument
    dv.count_steps = count_steps ? 1u : 0u;
#ifdef RT_ORACLE_BEST
    if (tune("RT_TUNE_INST_EXPERIMENT", 0)) dv.oracle_best = (uint32_t)tune("RT_TUNE_INST_EXPERIMENT", 0);
#endif
    const uint32_t wpb = RT_TRACE_BLOCK / 64u;
    const uint64_t n_waves = (lanes & RT_LANES_AUTO) ? waves_auto(n_bound) : ((uint64_t)n_bound + lanes - 1u) / lanes;
    const dim3 grid((uint32_t)((n_waves + wpb - 1u) / wpb));  // waves of `lanes` rays, RT_TRACE_BLOCK / 64 waves per block
#define RT_LAUNCH_TRACE_S(C, I, R, S) hipLaunchKernelGGL((k_trace<C, I, R, S>), grid, dim3(RT_TRACE_BLOCK), 0, st, dv, queue, n_ptr, n_bound, rays, lcap, t_min, t_max, out_prim, out_inst, out_t, lstats, lanes, t_stride, out_hit, remap)
#define RT_LAUNCH_TRACE_R(C, I, R) do { if (shadow && !C) RT_LAUNCH_TRACE_S(C, I, R, !C); else RT_LAUNCH_TRACE_S(C, I, R, false); } while (0)
#define RT_LAUNCH_TRACE(C, I) do { if (sc->reg_ray) RT_LAUNCH_TRACE_R(C, I, true); else RT_LAUNCH_TRACE_R(C, I, false); } while (0)
    const bool inst = sc->has_instances;  // scenes without BVHNode objects use the kernel variant compiled without the nested walk
#ifdef RT_TRACE_DIAG
    unsigned long long d0[16];
    (void)hipMemcpyAsync(d0, lstats, sizeof d0, hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    static uint4* dlog = nullptr; static unsigned int dlog_cap = 0; static int dlog_launch = 0;
    const char* dlog_dir = getenv("RT_DIAG_LOG");
    if (dlog_dir && *dlog_dir && !count) {
        if (!dlog) { dlog_cap = 6u << 20; if (hipMalloc((void**)&dlog, (size_t)dlog_cap * sizeof(uint4)) != hipSuccess) { dlog = nullptr; dlog_cap = 0; } }
        const unsigned int zero = 0, every = (unsigned int)std::max(tune("RT_DIAG_LOG_EVERY", 400), 1);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dlog), &dlog, sizeof dlog); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dlog_n), &zero, sizeof zero);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dlog_cap), &dlog_cap, sizeof dlog_cap); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dlog_every), &every, sizeof every);
        const unsigned int group = (unsigned int)std::max(tune("RT_DIAG_LOG_GROUP", 64), 1);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dlog_group), &group, sizeof group);
        static float* dray = nullptr; static unsigned int dray_cap = 0;
        if (!dray) { dray_cap = 2u << 20; if (hipMalloc((void**)&dray, (size_t)dray_cap * 8u * sizeof(float)) != hipSuccess) { dray = nullptr; dray_cap = 0; } }
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dray), &dray, sizeof dray); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dray_n), &zero, sizeof zero);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dray_cap), &dray_cap, sizeof dray_cap);
    }
#endif
    if (count) { if (inst) RT_LAUNCH_TRACE(true, true); else RT_LAUNCH_TRACE(true, false); }
    else { if (inst) RT_LAUNCH_TRACE(false, true); else RT_LAUNCH_TRACE(false, false); }
#undef RT_LAUNCH_TRACE_S
#undef RT_LAUNCH_TRACE_R
#undef RT_LAUNCH_TRACE
#ifdef RT_TRACE_DIAG
    {
        unsigned long long d[16];
        (void)hipMemcpyAsync(d, lstats, sizeof d, hipMemcpyDeviceToHost, st);
        (void)hipStreamSynchronize(st);
        for (int k = 0; k < 16; ++k) if (k != 10) d[k] -= d0[k];
        (void)hipMemsetAsync(lstats + 10, 0, sizeof(unsigned long long), st);
        if (dlog_dir && *dlog_dir && !count && dlog) {   // the sampled waves' step log of this launch -> <dir>/launch_NNN_{p|s}.bin (tools/lane_log.py)
            unsigned int nlog = 0;
            (void)hipMemcpyFromSymbol(&nlog, HIP_SYMBOL(g_dlog_n), sizeof nlog);
            nlog = std::min(nlog, dlog_cap);
            std::vector<uint4> hl(nlog);
            if (nlog) (void)hipMemcpy(hl.data(), dlog, (size_t)nlog * sizeof(uint4), hipMemcpyDeviceToHost);
            char fn[512]; snprintf(fn, sizeof fn, "%s/launch_%03d_%c_%llu.bin", dlog_dir, dlog_launch++, shadow ? 's' : 'p', d[0]);
            if (FILE* f = fopen(fn, "wb")) { if (nlog) fwrite(hl.data(), sizeof(uint4), nlog, f); fclose(f); }
            unsigned int nray = 0, ray_cap = 0; float* dray_p = nullptr;
            (void)hipMemcpyFromSymbol(&nray, HIP_SYMBOL(g_dray_n), sizeof nray); (void)hipMemcpyFromSymbol(&ray_cap, HIP_SYMBOL(g_dray_cap), sizeof ray_cap);
            (void)hipMemcpyFromSymbol(&dray_p, HIP_SYMBOL(g_dray), sizeof dray_p);
            nray = std::min(nray, ray_cap);
            if (dray_p && nray) {   // the same waves' rays -> <dir>/rays_NNN_{p|s}_<rays>.bin
                std::vector<float> hr((size_t)nray * 8u);
                (void)hipMemcpy(hr.data(), dray_p, hr.size() * sizeof(float), hipMemcpyDeviceToHost);
                snprintf(fn, sizeof fn, "%s/rays_%03d_%c_%llu.bin", dlog_dir, dlog_launch - 1, shadow ? 's' : 'p', d[0]);
                if (FILE* f = fopen(fn, "wb")) { fwrite(hr.data(), sizeof(float), hr.size(), f); fclose(f); }
            }
        }
        const double w = (double)std::max(d[8], 1ull);
        fprintf(stderr, "[diag] rays %llu waves %llu  steps/wave %.0f  leaf steps/wave %.0f  lane use %.1f%%  cycles/wave %.0f  cycles/step %.0f  wait/step %.0f  max steps %llu  exact box steps/wave %.1f  exact tri steps/wave %.1f  | of a wave's cycles: set-up %.1f%%  leaf steps %.1f%% (exact tri blocks %.1f%%)  exact box blocks %.1f%%  scalar-load wait %.1f%%\n",
                d[0], d[8], d[4] / w, d[5] / w, 100.0 * d[9] / (64.0 * std::max(d[4], 1ull)), d[7] / w, (double)d[7] / std::max(d[4], 1ull), (double)d[6] / std::max(d[4], 1ull), d[10], d[11] / w, d[12] / w,
                100.0 * d[1] / std::max(d[7], 1ull), 100.0 * d[2] / std::max(d[7], 1ull), 100.0 * d[13] / std::max(d[7], 1ull), 100.0 * d[14] / std::max(d[7], 1ull), 100.0 * d[6] / std::max(d[7], 1ull));
    }
#endif
}

static void launch_trace(rt_scene* sc, bool count, const uint32_t* queue, const uint32_t* n_ptr, uint32_t n_bound, const double* rays, double t_min,
                         double t_max, uint32_t* out_prim, uint32_t* out_inst, double* out_t) {
    launch_trace(sc, sc->ws.stats, sc->ws.cap, sc->ctx->stream, count, queue, n_ptr, n_bound, rays, t_min, t_max, out_prim, out_inst, out_t);
}

// Secondary / shadow rays of scenes k_walk serves: per-ray set-up, the per-lane walk on a persistent grid, and a k_trace launch
// over the (normally zero) rays that need the literal box test.
static int launch_walk(rt_scene* sc, Workspace& ws, hipStream_t st, const uint32_t* queue, const uint32_t* n_ptr, uint32_t n_bound, const double* rays, double t_min,
                       double t_max, PathHit* out_hit, double* out_t, uint32_t t_stride, bool shadow) {
    if (n_bound == 0) return RT_OK;
    if (!ws.walk_state) {   // first use on this workspace: 160 B of state per path + the left-over list (freed with the workspace)
        int rc;
        if ((rc = ws_alloc(ws, ws.walk_state, (size_t)RT_WALK_STATE_VEC * ws.cap)) != RT_OK || (rc = ws_alloc(ws, ws.walk_odd, ws.cap)) != RT_OK ||
            (rc = ws_alloc(ws, ws.walk_ctr, 4)) != RT_OK) { ws.walk_state = nullptr; return rc; }
    }
    HIP_OK(hipMemsetAsync(ws.walk_ctr, 0, 4 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_walk_setup, grid_for(n_bound), dim3(256), 0, st, sc->dev, queue, n_ptr, rays, ws.walk_state, ws.cap, t_min, t_max, ws.walk_odd, ws.walk_ctr + 1, ws.stats);
    const uint32_t blocks = std::min<uint32_t>((n_bound + 255u) / 256u, (uint32_t)sc->n_cu * 4u);   // persistent: 4 blocks (16 waves) per CU
    if (shadow) hipLaunchKernelGGL((k_walk<true>), dim3(blocks), dim3(256), 0, st, sc->dev, queue, n_ptr, rays, ws.walk_state, ws.cap, t_min, t_max, out_hit, out_t, t_stride, ws.walk_ctr, ws.stats);
    else hipLaunchKernelGGL((k_walk<false>), dim3(blocks), dim3(256), 0, st, sc->dev, queue, n_ptr, rays, ws.walk_state, ws.cap, t_min, t_max, out_hit, out_t, t_stride, ws.walk_ctr, ws.stats);
    launch_trace(sc, ws.stats, ws.cap, st, false, queue, ws.walk_ctr + 1, n_bound, rays, t_min, t_max, nullptr, nullptr, out_t, 64u, t_stride, out_hit, shadow, ws.walk_odd);
    return RT_OK;
}

// Runs the wavefront loop for n paths whose (pixel, sample) come either from explicit device lists
// (ws.in_pixel / ws.in_sample) or from the frame formula; leaves per-path radiance in ws.pb.radiance.
// stable LSD radix sort of the n live (key, path) pairs of a queue
static int sort_queue(Workspace& ws, hipStream_t st, uint32_t n, uint32_t* keys_in, uint32_t* queue_in, uint32_t* out_queue) {
    size_t bytes = ws.sort_tmp_bytes;
    HIP_OK(rocprim::radix_sort_pairs(ws.sort_tmp, bytes, keys_in, ws.sorted_keys, queue_in, out_queue, (size_t)n, 0u, RT_SORT_BITS, st));
    return RT_OK;
}

// rt_capture_rays: copies out every stride-th ray of one queue of the render loop, with what the loop's own traversal launch reported for it
struct Capture {
    uint32_t bounce = 0; bool shadow = false; uint32_t stride = 1, max_n = 0;
    std::vector<double> rays, t, t_light; std::vector<uint32_t> prim, inst;
    uint32_t queue_len = 0;
};
static int capture_queue(Workspace& ws, hipStream_t st, const uint32_t* queue, uint32_t live, const double* rays_dev, const PathHit* hit_dev, const double* rec_dev, Capture& cap) {
    HIP_OK(hipStreamSynchronize(st));
    cap.queue_len = live;
    std::vector<uint32_t> q(live);
    HIP_OK(hipMemcpy(q.data(), queue, (size_t)live * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < live && cap.prim.size() < cap.max_n; k += cap.stride) {
        const uint32_t path = q[k];
        double r[8];
        HIP_OK(hipMemcpy(r, rays_dev + (size_t)path * RT_RAY_STRIDE, sizeof r, hipMemcpyDeviceToHost));
        for (int c = 0; c < 7; ++c) cap.rays.push_back(r[c]);
        if (hit_dev) {   // path rays: the hit record k_shade will consume (indexed by queue slot)
            PathHit ph;
            HIP_OK(hipMemcpy(&ph, hit_dev + k, sizeof ph, hipMemcpyDeviceToHost));
            cap.prim.push_back(ph.prim); cap.inst.push_back(ph.inst); cap.t.push_back(ph.t); cap.t_light.push_back(0.0);
        } else {         // shadow rays: t_light travels in the ray block, the walk's t goes to the bounce record
            double ts;
            HIP_OK(hipMemcpy(&ts, rec_dev + (size_t)path * REC_STRIDE, sizeof ts, hipMemcpyDeviceToHost));
            cap.prim.push_back(RT_NONE); cap.inst.push_back(RT_NONE); cap.t.push_back(ts); cap.t_light.push_back(r[7]);
        }
    }
    return RT_OK;
}

static int run_paths(rt_scene* sc, Workspace& ws, hipStream_t st, uint32_t n, bool explicit_lists, uint32_t width, uint32_t height, uint32_t sample0, int32_t max_depth,
                     int32_t integrator, uint64_t seed, bool count_tests, bool sort_rays, EventTimer& tm, uint64_t& launches, Capture* cap = nullptr,
                     bool lane_walk = false, bool want_sig = false, bool bins = false, bool count_steps = false, uint64_t* tail_bounces = nullptr, bool serial_shadow = false) {
    PathBuffers pbk = ws.pb;   // what the kernels see: signatures only for rt_sample_radiance_sig
    if (!want_sig) pbk.sig = nullptr;
    else HIP_OK(hipMemsetAsync(pbk.sig, 0, (size_t)n * sizeof(unsigned long long), st));
    const bool compact = !(integrator == RT_INT_MIS && sc->dev.n_lights > 0);   // no light sample can occur: 32-byte bounce records (rt_shade.hip.h)
    const bool use_walk = sc->walk_ok && !count_tests && lane_walk;   // opt-in (RT_RENDER_LANE_WALK): bounce >= 1 and shadow rays by the per-lane walk (k_walk)
    const uint32_t npix = width * height;
    if (max_depth > ws.depth && max_depth > 1) return fail(RT_ERR_BAD_ARG, "run_paths: workspace sized for a smaller depth");
    tm.begin(CAT_OTHER);
    // the counts ring starts at zero: k_shade of bounce b adds to [2b + 1] (shadow rays) and [2b + 2] (rays of bounce b + 1)
    HIP_OK(hipMemsetAsync(ws.counts, 0, 2 * ((size_t)max_depth + 1) * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_raygen, grid_for(n), dim3(256), 0, st, sc->dev.camera, width, height, n, sample0, npix,
                       explicit_lists ? ws.in_pixel : nullptr, explicit_lists ? ws.in_sample : nullptr, seed, ws.pb.rays, ws.cap, ws.pb.info,
                       explicit_lists ? 0u : 1u);
    hipLaunchKernelGGL(k_iota, grid_for(n), dim3(256), 0, st, ws.queue[0], n);
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, st, ws.counts + 0, n);
    tm.end();
    launches += 3;
    int cur = 0;
    // `live`: rays in queue[cur], or an upper bound of them.
    //  * Large queues: read back after every shade (one 8-byte copy and a stream sync per bounce), so that the sorts and launches of the
    //    next bounce are sized to the survivors, not to the pass.
    //  * Small queues (<= RT_SMALL_QUEUE rays: a traversal launch over them is shorter than that round trip costs -- the 10-spp frame
    //    of configs[0] is 50 bounces of a few thousand rays): no blocking read-back and no sort.  Every kernel takes its count from the
    //    device (n_ptr) and is launched over the bound; the counts are copied to the host asynchronously and the host stays at most RT_LAG
    //    bounces ahead of the newest count it has seen, which (counts only shrink) tightens the bound and ends the loop once a queue
    //    is empty.  A path's radiance does not depend on where in a queue it sits, so the frame is the same bit for bit.
    uint32_t live = n;
    int32_t lag_from = -1;   // first bounce run in the small-queue mode
    const bool may_lag = cap == nullptr && !use_walk && !(sc->bin_materials && bins);
    // Shadow rays (integrator.rs:82) are needed by k_unwind only: their traversal runs on the workspace's side stream, beside the next
    // bounce's path-ray traversal.  The next k_shade rewrites the per-path shadow-ray blocks, so it waits for that traversal.
    static const bool side_on = tune("RT_TUNE_SHADOW_SIDE", 1) != 0;
    // "small" = a traversal launch over the queue is shorter than the sort and the host round trip that would size it: scenes of a few hundred
    // nodes walk a few dozen records per ray (random spheres: 900 k unsorted rays in 0.15-0.4 ms against 0.4 ms for the two sorts alone)
    static const int small_tuned = tune("RT_TUNE_SMALL_QUEUE", 0);
    const uint32_t small_queue = small_tuned > 0 ? (uint32_t)small_tuned : sc->hdr.n_nodes <= 4096u ? RT_SMALL_QUEUE_TINY_SCENE : RT_SMALL_QUEUE;
    const bool side = side_on && !serial_shadow && cap == nullptr && !use_walk && !count_tests;
    bool shadow_pending = false;
    for (int32_t bounce = 0; bounce < max_depth && live > 0; ++bounce) {
        const int nxt = cur ^ 1;
        uint32_t* c_in = ws.counts + 2 * (size_t)bounce;
        uint32_t* c_shadow = c_in + 1;
        uint32_t* c_next = c_in + 2;
        const bool small = may_lag && live <= small_queue;
        if (small && lag_from < 0) { lag_from = bounce; tm.begin(CAT_TAIL); }   // from here on ONE span: an event pair per launch would cost more than the launches
        // every RT_LAG-th small bounce leaves its counts on the host; RT_LAG bounces later they have arrived (or are about to)
        if (small && bounce - lag_from >= (int32_t)RT_LAG && (bounce - lag_from) % (int32_t)RT_LAG == 0) {
            const int32_t j = bounce - (int32_t)RT_LAG;
            HIP_OK(hipEventSynchronize(ws.lag_ev[0]));
            live = std::min(live, ws.h_counts[2 * (size_t)j + 2]);   // rays that entered bounce j + 1 >= rays entering this one
            if (live == 0) break;
        }
        if (small && tail_bounces) *tail_bounces += 1;
        if (!small) tm.begin(CAT_TRACE);
        if (use_walk && bounce >= 1) { int rc = launch_walk(sc, ws, st, ws.queue[cur], c_in, live, ws.pb.rays, 0.001, 100000.0, ws.pb.hit, nullptr, 1u, false); if (rc != RT_OK) return rc; }
        else launch_trace(sc, ws.stats, ws.cap, st, count_tests, ws.queue[cur], c_in, live, ws.pb.rays, 0.001, 100000.0, nullptr, nullptr, nullptr, small ? lanes_auto() : 64u, 1u, ws.pb.hit, false, nullptr, count_steps);  // integrator.rs:38
        if (!small) tm.end();
#ifdef RT_ORACLE_BEST
        if (tune("RT_TUNE_ORACLE_BEST", 0)) {   // the same launch again, every walk starting from the hit distance the first one found
            SceneDev keep = sc->dev; sc->dev.oracle_best = 1u;
            launch_trace(sc, ws.stats, ws.cap, st, count_tests, ws.queue[cur], c_in, live, ws.pb.rays, 0.001, 100000.0, nullptr, nullptr, nullptr, 64u, 1u, ws.pb.hit, false, nullptr, count_steps);
            sc->dev = keep;
        }
#endif
        if (cap && !cap->shadow && (uint32_t)bounce == cap->bounce) return capture_queue(ws, st, ws.queue[cur], live, ws.pb.rays, ws.pb.hit, nullptr, *cap);
        const bool mis_shadow = integrator == RT_INT_MIS && sc->dev.n_lights > 0;
        const bool last_bounce = bounce + 1 >= max_depth;
        if (shadow_pending) { HIP_OK(hipStreamWaitEvent(st, ws.ev_shadowed, 0)); shadow_pending = false; }   // k_shade rewrites the shadow-ray blocks
        if (!small) tm.begin(CAT_SHADE);
        const uint32_t* order = nullptr;
        if (sc->bin_materials && bins) {   // opt-in (RT_RENDER_MATERIAL_BINS): regroup the queue's slots by the material class of their hits (compaction by material)
            if (!ws.order) {   // first use on this workspace (freed with it)
                int rc;
                if ((rc = ws_alloc(ws, ws.order, ws.cap)) != RT_OK || (rc = ws_alloc(ws, ws.cls, ws.cap)) != RT_OK ||
                    (rc = ws_alloc(ws, ws.bins, (size_t)RT_N_CLASSES * ((ws.cap + 511) / 512) + 16)) != RT_OK) return rc;
            }
            const uint32_t rounds = live > (8u << 20) ? 32u : 8u, nunits = (live + rounds * 64u - 1u) / (rounds * 64u);
            const dim3 ugrid((nunits + 3u) / 4u);
            hipLaunchKernelGGL(k_bin_count, ugrid, dim3(256), 0, st, sc->dev, ws.pb.hit, live, nunits, rounds, ws.cls, ws.bins);
            hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, st, ws.bins, RT_N_CLASSES * nunits);
            hipLaunchKernelGGL(k_bin_scatter, ugrid, dim3(256), 0, st, ws.cls, live, nunits, rounds, ws.bins, ws.order);
            order = ws.order;
            launches += 3;
        }
        // shade appends survivors compactly: to the raw queues with their coherence keys when the next bounce is sorted, else straight to the next queues
        const bool sorted = sort_rays && !small;
        const dim3 sgrid((live + RT_SHADE_BLOCK * RT_SHADE_TILES - 1u) / (RT_SHADE_BLOCK * RT_SHADE_TILES));
#define RT_LAUNCH_SHADE(KERNEL, NQ, NK, SQ, SK) hipLaunchKernelGGL(KERNEL, sgrid, dim3(RT_SHADE_BLOCK), 0, st, sc->dev, pbk, ws.queue[cur], order, c_in, (uint32_t)bounce, max_depth, integrator, \
                                                                   seed, NQ, NK, c_next, SQ, SK, c_shadow)
        // one build of the body per integrator (and record form): the MIS arm's light sample is dead weight in the three others
#define RT_SHADE_SORTED(C, I) RT_LAUNCH_SHADE((k_shade<C, I>), ws.raw_queue, ws.raw_keys, ws.raw_squeue, ws.raw_skeys)
#define RT_SHADE_DENSE(C, I) RT_LAUNCH_SHADE((k_shade_dense<C, I>), ws.queue[nxt], (uint32_t*)nullptr, ws.shadow_queue, (uint32_t*)nullptr)
#define RT_SHADE_PICK(WHICH) do { \
            if (integrator == RT_INT_MIS) { if (compact) WHICH(true, RT_INT_MIS); else WHICH(false, RT_INT_MIS); } \
            else if (integrator == RT_INT_LIGHT_IS) WHICH(true, RT_INT_LIGHT_IS); \
            else if (integrator == RT_INT_BRDF) WHICH(true, RT_INT_BRDF); \
            else WHICH(true, RT_INT_UNIFORM); } while (0)
        if (sorted) RT_SHADE_PICK(RT_SHADE_SORTED); else RT_SHADE_PICK(RT_SHADE_DENSE);
#undef RT_SHADE_PICK
#undef RT_SHADE_DENSE
#undef RT_SHADE_SORTED
#undef RT_LAUNCH_SHADE
        uint32_t next_live = live, shadow_live = live;   // small queues: bounds
        if (small) {
            if ((bounce - lag_from) % (int32_t)RT_LAG == 0) {
                HIP_OK(hipMemcpyAsync(ws.h_counts + 2 * (size_t)bounce + 1, c_shadow, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
                HIP_OK(hipEventRecord(ws.lag_ev[0], st));
            }
        } else {
            HIP_OK(hipMemcpyAsync(ws.h_counts + 2 * (size_t)bounce + 1, c_shadow, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            next_live = std::min(ws.h_counts[2 * (size_t)bounce + 2], n); shadow_live = std::min(ws.h_counts[2 * (size_t)bounce + 1], n);
        }
        const bool do_shadow = mis_shadow && shadow_live;
        if (sorted && do_shadow) { int rc = sort_queue(ws, st, shadow_live, ws.raw_skeys, ws.raw_squeue, ws.shadow_queue); if (rc != RT_OK) return rc; launches += 1; }   // the sorted order lands in shadow_queue
        hipStream_t sst = st;
        if (do_shadow && side) {   // everything the shadow traversal reads is in place on the main stream here
            HIP_OK(hipEventRecord(ws.ev_shaded, st));
            HIP_OK(hipStreamWaitEvent(ws.side, ws.ev_shaded, 0));
            sst = ws.side;
        }
        if (sorted && !last_bounce && next_live) { int rc = sort_queue(ws, st, next_live, ws.raw_keys, ws.raw_queue, ws.queue[nxt]); if (rc != RT_OK) return rc; launches += 1; }
        if (!small) tm.end();
        launches += 2;
        if (do_shadow) {
            if (!small) tm.begin(CAT_SHADOW, sst);
            double* t_shadow = ws.pb.records + (size_t)bounce * ws.cap * REC_STRIDE + REC_T_SHADOW;
            if (use_walk) { int rc = launch_walk(sc, ws, st, ws.shadow_queue, c_shadow, shadow_live, ws.pb.shadow_rays, 0.001, 10000.0, nullptr, t_shadow, (uint32_t)REC_STRIDE, true); if (rc != RT_OK) return rc; }
            else launch_trace(sc, ws.stats, ws.cap, sst, count_tests, ws.shadow_queue, c_shadow, shadow_live, ws.pb.shadow_rays, 0.001, 10000.0, nullptr, nullptr,
                              t_shadow, small ? lanes_auto() : 64u, (uint32_t)REC_STRIDE, nullptr, true, nullptr, count_steps);  // integrator.rs:82
            if (!small) tm.end();
            launches += 1;
            if (sst != st) { HIP_OK(hipEventRecord(ws.ev_shadowed, sst)); shadow_pending = true; }
            if (cap && cap->shadow && (uint32_t)bounce == cap->bounce)
                return capture_queue(ws, st, ws.shadow_queue, shadow_live, ws.pb.shadow_rays, nullptr, ws.pb.records + (size_t)bounce * ws.cap * REC_STRIDE + REC_T_SHADOW, *cap);
        }
        cur = nxt;
        live = next_live;
    }
    if (shadow_pending) HIP_OK(hipStreamWaitEvent(st, ws.ev_shadowed, 0));   // k_unwind reads the visibility results
    if (lag_from >= 0) tm.end();   // CAT_TAIL (the visibility launches on the side stream have been waited for)
    tm.begin(CAT_OTHER);
    if (compact) hipLaunchKernelGGL(k_unwind<true>, grid_for(n), dim3(256), 0, st, pbk, n); else hipLaunchKernelGGL(k_unwind<false>, grid_for(n), dim3(256), 0, st, pbk, n);
    hipLaunchKernelGGL(k_sum_shadow_counts, dim3(1), dim3(64), 0, st, ws.counts, (uint32_t)max_depth, ws.stats + RT_STAT_SHADOW_RAYS);   // rt_stats.shadow_rays, whichever mode the bounces ran in
    tm.end();
    launches += 2;
    HIP_OK(hipGetLastError());
    return RT_OK;
}


#include "rt_prune_host.hpp"
extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return n;
}

int rt_context_create(int device, void* stream, rt_context** out) {
    if (!out) return fail(RT_ERR_BAD_ARG, "rt_context_create: out is NULL");
    int n = 0;
    HIP_OK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(RT_ERR_BAD_ARG, "rt_context_create: no such device");
    HIP_OK(hipSetDevice(device));
    rt_context* c = new (std::nothrow) rt_context();
    if (!c) return fail(RT_ERR_OOM, "rt_context_create: host allocation failed");
    c->device = device;
    {   // device address of this GPU's copy of the no-alloc scratch array (the device is current: hipSetDevice above)
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) { delete c; return fail(RT_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e)); }
        c->scratch = (unsigned char*)ptr;
    }
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete c; return fail(RT_ERR_HIP, hipGetErrorString(e)); }
        c->own_stream = true;
    }
    *out = c;
    return RT_OK;
}
int rt_context_destroy(rt_context* ctx) {
    if (!ctx) return RT_OK;
    if (ctx->live_scenes > 0) return fail(RT_ERR_BAD_ARG, "rt_context_destroy: " + std::to_string(ctx->live_scenes) + " scene(s) of this context are still alive (destroy them first)");
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}
int rt_context_synchronize(rt_context* ctx) {
    if (!ctx) return fail(RT_ERR_BAD_ARG, "null context");
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

static bool range_ok(uint64_t off, uint64_t count, uint64_t elem, uint64_t total) {
    return off % 16 == 0 && off <= total && count * elem <= total - off;
}


// Everything rt_scene_upload derives from a blob without touching the device: validation of every index the kernels
// follow, and the trace stream.  Also behind rt_stream_build_host, so the CPU-only tests can check the stream.
struct ScenePrep {
    rt_blob_header h;
    bool boxes_finite = true, boxes_nested = false, reg_ray = false, has_instances = false, walk_ok = false;
    std::vector<uint8_t> uses_uv, stream;
    std::vector<uint32_t> rec_off, end_off;
    std::vector<uint8_t> stream_prod;  // production stream (see prepare_scene): the full one minus the always-entered inner nodes
    std::vector<uint32_t> rec_off_prod, end_off_prod;
    std::vector<rt_node> nodes_ext;    // re-bracketed production hierarchy (prepare_scene): the blob's nodes followed by the new scene-level range; empty = none
    uint32_t prod_end_off = 0;         // END record of the scene-level range in the production stream
    uint32_t prod_r0 = 0, prod_r1 = 0; // scene-level range of the production stream's node array
    std::vector<uint8_t> prod_elided;  // per node of that array: no record
    uint32_t root_off_prod = 0, n_elided = 0;
    double box_abs_max = 0.0;
    uint32_t root_off = 0;
    PruneHost prune;
    std::vector<InstCull> inst_cull; std::vector<InstPlane> inst_planes;   // scenes with BVHNode instances: whole-instance strict pruning (inst_line_misses)
};
static int prepare_scene(const void* blob, uint64_t nbytes, ScenePrep& P) {
    if (nbytes < sizeof(rt_blob_header)) return fail(RT_ERR_BAD_BLOB, "blob smaller than its header");
    rt_blob_header& h = P.h;
    memcpy(&h, blob, sizeof h);
    if (h.magic != RT_BLOB_MAGIC || h.version != RT_BLOB_VERSION) return fail(RT_ERR_BAD_BLOB, "bad blob magic/version");
    if (h.total_bytes != nbytes) return fail(RT_ERR_BAD_BLOB, "blob size mismatch");
    // operand-shape checks: every index the kernels follow is validated here, on the host
    if (!range_ok(h.off_nodes, h.n_nodes, sizeof(rt_node), nbytes) || !range_ok(h.off_prims, h.n_prims, sizeof(rt_prim), nbytes) ||
        !range_ok(h.off_geom, h.n_prims, sizeof(rt_geom), nbytes) || !range_ok(h.off_tri_attrs, h.n_tris, sizeof(rt_tri_attrs), nbytes) ||
        !range_ok(h.off_xforms, h.n_xforms, sizeof(rt_xform), nbytes) ||
        !range_ok(h.off_materials, h.n_materials, sizeof(rt_material), nbytes) || !range_ok(h.off_textures, h.n_textures, sizeof(rt_texture), nbytes) ||
        !range_ok(h.off_texels, h.n_texel_bytes, 1, nbytes) || !range_ok(h.off_lights, h.n_lights, sizeof(rt_light), nbytes))
        return fail(RT_ERR_BAD_BLOB, "table offset/size out of range");
    const uint8_t* b = (const uint8_t*)blob;
    const rt_node* nodes = (const rt_node*)(b + h.off_nodes);
    const rt_prim* prims = (const rt_prim*)(b + h.off_prims);
    const rt_geom* geom = (const rt_geom*)(b + h.off_geom);
    const rt_material* mats = (const rt_material*)(b + h.off_materials);
    const rt_texture* tex = (const rt_texture*)(b + h.off_textures);
    if (h.root_node > h.root_end || h.root_end > h.n_nodes) return fail(RT_ERR_BAD_BLOB, "root range out of bounds");
    bool& boxes_finite = P.boxes_finite;
    for (uint32_t i = 0; i < h.n_nodes; ++i) {
        for (int k = 0; k < 3 && !(nodes[i].flags & RT_NODE_NO_BOX); ++k)
            boxes_finite = boxes_finite && std::isfinite(nodes[i].bmin[k]) && std::isfinite(nodes[i].bmax[k]) && nodes[i].bmin[k] <= nodes[i].bmax[k];
        if (nodes[i].skip <= i || nodes[i].skip > h.n_nodes) return fail(RT_ERR_BAD_BLOB, "node skip link not strictly forward / out of range");
        if ((nodes[i].prim0 != RT_NONE && nodes[i].prim0 >= h.n_prims) || (nodes[i].prim1 != RT_NONE && nodes[i].prim1 >= h.n_prims))
            return fail(RT_ERR_BAD_BLOB, "node primitive index out of range");
        if (nodes[i].prim0 == RT_NONE && nodes[i].prim1 != RT_NONE) return fail(RT_ERR_BAD_BLOB, "node with prim1 but no prim0");
    }
    for (uint32_t i = 0; i < h.n_prims; ++i) {
        const rt_prim& p = prims[i];
        if (p.kind > RT_PRIM_INSTANCE) return fail(RT_ERR_BAD_BLOB, "primitive kind out of range");
        if ((p.kind == RT_PRIM_TRIANGLE && p.index >= h.n_tris) || (p.kind == RT_PRIM_INSTANCE && p.index >= h.n_nodes))
            return fail(RT_ERR_BAD_BLOB, "primitive index out of range");
        if (geom[i].kind != p.kind || geom[i].xform != p.xform) return fail(RT_ERR_BAD_BLOB, "geom record does not match its primitive");
        if (p.kind == RT_PRIM_INSTANCE && (geom[i].d[0] != (double)p.index || geom[i].d[1] != (double)p.sub_end))
            return fail(RT_ERR_BAD_BLOB, "instance geom range does not match its primitive");
        if (p.xform != RT_NONE && p.xform >= h.n_xforms) return fail(RT_ERR_BAD_BLOB, "primitive xform out of range");
        if (p.kind == RT_PRIM_INSTANCE) {
            if (p.sub_end <= p.index || p.sub_end > h.n_nodes || p.index < h.root_end) return fail(RT_ERR_BAD_BLOB, "instance sub-array out of range");
            for (uint32_t k = p.index; k < p.sub_end; ++k) {
                if (nodes[k].skip > p.sub_end) return fail(RT_ERR_BAD_BLOB, "instance skip link leaves its sub-array");
                for (uint32_t q : {nodes[k].prim0, nodes[k].prim1})
                    if (q != RT_NONE && (prims[q].kind == RT_PRIM_INSTANCE || prims[q].xform != RT_NONE))
                        return fail(RT_ERR_UNSUPPORTED, "nested instance / transformed primitive inside an instance");
            }
        } else if (p.material >= h.n_materials) return fail(RT_ERR_BAD_BLOB, "primitive material out of range");
    }
    for (uint32_t k = h.root_node; k < h.root_end; ++k)
        if (nodes[k].skip > h.root_end) return fail(RT_ERR_BAD_BLOB, "scene-level skip link leaves the scene array");
    std::vector<uint8_t>& uses_uv = P.uses_uv;
    uses_uv.assign(std::max<uint32_t>(h.n_materials, 1), 0);
    for (uint32_t i = 0; i < h.n_textures; ++i) {
        if (tex[i].kind > RT_TEX_IMAGE) return fail(RT_ERR_BAD_BLOB, "texture kind");
        if (tex[i].kind == RT_TEX_CHECKER && (tex[i].a >= h.n_textures || tex[i].b >= h.n_textures)) return fail(RT_ERR_BAD_BLOB, "checker child out of range");
        if (tex[i].kind == RT_TEX_IMAGE && (tex[i].b == 0 || tex[i].c == 0 || (uint64_t)tex[i].a + (uint64_t)tex[i].b * tex[i].c * 3 > h.n_texel_bytes))
            return fail(RT_ERR_BAD_BLOB, "image texture outside the texel pool");
    }
    for (uint32_t i = 0; i < h.n_materials; ++i) {
        if (mats[i].kind > RT_MAT_DIFFUSE_LIGHT) return fail(RT_ERR_BAD_BLOB, "material kind");
        bool needs_tex = mats[i].kind == RT_MAT_LAMBERTIAN || mats[i].kind == RT_MAT_DIFFUSE_LIGHT;
        if (needs_tex && mats[i].texture >= h.n_textures) return fail(RT_ERR_BAD_BLOB, "material texture out of range");
        if (needs_tex) {  // does the texture tree reach an image? (bounded walk; checker trees deeper than 8 are rejected)
            std::vector<std::pair<uint32_t, int>> stack{{mats[i].texture, 0}};
            while (!stack.empty()) {
                auto [t, dpt] = stack.back(); stack.pop_back();
                if (tex[t].kind == RT_TEX_IMAGE) uses_uv[i] = 1;
                else if (tex[t].kind == RT_TEX_CHECKER) {
                    if (dpt >= 8) return fail(RT_ERR_UNSUPPORTED, "checker texture nesting deeper than 8");
                    stack.push_back({tex[t].a, dpt + 1}); stack.push_back({tex[t].b, dpt + 1});
                }
            }
        }
    }
    const rt_light* lights = (const rt_light*)(b + h.off_lights);
    for (uint32_t i = 0; i < h.n_lights; ++i)
        if (lights[i].material >= h.n_materials) return fail(RT_ERR_BAD_BLOB, "light material out of range");

    for (uint32_t i = 0; i < h.n_prims; ++i) P.has_instances = P.has_instances || prims[i].kind == RT_PRIM_INSTANCE;
    {   // do the boxes of the scene-level array nest?  (children of node i: i + 1, then along the skip links up to skip(i))
        bool nested = boxes_finite;
        for (uint32_t i = h.root_node; i < h.root_end && nested; ++i) {
            const rt_node& a = nodes[i];
            if (a.flags & RT_NODE_NO_BOX) { nested = false; break; }
            if (a.prim0 != RT_NONE) continue;
            for (uint32_t c = i + 1; c < a.skip && nested; c = nodes[c].skip) {
                if (nodes[c].skip > a.skip) { nested = false; break; }
                for (int k = 0; k < 3; ++k) nested = nested && a.bmin[k] <= nodes[c].bmin[k] && nodes[c].bmax[k] <= a.bmax[k];
            }
        }
        P.boxes_nested = nested;
    }
    // trace stream: pre-order fp32 filter records (see k_trace)
    if (h.n_nodes >= RT_STREAM_MAX_NODES) { return fail(RT_ERR_UNSUPPORTED, "more than 2^24 BVH nodes"); }
    auto is_tri2 = [&](const rt_node& nd) {  // one triangle, or two under the same transform: leaf with a filter block
        if (nd.prim0 == RT_NONE || geom[nd.prim0].kind != RT_PRIM_TRIANGLE) return false;
        return nd.prim1 == RT_NONE || (geom[nd.prim1].kind == RT_PRIM_TRIANGLE && geom[nd.prim0].xform == geom[nd.prim1].xform);
    };
    auto is_sph2 = [&](const rt_node& nd) {  // one sphere, or two under the same transform
        if (nd.prim0 == RT_NONE || geom[nd.prim0].kind != RT_PRIM_SPHERE) return false;
        return nd.prim1 == RT_NONE || (geom[nd.prim1].kind == RT_PRIM_SPHERE && geom[nd.prim0].xform == geom[nd.prim1].xform);
    };
    build_prune(h, nodes, geom, is_tri2, is_sph2, P.prune);
    if (P.has_instances) build_inst_cull(h, nodes, prims, geom, P.inst_cull, P.inst_planes);
    if (tune("RT_TUNE_DEBUG", 0)) {
        size_t ok = 0; for (auto& c : P.inst_cull) ok += c.ok;
        fprintf(stderr, "[rt] instances with a cull box: %zu of %zu prims, %zu planes\n", ok, P.inst_cull.size(), P.inst_planes.size());
        for (auto& c : P.inst_cull) if (c.ok) fprintf(stderr, "[rt]   box %g %g %g .. %g %g %g tau_coef %g radius %g planes %u\n", c.lo[0], c.lo[1], c.lo[2], c.hi[0], c.hi[1], c.hi[2], c.tau_coef, c.radius, c.n_planes);
    }

    // ranges: the scene-level array and every instance sub-array (validated above: disjoint, links stay inside);
    // range_end[i] = end index of the range node i belongs to, ends_here[e] = some range ends before node e
    std::vector<uint32_t> range_end((size_t)h.n_nodes, h.n_nodes);
    std::vector<uint8_t> ends_here((size_t)h.n_nodes + 1, 0);
    for (uint32_t k = h.root_node; k < h.root_end; ++k) range_end[k] = h.root_end;
    ends_here[h.root_end] = 1;
    for (uint32_t i = 0; i < h.n_prims; ++i)
        if (prims[i].kind == RT_PRIM_INSTANCE) {
            for (uint32_t k = prims[i].index; k < prims[i].sub_end; ++k) range_end[k] = prims[i].sub_end;
            ends_here[prims[i].sub_end] = 1;
        }
    auto f32_up = [](double v) {  // smallest float >= v (NaN / overflow: +inf, which disables the filter for that record)
        if (!(v <= 3.4e38)) return std::numeric_limits<float>::infinity();
        float f = (float)v;
        if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
        return f;
    };
    double& box_abs_max = P.box_abs_max;
    // Lays the records out and fills them; `elide` marks inner nodes that get NO record (their children follow directly and
    // links to them land on their first child): see the production stream below.
    // (the node array it lays out is a parameter: the production stream may come from a re-bracketed copy, see below)
    const rt_node* all_nodes = nodes;
    auto emit = [&](const rt_node* nodes, uint32_t n_nodes, const std::vector<uint32_t>& range_end, const std::vector<uint8_t>& ends_here, const float* local_box,
                    const std::vector<uint8_t>& elide, std::vector<uint8_t>& stream, std::vector<uint32_t>& rec_off, std::vector<uint32_t>& end_off) -> int {
    rec_off.assign((size_t)n_nodes + 1, 0); end_off.assign((size_t)n_nodes + 1, 0);
    uint64_t stream_bytes = 0;
    for (uint32_t i = 0; i <= n_nodes; ++i) {
        if (ends_here[i]) { end_off[i] = (uint32_t)stream_bytes; stream_bytes += RT_STREAM_REC; }  // END record of the range(s) ending here
        rec_off[i] = (uint32_t)stream_bytes;
        if (i < n_nodes && !elide[i]) stream_bytes += is_tri2(nodes[i]) ? RT_STREAM_REC_TRI2 : is_sph2(nodes[i]) ? RT_STREAM_REC_SPH2 : nodes[i].prim0 != RT_NONE ? RT_STREAM_REC_LEAF : RT_STREAM_REC_INNER;
        if (stream_bytes + RT_STREAM_PAD > 0xFFFFFFF0ull) { return fail(RT_ERR_UNSUPPORTED, "trace stream larger than 4 GiB"); }
    }
    // a link to the end of the node's own range lands on that range's END record
    auto link_off = [&](uint32_t from, uint32_t to) { return to == range_end[from] ? end_off[to] : rec_off[to]; };
    stream.assign((size_t)stream_bytes + RT_STREAM_PAD, 0);
    for (uint32_t e = 0; e <= n_nodes; ++e) {
        if (!ends_here[e]) continue;
        const float inf = std::numeric_limits<float>::infinity();
        const float box[6] = {-inf, inf, -inf, inf, -inf, inf};
        const uint32_t words[2] = {end_off[e], RT_NODE_NO_BOX | RT_STREAM_LEAF | ((RT_STREAM_REC / 16u) << 4) | (RT_STREAM_END_NODE << 8)};
        memcpy(stream.data() + end_off[e], box, 24); memcpy(stream.data() + end_off[e] + 24, words, 8);
    }
    for (uint32_t i = 0; i < n_nodes; ++i) {
        if (elide[i]) continue;
        const rt_node& nd = nodes[i];
        uint8_t* rec = stream.data() + rec_off[i];
        const bool nb = (nd.flags & RT_NODE_NO_BOX) != 0, tri2 = is_tri2(nd), sph2 = is_sph2(nd);
        uint32_t fl = nd.flags & (RT_NODE_NO_BOX | RT_NODE_DUP_RIGHT);
        fl |= tri2 ? RT_STREAM_TRI2 : sph2 ? RT_STREAM_SPH2 : nd.prim0 != RT_NONE ? RT_STREAM_LEAF : RT_STREAM_INNER;
        const uint32_t rec_bytes = tri2 ? RT_STREAM_REC_TRI2 : sph2 ? RT_STREAM_REC_SPH2 : nd.prim0 != RT_NONE ? RT_STREAM_REC_LEAF : RT_STREAM_REC_INNER;
        float box[6];
        for (int k = 0; k < 3; ++k) {  // list items have no box in the reference: see the box filter
            box[2 * k] = nb ? -std::numeric_limits<float>::infinity() : (float)nd.bmin[k];
            box[2 * k + 1] = nb ? std::numeric_limits<float>::infinity() : (float)nd.bmax[k];
            if (!nb && std::isfinite(nd.bmin[k]) && std::isfinite(nd.bmax[k])) box_abs_max = std::max(box_abs_max, std::max(std::fabs(nd.bmin[k]), std::fabs(nd.bmax[k])));
        }
        const uint32_t words[2] = {link_off(i, nd.skip), fl | ((rec_bytes / 16u) << 4) | (i << 8)};
        memcpy(rec, box, 24); memcpy(rec + 24, words, 8);
        if (nd.prim0 == RT_NONE) memcpy(rec + RT_STREAM_REC, local_box + (size_t)i * 6, 24);  // inner: box of the subtree's triangles in their own space
        if (!tri2 && !sph2 && nd.prim0 != RT_NONE) {
            const uint32_t ids[4] = {nd.prim0, nd.prim1, geom[nd.prim0].xform, 0u};
            memcpy(rec + RT_STREAM_REC, ids, 16);
        }
        if (sph2) {
            const uint32_t head[2] = {geom[nd.prim0].xform, nd.prim1 != RT_NONE ? 2u : 1u};
            memcpy(rec + RT_STREAM_REC, head, 8);
            memcpy(rec + RT_STREAM_REC + 8, geom[nd.prim0].d, 32);  // centre xyz, radius
            if (nd.prim1 != RT_NONE) memcpy(rec + RT_STREAM_REC + 40, geom[nd.prim1].d, 32);
        }
        if (!tri2) continue;
        float blk[28] = {};
        const uint32_t xf = geom[nd.prim0].xform;
        memcpy(&blk[0], &xf, 4);
        for (int t = 0; t < 2; ++t) {  // pairs (A, B): e2, ml = v0 x e2, -n = -(e1 x e2), then the error weights (see the triangle filter)
            const rt_geom& g = geom[(t == 0 || nd.prim1 == RT_NONE) ? nd.prim0 : nd.prim1];  // a single triangle fills both halves
            const double* v0 = g.d; const double* e1 = g.d + 3; const double* e2 = g.d + 6;
            const double ml[3] = {v0[1] * e2[2] - v0[2] * e2[1], v0[2] * e2[0] - v0[0] * e2[2], v0[0] * e2[1] - v0[1] * e2[0]};
            const double nn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            auto n1 = [](const double* v) { return std::fabs(v[0]) + std::fabs(v[1]) + std::fabs(v[2]); };
            auto ninf = [](const double* v) { return std::max(std::max(std::fabs(v[0]), std::fabs(v[1])), std::fabs(v[2])); };
            for (int k = 0; k < 3; ++k) {
                blk[2 + 2 * k + t] = (float)e2[k];
                blk[2 + 2 * (3 + k) + t] = (float)ml[k];
                blk[2 + 2 * (6 + k) + t] = (float)-nn[k];
            }
            blk[2 + 2 * 9 + t] = f32_up(0x1p-20 * ninf(e2));
            blk[2 + 2 * 10 + t] = f32_up(0x1p-20 * n1(ml) + 0x1p-46 * ninf(v0) * n1(e2));
            blk[2 + 2 * 11 + t] = f32_up((1.0 + 0x1p-19) * (0x1p-20 * n1(nn) + 0x1p-46 * n1(e1) * n1(e2)));
        }
        memcpy(rec + RT_STREAM_REC, blk, sizeof blk);
    }
    return RT_OK;
    };
    // Full stream: every node has a record (the counting kernel and NaN-capable waves walk it).
    std::vector<uint8_t> elide((size_t)h.n_nodes, 0);
    { int rc = emit(all_nodes, h.n_nodes, range_end, ends_here, P.prune.local_box.data(), elide, P.stream, P.rec_off, P.end_off); if (rc != RT_OK) return rc; }
    P.root_off = h.root_node == h.root_end ? P.end_off[h.root_end] : P.rec_off[h.root_node];
    // Production stream.  Sound when the reference's boxes nest (checked above): every leaf re-decides with the reference's test on its own box, a leaf
    // box that passes with the current best implies that each of the reference's ancestor boxes (which contain it) passed with its larger best --
    // the slab values are monotone in the plane coordinates -- so the reference reaches exactly the leaves whose own box passes, in pre-order.  What
    // stands ABOVE the leaves in the stream is therefore free, as long as it never keeps a ray from a leaf whose box it passes: any bracketing of
    // the leaf sequence by boxes that enclose their leaves' boxes (and, for the strict test, local boxes that enclose their leaves' local boxes)
    // visits a superset of the reference's leaves in the reference's order.  Three uses of that freedom:
    //  (1) r03 / RT_REBRACKET (default on): the hierarchy above the scene-level leaves is rebuilt over the SAME leaf order, every split placed where
    //      area(left) * leaves(left) + area(right) * leaves(right) is smallest instead of at the median the reference takes (bvh.rs:45-66);
    //  (2) inner nodes that can never be pruned and whose box holds everything that can be (the ancestors of a 1000-unit ground sphere) get no record;
    //  (3) r03y: no record for an inner node whose box has at least RT_ELIDE_AREA (default 0.7) of the surface area of its nearest ancestor that
    //      kept one -- a wave pays one step for a record and, if any of its 64 rays passes, both children anyway; such boxes are passed by nearly
    //      every wave that reaches them (found with tools/elide_eval.py on the step log of a real pass).
    // NaN-capable waves and the counting kernel walk the full stream, which is the reference's tree record for record.
    const rt_node* pn = all_nodes;
    uint32_t pN = h.n_nodes, pr0 = h.root_node, pr1 = h.root_end;
    const float* plb = P.prune.local_box.data();
    std::vector<float> lb_ext;
    std::vector<uint32_t> p_range_end, leaf_ids;
    std::vector<uint8_t> p_ends_here;
    if (P.boxes_nested) for (uint32_t i = h.root_node; i < h.root_end; ++i) if (nodes[i].prim0 != RT_NONE) leaf_ids.push_back(i);
    {
        const char* rb = getenv("RT_REBRACKET");
        const size_t nL = leaf_ids.size();
        if (P.boxes_nested && nL >= 4 && !(rb && atoi(rb) == 0) && (uint64_t)h.n_nodes + 2 * (uint64_t)nL < RT_STREAM_MAX_NODES) {
            const uint32_t m = (uint32_t)(2 * nL - 1), base = h.n_nodes;
            P.nodes_ext.assign(nodes, nodes + h.n_nodes);
            P.nodes_ext.resize((size_t)base + m);
            lb_ext.assign(P.prune.local_box.begin(), P.prune.local_box.end());
            lb_ext.resize(((size_t)base + m) * 6);
            auto area3 = [](const double* lo, const double* hi) {
                const double x = std::max(hi[0] - lo[0], 0.0), y = std::max(hi[1] - lo[1], 0.0), z = std::max(hi[2] - lo[2], 0.0);
                return x * y + y * z + z * x;
            };
            std::vector<double> pre((size_t)nL * 6), suf((size_t)nL * 6);   // scratch: prefix / suffix boxes of the range being split
            struct Frame { uint32_t a, b, idx; };
            uint64_t sweep_work = 0;
            std::vector<Frame> todo;
            todo.push_back({0u, (uint32_t)nL, base});
            while (!todo.empty()) {
                const Frame f = todo.back(); todo.pop_back();
                rt_node& nd = P.nodes_ext[f.idx];
                const uint32_t n = f.b - f.a;
                if (n == 1) {   // the leaf itself, copied (records carry the copy's index; rec_off of the original is pointed at it below)
                    nd = nodes[leaf_ids[f.a]];
                    nd.skip = f.idx + 1;
                    continue;
                }
                for (uint32_t k = 0; k < n; ++k) {
                    const rt_node& lf = nodes[leaf_ids[f.a + k]];
                    for (int c = 0; c < 3; ++c) {
                        pre[(size_t)k * 6 + c] = k ? std::min(pre[(size_t)(k - 1) * 6 + c], lf.bmin[c]) : lf.bmin[c];
                        pre[(size_t)k * 6 + 3 + c] = k ? std::max(pre[(size_t)(k - 1) * 6 + 3 + c], lf.bmax[c]) : lf.bmax[c];
                    }
                }
                for (uint32_t k = n; k-- > 0;) {
                    const rt_node& lf = nodes[leaf_ids[f.a + k]];
                    for (int c = 0; c < 3; ++c) {
                        suf[(size_t)k * 6 + c] = k + 1 < n ? std::min(suf[(size_t)(k + 1) * 6 + c], lf.bmin[c]) : lf.bmin[c];
                        suf[(size_t)k * 6 + 3 + c] = k + 1 < n ? std::max(suf[(size_t)(k + 1) * 6 + 3 + c], lf.bmax[c]) : lf.bmax[c];
                    }
                }
                uint32_t best_s = n / 2; double best_c = INFINITY;
                const uint32_t stride = std::max(1u, n / 4096u);   // huge ranges: every stride-th position (the search stays O(n) per level either way)
                // one-sided splits make the tree deep and this loop O(n) per level: once the sweeps have cost 256 passes over the leaves, the rest of the
                // tree takes the median (what the reference does; the boxes above are still the unions)
                sweep_work += n;
                const bool search = sweep_work <= 256ull * (uint64_t)nL;
                for (uint32_t sp = 1; search && sp < n; sp += stride) {      // left = [a, a + sp), right = [a + sp, b)
                    const double c = area3(&pre[(size_t)(sp - 1) * 6], &pre[(size_t)(sp - 1) * 6 + 3]) * (double)sp + area3(&suf[(size_t)sp * 6], &suf[(size_t)sp * 6 + 3]) * (double)(n - sp);
                    if (c < best_c) { best_c = c; best_s = sp; }
                }
                for (int c = 0; c < 3; ++c) { nd.bmin[c] = pre[(size_t)(n - 1) * 6 + c]; nd.bmax[c] = pre[(size_t)(n - 1) * 6 + 3 + c]; }
                nd.skip = f.idx + 2 * n - 1; nd.flags = 0; nd.prim0 = RT_NONE; nd.prim1 = RT_NONE;
                todo.push_back({f.a + best_s, f.b, f.idx + 2 * best_s});   // right child after the left subtree's 2 s - 1 nodes
                todo.push_back({f.a, f.a + best_s, f.idx + 1});
            }
            // local boxes of the new inner nodes, bottom-up from the leaves' (never pruned unless every leaf below can be covered)
            const float inf = std::numeric_limits<float>::infinity();
            std::vector<uint8_t> okv(m, 0);
            std::vector<double> boxv((size_t)m * 6);
            double babs = P.prune.box_abs_max;
            {
                std::vector<uint32_t> copy_leaf(m, RT_NONE);   // virtual index -> position in leaf_ids
                uint32_t seen = 0;
                for (uint32_t v = 0; v < m; ++v) if (P.nodes_ext[base + v].prim0 != RT_NONE) copy_leaf[v] = seen++;
                for (uint32_t v = m; v-- > 0;) {
                    const rt_node& nd = P.nodes_ext[base + v];
                    bool good = !P.prune.node_ok.empty();
                    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                    if (nd.prim0 != RT_NONE) {
                        const uint32_t orig = leaf_ids[copy_leaf[v]] - h.root_node;
                        good = good && P.prune.node_ok[orig] != 0;
                        for (int k = 0; k < 3 && good; ++k) { lo[k] = P.prune.node_box[(size_t)orig * 6 + 2 * k]; hi[k] = P.prune.node_box[(size_t)orig * 6 + 2 * k + 1]; }
                    } else {
                        for (uint32_t c = base + v + 1; good && c < nd.skip; c = P.nodes_ext[c].skip) {
                            good = okv[c - base] != 0;
                            for (int k = 0; k < 3 && good; ++k) { lo[k] = std::min(lo[k], boxv[(size_t)(c - base) * 6 + 2 * k]); hi[k] = std::max(hi[k], boxv[(size_t)(c - base) * 6 + 2 * k + 1]); }
                        }
                    }
                    okv[v] = good ? 1 : 0;
                    for (int k = 0; k < 3; ++k) {
                        boxv[(size_t)v * 6 + 2 * k] = lo[k]; boxv[(size_t)v * 6 + 2 * k + 1] = hi[k];
                        const bool carry = good && nd.prim0 == RT_NONE && P.prune.on;
                        lb_ext[((size_t)base + v) * 6 + 2 * k] = carry ? f32_down_host(lo[k]) : -inf;
                        lb_ext[((size_t)base + v) * 6 + 2 * k + 1] = carry ? f32_up_host(hi[k]) : inf;
                        if (carry) babs = std::max(babs, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
                    }
                }
            }
            P.prune.box_abs_max = babs;   // the strict filter's rounding margin is stated against the largest local coordinate in any record
            pn = P.nodes_ext.data(); pN = base + m; pr0 = base; pr1 = base + m; plb = lb_ext.data();
            p_range_end.assign((size_t)pN, pN);
            for (uint32_t k = 0; k < h.n_nodes; ++k) p_range_end[k] = range_end[k];
            p_ends_here.assign((size_t)pN + 1, 0);
            for (uint32_t k = 0; k <= h.n_nodes; ++k) p_ends_here[k] = ends_here[k];
            p_ends_here[h.root_end] = 0;   // the reference's scene-level range has no records in this stream ...
            for (uint32_t i2 = 0; i2 < h.n_prims; ++i2) if (prims[i2].kind == RT_PRIM_INSTANCE && prims[i2].sub_end == h.root_end) p_ends_here[h.root_end] = 1;   // ... unless an instance range ends there too
            p_ends_here[pN] = 1;
            elide.assign((size_t)pN, 0);
            for (uint32_t k = h.root_node; k < h.root_end; ++k) elide[k] = 1;
        } else {
            p_range_end = range_end; p_ends_here = ends_here;
        }
    }
    if (P.prune.on && P.boxes_nested) {   // rule (2)
        double core_lo[3] = {INFINITY, INFINITY, INFINITY}, core_hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint32_t i = pr0; i < pr1; ++i)
            if (pn[i].prim0 == RT_NONE && std::isfinite(plb[(size_t)i * 6]))
                for (int k = 0; k < 3; ++k) { core_lo[k] = std::min(core_lo[k], pn[i].bmin[k]); core_hi[k] = std::max(core_hi[k], pn[i].bmax[k]); }
        for (uint32_t i = pr0; i < pr1; ++i) {
            if (pn[i].prim0 != RT_NONE || std::isfinite(plb[(size_t)i * 6]) || !(pn[i].skip > i + 1)) continue;
            bool holds = true;
            for (int k = 0; k < 3; ++k) holds = holds && pn[i].bmin[k] <= core_lo[k] && core_hi[k] <= pn[i].bmax[k];
            if (holds) { elide[i] = 1; ++P.n_elided; }
        }
    }
    if (P.boxes_nested) {   // rule (3)
        const char* qs = getenv("RT_ELIDE_AREA");
        const double q = qs ? atof(qs) : 0.7;
        if (q > 0.0) {
            auto half_area = [&](uint32_t i) {
                const double x = std::max(pn[i].bmax[0] - pn[i].bmin[0], 0.0), y = std::max(pn[i].bmax[1] - pn[i].bmin[1], 0.0),
                             z = std::max(pn[i].bmax[2] - pn[i].bmin[2], 0.0);
                return x * y + y * z + z * x;
            };
            std::vector<double> kept_area((size_t)pN, INFINITY);   // area of the nearest ancestor with a record (pre-order: parents come first)
            for (uint32_t i = pr0; i < pr1; ++i) {
                if (pn[i].prim0 != RT_NONE) continue;
                const double own = half_area(i);
                if (!elide[i] && pn[i].skip > i + 1 && std::isfinite(kept_area[i]) && own >= q * kept_area[i]) { elide[i] = 1; ++P.n_elided; }
                const double down = elide[i] ? kept_area[i] : own;
                for (uint32_t c = i + 1; c < pn[i].skip && c < pr1; c = pn[c].skip) kept_area[c] = down;
            }
        }
    }
    { int rc = emit(pn, pN, p_range_end, p_ends_here, plb, elide, P.stream_prod, P.rec_off_prod, P.end_off_prod); if (rc != RT_OK) return rc; }
    P.root_off_prod = pr0 == pr1 ? P.end_off_prod[pr1] : P.rec_off_prod[pr0];
    P.prod_end_off = P.end_off_prod[pr1];
    P.prod_r0 = pr0; P.prod_r1 = pr1; P.prod_elided = elide;
    if (!P.nodes_ext.empty()) {   // the certificate names leaves by the blob's node index: point those at the copies' records
        uint32_t seen = 0;
        for (uint32_t v = pr0; v < pr1; ++v) if (pn[v].prim0 != RT_NONE) P.rec_off_prod[leaf_ids[seen++]] = P.rec_off_prod[v];
    }
    static_assert(RT_STREAM_REC + 28 * 4 == RT_STREAM_REC_TRI2, "TRI2 record layout");
    {
        uint64_t n_leaf = 0, n_tri_leaf = 0;
        for (uint32_t i = 0; i < h.n_nodes; ++i) { n_leaf += nodes[i].prim0 != RT_NONE; n_tri_leaf += is_tri2(nodes[i]); }
        P.reg_ray = 2 * n_tri_leaf < n_leaf;
    }
    {   // k_walk (per-lane walk) serves scenes with a pruned space, no BVHNode instances, and every triangle leaf in that space
        bool ok = P.prune.on && !P.has_instances && P.boxes_finite && h.root_end > h.root_node;
        for (uint32_t i = h.root_node; i < h.root_end && ok; ++i)
            if (is_tri2(nodes[i]) && geom[nodes[i].prim0].xform != P.prune.xf) ok = false;
        P.walk_ok = ok;
    }
    return RT_OK;
}

int rt_stream_build_host(const void* blob, uint64_t nbytes, void* stream_out, uint64_t capacity, uint64_t* stream_bytes, uint32_t* root_off, uint32_t* node_off) {
    if (!blob || !stream_bytes) return fail(RT_ERR_BAD_ARG, "rt_stream_build_host: null argument");
    ScenePrep P;
    int rc = prepare_scene(blob, nbytes, P);
    if (rc != RT_OK) return rc;
    *stream_bytes = P.stream.size();
    if (root_off) *root_off = P.root_off;
    if (stream_out) {
        if (capacity < P.stream.size()) return fail(RT_ERR_BAD_ARG, "rt_stream_build_host: buffer too small");
        memcpy(stream_out, P.stream.data(), P.stream.size());
    }
    if (node_off) memcpy(node_off, P.rec_off.data(), P.rec_off.size() * sizeof(uint32_t));
    return RT_OK;
}

int rt_stream_build_prod_host(const void* blob, uint64_t nbytes, void* stream_out, uint64_t capacity, uint64_t* stream_bytes, uint32_t* root_off, uint32_t* node_off) {
    if (!blob || !stream_bytes) return fail(RT_ERR_BAD_ARG, "rt_stream_build_prod_host: null argument");
    ScenePrep P;
    int rc = prepare_scene(blob, nbytes, P);
    if (rc != RT_OK) return rc;
    *stream_bytes = P.stream_prod.size();
    if (root_off) *root_off = P.root_off_prod;
    if (stream_out) {
        if (capacity < P.stream_prod.size()) return fail(RT_ERR_BAD_ARG, "rt_stream_build_prod_host: buffer too small");
        memcpy(stream_out, P.stream_prod.data(), P.stream_prod.size());
    }
    if (node_off) memcpy(node_off, P.rec_off_prod.data(), ((size_t)P.h.n_nodes + 1) * sizeof(uint32_t));   // by the blob's node index (a re-bracketed stream has more nodes: rt_stream_nodes_prod_host)
    return RT_OK;
}

int rt_stream_nodes_prod_host(const void* blob, uint64_t nbytes, rt_node* nodes_out, uint32_t capacity_nodes, uint32_t* n_nodes_out, uint32_t* range_begin, uint32_t* range_end,
                              uint8_t* elided_out) {
    if (!blob || !n_nodes_out) return fail(RT_ERR_BAD_ARG, "rt_stream_nodes_prod_host: null argument");
    ScenePrep P;
    int rc = prepare_scene(blob, nbytes, P);
    if (rc != RT_OK) return rc;
    const rt_node* src = P.nodes_ext.empty() ? (const rt_node*)((const uint8_t*)blob + P.h.off_nodes) : P.nodes_ext.data();
    const uint32_t n = P.nodes_ext.empty() ? P.h.n_nodes : (uint32_t)P.nodes_ext.size();
    *n_nodes_out = n;
    if (range_begin) *range_begin = P.prod_r0;
    if (range_end) *range_end = P.prod_r1;
    if (nodes_out) {
        if (capacity_nodes < n) return fail(RT_ERR_BAD_ARG, "rt_stream_nodes_prod_host: buffer too small");
        memcpy(nodes_out, src, (size_t)n * sizeof(rt_node));
        if (elided_out) memcpy(elided_out, P.prod_elided.data(), n);
    }
    return RT_OK;
}

int rt_prune_info_host(const void* blob, uint64_t nbytes, double* info, uint32_t n_rays, const double* rays, uint8_t* certified) {
    if (!blob || !info) return fail(RT_ERR_BAD_ARG, "rt_prune_info_host: null argument");
    ScenePrep P;
    int rc = prepare_scene(blob, nbytes, P);
    if (rc != RT_OK) return rc;
    const PruneHost& R = P.prune;
    info[0] = R.on ? 1.0 : 0.0; info[1] = (double)R.xf; info[2] = (double)R.n_tris; info[3] = (double)R.n_prunable_nodes;
    info[4] = (double)(R.normals.size() / 3); info[5] = (double)R.entries.size(); info[6] = R.shape_min; info[7] = R.tau_coef;
    info[8] = R.delta_coef; info[9] = R.radius; info[10] = R.box_abs_max; info[11] = (double)P.n_elided;
    if (n_rays && (!rays || !certified)) return fail(RT_ERR_BAD_ARG, "rt_prune_info_host: null ray arguments");
    // the device's certificate, restated on the host for rays (o, d) ALREADY in the pruned space (tests compare it with a brute-force scan)
    for (uint32_t i = 0; i < n_rays; ++i) {
        const double* o = rays + 6 * (size_t)i; const double* d = o + 3;
        certified[i] = 0;
        if (!R.on) continue;
        const double ax = std::fabs(d[0]), ay = std::fabs(d[1]), az = std::fabs(d[2]);
        const double D = std::max(std::max(ax, ay), az), O = std::max(std::max(std::fabs(o[0]), std::fabs(o[1])), std::fabs(o[2]));
        if (!(D >= 0x1p-100 && D <= 0x1p100 && O <= 0x1p100)) continue;
        const int a = ax >= ay ? (ax >= az ? 0 : 2) : (ay >= az ? 1 : 2);
        const double u = d[(a + 1) % 3] / d[a], v = d[(a + 2) % 3] / d[a];
        const double kk = (double)RT_CERT_K * 0.5, top = (double)(RT_CERT_K - 1u);
        const uint32_t iu = (uint32_t)std::fmin(std::fmax(std::floor((u + 1.0) * kk), 0.0), top), iv = (uint32_t)std::fmin(std::fmax(std::floor((v + 1.0) * kk), 0.0), top);
        const uint32_t cell = ((uint32_t)a * RT_CERT_K + iv) * RT_CERT_K + iu;
        const double thr = RT_CERT_EPS * D, delta = R.delta_coef * (O + R.radius);
        bool ok = true;
        std::vector<uint32_t> sus;
        for (uint32_t e = R.cell_off[cell]; e < R.cell_off[cell + 1]; ++e) {
            const uint32_t c = R.entries[e].cluster;
            const double* n = &R.normals[3 * (size_t)c];
            if (std::fabs(d[0] * n[0] + d[1] * n[1] + d[2] * n[2]) >= thr) continue;
            const double q = o[0] * n[0] + o[1] * n[1] + o[2] * n[2];
            auto b = R.planes.begin() + R.plane_off[c], e2 = R.planes.begin() + R.plane_off[c + 1];
            for (auto it = std::lower_bound(b, e2, q - delta); it != e2 && !(*it > q + delta); ++it) {
                const uint32_t leaf = R.plane_leaf[(size_t)(it - R.planes.begin())];
                if (std::find(sus.begin(), sus.end(), leaf) == sus.end()) sus.push_back(leaf);
            }
        }
        ok = sus.size() <= RT_SUSPECT_SLOTS;
        certified[i] = !ok ? 0 : sus.empty() ? 1 : 2;
    }
    return RT_OK;
}

int rt_scene_upload(rt_context* ctx, const void* blob, uint64_t nbytes, rt_scene** out) {
    if (!ctx || !blob || !out) return fail(RT_ERR_BAD_ARG, "rt_scene_upload: null argument");
    ScenePrep P;
    {
        int rc = prepare_scene(blob, nbytes, P);
        if (rc != RT_OK) return rc;
    }
    const rt_blob_header& h = P.h;
    const uint8_t* b = (const uint8_t*)blob;
    const rt_node* nodes = (const rt_node*)(b + h.off_nodes);
    const rt_prim* prims = (const rt_prim*)(b + h.off_prims);
    const std::vector<uint8_t>& uses_uv = P.uses_uv; const std::vector<uint8_t>& stream = P.stream;
    const std::vector<uint32_t>& rec_off = P.rec_off;
    const bool boxes_finite = P.boxes_finite; const double box_abs_max = P.box_abs_max;
    HIP_OK(hipSetDevice(ctx->device));
    rt_scene* sc = new (std::nothrow) rt_scene();
    if (!sc) return fail(RT_ERR_OOM, "host allocation failed");
    sc->ctx = ctx; sc->hdr = h;
    ctx->live_scenes += 1;
    sc->host_prims.assign(prims, prims + h.n_prims);
    sc->has_instances = P.has_instances; sc->reg_ray = P.reg_ray;
    {   // kinds of scattering material in the scene (DiffuseLight terminates a path; it is a class of its own but no reason to bin)
        const rt_material* mats = (const rt_material*)(b + h.off_materials);
        uint32_t kinds = 0;
        for (uint32_t i = 0; i < h.n_materials; ++i) if (mats[i].kind != RT_MAT_DIFFUSE_LIGHT) kinds |= 1u << mats[i].kind;
        sc->bin_materials = __builtin_popcount(kinds) > 1;
    }
#ifndef RT_TRACE_DIAG   /* the diagnostic build reports k_trace's own step counts */
    sc->walk_ok = P.walk_ok;
#endif
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) sc->n_cu = prop.multiProcessorCount; }
    hipError_t e = hipMalloc(&sc->dblob, nbytes);
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_mat_uses_uv, uses_uv.size());
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_stream, P.stream_prod.size());
    if (e == hipSuccess) e = hipMemcpy(sc->d_stream, P.stream_prod.data(), P.stream_prod.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_node_off, P.rec_off_prod.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(sc->d_node_off, P.rec_off_prod.data(), P.rec_off_prod.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_stream_full, stream.size());
    if (e == hipSuccess) e = hipMemcpy(sc->d_stream_full, stream.data(), stream.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void**)&sc->d_node_off_full, rec_off.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(sc->d_node_off_full, rec_off.data(), rec_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(sc->dblob, blob, nbytes, hipMemcpyHostToDevice);
    if (e == hipSuccess && !P.nodes_ext.empty()) e = hipMalloc((void**)&sc->d_nodes_ext, P.nodes_ext.size() * sizeof(rt_node));
    if (e == hipSuccess && !P.nodes_ext.empty()) e = hipMemcpy(sc->d_nodes_ext, P.nodes_ext.data(), P.nodes_ext.size() * sizeof(rt_node), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(sc->d_mat_uses_uv, uses_uv.data(), uses_uv.size(), hipMemcpyHostToDevice);
    const PruneHost& pr = P.prune;
    if (pr.on) {
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_cells, pr.cell_off.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_cells, pr.cell_off.data(), pr.cell_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_entries, std::max<size_t>(pr.entries.size(), 1) * sizeof(CertEntry));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_entries, pr.entries.data(), pr.entries.size() * sizeof(CertEntry), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_normals, std::max<size_t>(pr.normals.size(), 1) * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_normals, pr.normals.data(), pr.normals.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_planes, std::max<size_t>(pr.planes.size(), 1) * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_planes, pr.planes.data(), pr.planes.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_plane_off, pr.plane_off.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_plane_off, pr.plane_off.data(), pr.plane_off.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_cert_plane_leaf, std::max<size_t>(pr.plane_leaf.size(), 1) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemcpy(sc->d_cert_plane_leaf, pr.plane_leaf.data(), pr.plane_leaf.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    }
    if (!P.inst_cull.empty()) {
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_inst_cull, P.inst_cull.size() * sizeof(InstCull));
        if (e == hipSuccess) e = hipMemcpy(sc->d_inst_cull, P.inst_cull.data(), P.inst_cull.size() * sizeof(InstCull), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&sc->d_inst_planes, std::max<size_t>(P.inst_planes.size(), 1) * sizeof(InstPlane));
        if (e == hipSuccess) e = hipMemcpy(sc->d_inst_planes, P.inst_planes.data(), P.inst_planes.size() * sizeof(InstPlane), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        if (sc->d_inst_cull) (void)hipFree(sc->d_inst_cull);
        if (sc->d_inst_planes) (void)hipFree(sc->d_inst_planes);
        if (sc->dblob) (void)hipFree(sc->dblob);
        if (sc->d_mat_uses_uv) (void)hipFree(sc->d_mat_uses_uv);
        if (sc->d_stream) (void)hipFree(sc->d_stream);
        if (sc->d_node_off) (void)hipFree(sc->d_node_off);
        if (sc->d_nodes_ext) (void)hipFree(sc->d_nodes_ext);
        if (sc->d_stream_full) (void)hipFree(sc->d_stream_full);
        if (sc->d_node_off_full) (void)hipFree(sc->d_node_off_full);
        if (sc->d_cert_cells) (void)hipFree(sc->d_cert_cells);
        if (sc->d_cert_entries) (void)hipFree(sc->d_cert_entries);
        if (sc->d_cert_normals) (void)hipFree(sc->d_cert_normals);
        if (sc->d_cert_planes) (void)hipFree(sc->d_cert_planes);
        if (sc->d_cert_plane_off) (void)hipFree(sc->d_cert_plane_off);
        if (sc->d_cert_plane_leaf) (void)hipFree(sc->d_cert_plane_leaf);
        ctx->live_scenes -= 1;
        delete sc;
        return fail(e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, hipGetErrorString(e));
    }
    const uint8_t* d = (const uint8_t*)sc->dblob;
    SceneDev& v = sc->dev;
    v.nodes = sc->d_nodes_ext ? sc->d_nodes_ext : (const rt_node*)(d + h.off_nodes); v.prims = (const rt_prim*)(d + h.off_prims);
    v.geom = (const rt_geom*)(d + h.off_geom); v.tri_attrs = (const rt_tri_attrs*)(d + h.off_tri_attrs);
    v.xforms = (const rt_xform*)(d + h.off_xforms);
    v.materials = (const rt_material*)(d + h.off_materials); v.textures = (const rt_texture*)(d + h.off_textures);
    v.texels = d + h.off_texels; v.lights = (const rt_light*)(d + h.off_lights);
    v.mat_uses_uv = sc->d_mat_uses_uv;
    v.inst_cull = sc->d_inst_cull; v.inst_planes = sc->d_inst_planes;
    if (tune("RT_TUNE_INST_CULL", 1) == 0) v.inst_cull = nullptr;   // A/B
    v.stream = sc->d_stream; v.root_off = P.root_off_prod;
    v.end_off = P.prod_end_off; v.end_off_full = P.end_off[h.root_end];
    v.stream_full = sc->d_stream_full; v.node_off_full = sc->d_node_off_full; v.root_off_full = P.root_off;
    v.node_off = sc->d_node_off; v.box_abs_max = box_abs_max;
    v.cert_cells = sc->d_cert_cells; v.cert_entries = sc->d_cert_entries; v.cert_normals = sc->d_cert_normals;
    v.cert_planes = sc->d_cert_planes; v.cert_plane_off = sc->d_cert_plane_off; v.cert_plane_leaf = sc->d_cert_plane_leaf; v.cert_n_normals = (uint32_t)(pr.normals.size() / 3); v.prune_time0 = pr.time0; v.prune_time1 = pr.time1; v.prune_delta_coef = pr.delta_coef;
    v.prune_on = pr.on ? 1u : 0u; v.prune_xf = pr.xf;
    v.translate_xf = RT_NONE;
    {   // apply_xform<true>'s shortcut: the pruned space's transform if it qualifies, else the first that does
        const rt_xform* xf = (const rt_xform*)((const uint8_t*)blob + h.off_xforms);
        auto pure_translation = [&](uint32_t i) {
            const double* m = xf[i].t_inv;
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r) if (m[4 * c + r] != (r == c ? 1.0 : 0.0)) return false;
            for (int r = 0; r < 3; ++r) if (!std::isfinite(m[12 + r]) || m[12 + r] == 0.0) return false;
            return true;
        };
        if (pr.on && pr.xf != RT_NONE && pr.xf < h.n_xforms && pure_translation(pr.xf)) v.translate_xf = pr.xf;
        else for (uint32_t i = 0; i < h.n_xforms && v.translate_xf == RT_NONE; ++i) if (pure_translation(i)) v.translate_xf = i;
    }
#ifdef RT_NO_TRANSLATE_SHORTCUT   /* A/B */
    v.translate_xf = RT_NONE;
#endif
    v.prune_tau_coef = pr.tau_coef; v.prune_radius = pr.radius; v.prune_box_abs_max = pr.box_abs_max;
    v.n_nodes = h.n_nodes; v.boxes_finite = boxes_finite ? 1u : 0u;
    v.camera = h.camera; v.n_lights = h.n_lights; v.root_node = h.root_node; v.root_end = h.root_end; v.background = h.background;
    {   // ray-sorting grid: mean +- 2.5 sigma of the scene-level leaf-box centres (robust against a 1000-unit ground sphere
        // or far-away light spheres); origins outside clamp to the border cells
        double sum[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
        uint64_t cnt = 0;
        for (uint32_t i = h.root_node; i < h.root_end; ++i) {
            const rt_node& nd = nodes[i];
            if ((nd.flags & RT_NODE_NO_BOX) || nd.prim0 == RT_NONE) continue;
            bool ok = true;
            for (int k = 0; k < 3; ++k) ok = ok && std::isfinite(nd.bmin[k]) && std::isfinite(nd.bmax[k]) && (nd.bmax[k] - nd.bmin[k]) <= 500.0;
            if (!ok) continue;
            for (int k = 0; k < 3; ++k) { double c = 0.5 * (nd.bmin[k] + nd.bmax[k]); sum[k] += c; sq[k] += c * c; }
            ++cnt;
        }
        sc->sortable = cnt >= 64;
        for (int k = 0; k < 3; ++k) {
            double mean = cnt ? sum[k] / (double)cnt : 0.0;
            double sd = cnt ? std::sqrt(std::max(sq[k] / (double)cnt - mean * mean, 0.0)) : 1.0;
            double ext = std::max(5.0 * sd, 1e-6);
            v.sort_min[k] = (float)(mean - 0.5 * ext);
            v.sort_scale[k] = (float)(32.0 / ext);
        }
        // a handful of scene-level leaves that are INSTANCES (cornell: six transformed cubes and a ground sphere, `main.rs:161-188`): every sub-walk a wave
        // enters is paid by all 64 lanes (ray transform, filter set-up, 23 records), so rays that go the same way are worth a sort here too.  Moments of
        // seven centres mean nothing (the ground sphere alone puts the whole room into one cell): the grid is the union of the leaf boxes that are not
        // outliers -- no axis longer than 8 x the median extent of that axis over the leaves
        if (!sc->sortable && sc->has_instances && !getenv("RT_NO_SMALL_SORT")) {
            // world box of a scene-level leaf's object (the list items of `hittable.rs:41-53` carry no box): an instance's is its sub-BVH's root box
            // taken through the forward transform (the blob holds the inverse only; a sort grid needs no exact bits), a sphere's is centre +- radius
            struct Box { double lo[3], hi[3]; };
            std::vector<Box> boxes;
            const rt_xform* xf = (const rt_xform*)(b + h.off_xforms);
            const rt_geom* geom = (const rt_geom*)(b + h.off_geom);
            auto object_box = [&](uint32_t p, Box& bx) -> bool {
                if (p == RT_NONE || p >= h.n_prims) return false;
                double olo[3], ohi[3];
                if (prims[p].kind == RT_PRIM_INSTANCE && prims[p].index < h.n_nodes) {
                    const rt_node& r = nodes[prims[p].index];
                    for (int k = 0; k < 3; ++k) { olo[k] = r.bmin[k]; ohi[k] = r.bmax[k]; }
                } else if (prims[p].kind == RT_PRIM_SPHERE) {
                    for (int k = 0; k < 3; ++k) { olo[k] = geom[p].d[k] - std::fabs(geom[p].d[3]); ohi[k] = geom[p].d[k] + std::fabs(geom[p].d[3]); }
                } else return false;
                for (int k = 0; k < 3; ++k) if (!std::isfinite(olo[k]) || !std::isfinite(ohi[k])) return false;
                if (prims[p].xform == RT_NONE || prims[p].xform >= h.n_xforms) { for (int k = 0; k < 3; ++k) { bx.lo[k] = olo[k]; bx.hi[k] = ohi[k]; } return true; }
                const double* m = xf[prims[p].xform].t_inv;   // column-major: m[4 c + r]
                double a3[3][3], f[3][3];
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) a3[r][c] = m[4 * c + r];
                const double det = a3[0][0] * (a3[1][1] * a3[2][2] - a3[1][2] * a3[2][1]) - a3[0][1] * (a3[1][0] * a3[2][2] - a3[1][2] * a3[2][0]) +
                                   a3[0][2] * (a3[1][0] * a3[2][1] - a3[1][1] * a3[2][0]);
                if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
                    const int r1 = (c + 1) % 3, r2 = (c + 2) % 3, c1 = (r + 1) % 3, c2 = (r + 2) % 3;   // adjugate: cofactor of (c, r)
                    f[r][c] = (a3[r1][c1] * a3[r2][c2] - a3[r1][c2] * a3[r2][c1]) / det;
                }
                for (int k = 0; k < 3; ++k) { bx.lo[k] = INFINITY; bx.hi[k] = -INFINITY; }
                for (int corner = 0; corner < 8; ++corner) {
                    double q[3];
                    for (int k = 0; k < 3; ++k) q[k] = ((corner >> k) & 1 ? ohi[k] : olo[k]) - m[12 + k];   // world = F (object - t_inv's translation)
                    for (int r = 0; r < 3; ++r) {
                        const double w = f[r][0] * q[0] + f[r][1] * q[1] + f[r][2] * q[2];
                        if (!std::isfinite(w)) return false;
                        bx.lo[r] = std::min(bx.lo[r], w); bx.hi[r] = std::max(bx.hi[r], w);
                    }
                }
                return true;
            };
            for (uint32_t i = h.root_node; i < h.root_end; ++i) {
                const rt_node& nd = nodes[i];
                if (nd.prim0 == RT_NONE) continue;
                Box b;
                if (object_box(nd.prim0, b)) boxes.push_back(b);
                if (nd.prim1 != RT_NONE && nd.prim1 != nd.prim0 && object_box(nd.prim1, b)) boxes.push_back(b);
            }
            double med[3] = {0, 0, 0}, lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int k = 0; k < 3 && !boxes.empty(); ++k) {
                std::vector<double> e;
                for (const Box& b : boxes) e.push_back(b.hi[k] - b.lo[k]);
                std::sort(e.begin(), e.end()); med[k] = e[e.size() / 2];
            }
            uint32_t kept = 0;
            for (const Box& b : boxes) {
                bool in = true;
                for (int k = 0; k < 3; ++k) in = in && (b.hi[k] - b.lo[k]) <= 8.0 * med[k];
                if (!in) continue;
                ++kept;
                for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
            }
            if (kept >= 2) {
                sc->sortable = true;
                for (int k = 0; k < 3; ++k) {
                    double e = std::max(hi[k] - lo[k], 1e-6);
                    v.sort_min[k] = (float)lo[k];
                    v.sort_scale[k] = (float)(32.0 / e);
                }
            }
        }
        if (getenv("RT_PRINT_SORTGRID")) fprintf(stderr, "[sortgrid] sortable %d  min %.6g %.6g %.6g  cells per unit %.6g %.6g %.6g\n", (int)sc->sortable, v.sort_min[0], v.sort_min[1], v.sort_min[2], v.sort_scale[0], v.sort_scale[1], v.sort_scale[2]);
#ifdef RT_TRACE_DIAG
        fprintf(stderr, "[diag-sortgrid] %.9g %.9g %.9g %.9g %.9g %.9g\n", v.sort_min[0], v.sort_min[1], v.sort_min[2], v.sort_scale[0], v.sort_scale[1], v.sort_scale[2]);
#endif
    }
    *out = sc;
    return RT_OK;
}

int rt_render_end(rt_job* job, rt_stats* stats);
int rt_scene_destroy(rt_scene* sc) {
    if (!sc) return RT_OK;
    (void)hipSetDevice(sc->ctx->device);
    for (int k = 0; k < 2; ++k) {   // jobs nobody ended: in the order they were begun
        rt_job* first = sc->lane_job[0] && (!sc->lane_job[1] || sc->lane_job[0]->seq < sc->lane_job[1]->seq) ? sc->lane_job[0] : sc->lane_job[1];
        if (first) (void)rt_render_end(first, nullptr);
    }
    { std::lock_guard<std::mutex> lk(sc->job_mu); sc->workers_quit = true; }
    sc->job_cv.notify_all();
    for (auto& w : sc->worker) if (w.joinable()) w.join();
    for (auto& e : sc->job_done) if (e) (void)hipEventDestroy(e);
    (void)hipStreamSynchronize(sc->ctx->stream);
    for (auto& pl : sc->ev_pool) pl.destroy();
    ws_free(sc->ws); ws_free(sc->ws2);
    if (sc->stream2) (void)hipStreamDestroy(sc->stream2);
    (void)hipFree(sc->dblob);
    (void)hipFree(sc->d_mat_uses_uv);
    (void)hipFree(sc->d_stream);
    (void)hipFree(sc->d_node_off);
    if (sc->d_nodes_ext) (void)hipFree(sc->d_nodes_ext);
    (void)hipFree(sc->d_stream_full);
    (void)hipFree(sc->d_node_off_full);
    if (sc->d_cert_cells) (void)hipFree(sc->d_cert_cells);
    if (sc->d_cert_entries) (void)hipFree(sc->d_cert_entries);
    if (sc->d_cert_normals) (void)hipFree(sc->d_cert_normals);
    if (sc->d_cert_planes) (void)hipFree(sc->d_cert_planes);
    if (sc->d_cert_plane_off) (void)hipFree(sc->d_cert_plane_off);
    if (sc->d_cert_plane_leaf) (void)hipFree(sc->d_cert_plane_leaf);
    if (sc->d_inst_cull) (void)hipFree(sc->d_inst_cull);
    if (sc->d_inst_planes) (void)hipFree(sc->d_inst_planes);
    sc->ctx->live_scenes -= 1;
    delete sc;
    return RT_OK;
}

int rt_trace_closest_device(rt_scene* sc, uint32_t n, const double* rays_dev, double t_min, double t_max, uint32_t* prim_id_dev, double* t_dev) {
    if (!sc || (n && (!rays_dev || !prim_id_dev || !t_dev))) return fail(RT_ERR_BAD_ARG, "rt_trace_closest_device: null argument");
    if (n == 0) return RT_OK;
    RT_NOT_BUSY(sc, "rt_trace_closest_device");
    HIP_OK(hipSetDevice(sc->ctx->device));
    int rc = ws_ensure(sc, n, sc->ws.depth > 0 ? sc->ws.depth : 1);
    if (rc != RT_OK) return rc;
    hipStream_t st = sc->ctx->stream;
    hipLaunchKernelGGL(k_aos_to_soa, grid_for(n), dim3(256), 0, st, rays_dev, n, sc->ws.pb.rays, sc->ws.cap);
    launch_trace(sc, false, nullptr, nullptr, n, sc->ws.pb.rays, t_min, t_max, prim_id_dev, nullptr, t_dev);
    HIP_OK(hipGetLastError());
    return RT_OK;
}

int rt_trace_closest_device_noalloc(rt_scene* sc, uint32_t n, const double* rays_dev, double t_min, double t_max, uint32_t* prim_id_dev, double* t_dev) {
    if (!sc || (n && (!rays_dev || !prim_id_dev || !t_dev))) return fail(RT_ERR_BAD_ARG, "rt_trace_closest_device_noalloc: null argument");
    if (n > RT_NOALLOC_MAX_RAYS) return fail(RT_ERR_BAD_ARG, "rt_trace_closest_device_noalloc: more than " + std::to_string(RT_NOALLOC_MAX_RAYS) + " rays per call (the scratch array is fixed)");
    if (n == 0) return RT_OK;
    RT_NOT_BUSY(sc, "rt_trace_closest_device_noalloc");   // it launches on the context stream, which is lane 0 of the render jobs
    HIP_OK(hipSetDevice(sc->ctx->device));
    hipStream_t st = sc->ctx->stream;
    unsigned char* base = sc->ctx->scratch;  // never the array's name: that is a host-side stand-in, not a device address
    unsigned long long* stats = (unsigned long long*)base;
    double* staged = (double*)(base + RT_NOALLOC_STATS_BYTES);
    HIP_OK(hipMemsetAsync(stats, 0, RT_N_STATS * sizeof(unsigned long long), st));  // the array keeps the last call's contents; the ray blocks are rewritten below
    hipLaunchKernelGGL(k_aos_to_soa, grid_for(n), dim3(256), 0, st, rays_dev, n, staged, RT_NOALLOC_MAX_RAYS);
    launch_trace(sc, stats, RT_NOALLOC_MAX_RAYS, st, false, nullptr, nullptr, n, staged, t_min, t_max, prim_id_dev, nullptr, t_dev);
    HIP_OK(hipGetLastError());
    return RT_OK;
}

int rt_trace_closest(rt_scene* sc, uint32_t n, const double* rays, double t_min, double t_max, uint32_t* prim_id, uint32_t* inst_id, double* t,
                     uint64_t* counters) {
    if (!sc || (n && (!rays || !prim_id || !t))) return fail(RT_ERR_BAD_ARG, "rt_trace_closest: null argument");
    if (n == 0) { if (counters) for (int k = 0; k < 8; ++k) counters[k] = 0; return RT_OK; }
    RT_NOT_BUSY(sc, "rt_trace_closest");
    HIP_OK(hipSetDevice(sc->ctx->device));
    int rc = ws_ensure(sc, n, sc->ws.depth > 0 ? sc->ws.depth : 1);
    if (rc != RT_OK) return rc;
    Workspace& ws = sc->ws;
    hipStream_t st = sc->ctx->stream;
    std::vector<double> soa(RT_RAY_STRIDE * (size_t)n, 0.0);  // internal layout: one 64-byte line per ray
    for (uint32_t i = 0; i < n; ++i)
        for (int c = 0; c < 7; ++c) soa[(size_t)i * RT_RAY_STRIDE + c] = rays[(size_t)i * 7 + c];
    HIP_OK(hipMemcpyAsync(ws.pb.rays, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemsetAsync(ws.stats, 0, RT_N_STATS * sizeof(unsigned long long), st));
    launch_trace(sc, counters != nullptr, nullptr, nullptr, n, ws.pb.rays, t_min, t_max, ws.pb.hit_prim, ws.pb.hit_inst, ws.pb.hit_t);
    HIP_OK(hipGetLastError());
    std::vector<uint32_t> hp(n), hi(n);
    unsigned long long st4[RT_N_STATS];
    HIP_OK(hipMemcpyAsync(hp.data(), ws.pb.hit_prim, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(hi.data(), ws.pb.hit_inst, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(t, ws.pb.hit_t, n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(st4, ws.stats, sizeof st4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < n; ++i) {
        if (hp[i] == RT_NONE) { prim_id[i] = RT_NONE; if (inst_id) inst_id[i] = RT_NONE; t[i] = 0.0; continue; }
        const rt_prim& p = sc->host_prims[hp[i]];
        prim_id[i] = p.object_id;
        if (inst_id) inst_id[i] = hi[i] != RT_NONE ? sc->host_prims[hi[i]].inst_id : p.inst_id;
    }
    if (counters) { counters[0] = st4[1]; counters[1] = st4[2]; counters[2] = st4[3]; counters[3] = st4[13]; counters[4] = st4[14]; counters[5] = counters[6] = counters[7] = 0; }
    return RT_OK;
}

int rt_primary_rays(rt_scene* sc, uint32_t width, uint32_t height, uint32_t sample, uint64_t seed, double* rays) {
    if (!sc || !rays || width < 2 || height < 2) return fail(RT_ERR_BAD_ARG, "rt_primary_rays: bad argument");
    const uint64_t n64 = (uint64_t)width * height;
    if (n64 > 0x7FFFFFFFull) return fail(RT_ERR_BAD_ARG, "rt_primary_rays: frame too large");
    const uint32_t n = (uint32_t)n64;
    HIP_OK(hipSetDevice(sc->ctx->device));
    int rc = ws_ensure(sc, n, sc->ws.depth > 0 ? sc->ws.depth : 1);
    if (rc != RT_OK) return rc;
    Workspace& ws = sc->ws;
    hipStream_t st = sc->ctx->stream;
    hipLaunchKernelGGL(k_raygen, grid_for(n), dim3(256), 0, st, sc->dev.camera, width, height, n, sample, n, (const uint32_t*)nullptr, (const uint32_t*)nullptr, seed,
                       ws.pb.rays, ws.cap, (PathInfo*)nullptr, 0u);
    double* aos = nullptr;
    HIP_OK(hipMalloc((void**)&aos, (size_t)n * 7 * sizeof(double)));
    hipLaunchKernelGGL(k_soa_to_aos, grid_for(n), dim3(256), 0, st, ws.pb.rays, n, ws.cap, aos);
    hipError_t e = hipMemcpyAsync(rays, aos, (size_t)n * 7 * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(aos);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

// samples_per_pass == 0: a pure function of the parameters and the scene's pass budget (see rt_render)
static uint32_t default_pass_size(uint32_t npix, int32_t max_depth, uint64_t budget) {
    const uint64_t by_paths = std::max<uint64_t>(1, (128ull << 20) / npix);
    const uint64_t by_mem = std::max<uint64_t>(1, budget / ((uint64_t)npix * ((uint64_t)max_depth * REC_STRIDE * 8 + 320)));
    return (uint32_t)std::min<uint64_t>(std::min(by_paths, by_mem), 256);
}
// Path buffers of one lane for a call that left the pass size to the library.  When they do not fit (a GPU that something else
// occupies in part), the scene's budget is halved -- 80, 40, 20, 10, 5, 2.5 GiB: a fixed ladder -- and the pass size recomputed,
// so the call renders in smaller passes instead of failing; the budget stays lowered for the scene's later calls (they cut the
// same passes) and rt_stats.samples_per_pass reports what was used.  `halve_call`: rt_render's "at most half the call's samples".
static int ws_ensure_default(rt_scene* sc, Workspace& ws, uint32_t npix, int32_t max_depth, uint32_t spp, bool halve_call, uint32_t& spass, bool compact) {
    for (int step = 0;; ++step) {
        spass = default_pass_size(npix, max_depth, sc->pass_budget);
        if (halve_call) spass = std::min(spass, (spp + 1u) / 2u);
        spass = std::max(1u, std::min(spass, spp));
        spass = (spp + (spp + spass - 1u) / spass - 1u) / ((spp + spass - 1u) / spass);   // the same number of passes, of near-equal size (25 spp at 12 per pass: 9 + 8 + 8, not 12 + 12 + 1)
        if ((uint64_t)npix * spass > 0x7FFFFF00ull) return fail(RT_ERR_BAD_ARG, "samples_per_pass * pixels too large");
        int rc = ws_ensure(sc, ws, npix * spass, max_depth, compact);
        if (rc != RT_ERR_OOM || step >= 5 || default_pass_size(npix, max_depth, sc->pass_budget) == 1) return rc;
        (void)hipGetLastError();
        sc->pass_budget /= 2;
    }
}

static int check_params(const rt_render_params* p) {
    if (!p) return fail(RT_ERR_BAD_ARG, "null params");
    if (p->width < 2 || p->height < 2) return fail(RT_ERR_BAD_ARG, "width/height must be >= 2 (u = (x+xi)/(width-1), main.rs:996)");
    if ((uint64_t)p->width * p->height > (1ull << 28)) return fail(RT_ERR_BAD_ARG, "frame too large");
    if (p->spp_end < p->spp_begin) return fail(RT_ERR_BAD_ARG, "spp_end < spp_begin");
    if (p->max_depth < 0 || p->max_depth > 4096) return fail(RT_ERR_BAD_ARG, "max_depth out of range");
    if (p->integrator < RT_INT_MIS || p->integrator > RT_INT_UNIFORM) return fail(RT_ERR_BAD_ARG, "unknown integrator");
    return RT_OK;
}

int rt_accum_zero(rt_context* ctx, float* accum_dev, uint32_t width, uint32_t height) {
    if (!ctx || !accum_dev) return fail(RT_ERR_BAD_ARG, "rt_accum_zero: null argument");
    if ((uint64_t)width * height > (1ull << 28)) return fail(RT_ERR_BAD_ARG, "rt_accum_zero: frame too large");
    HIP_OK(hipSetDevice(ctx->device));
    HIP_OK(hipMemsetAsync(accum_dev, 0, (size_t)width * height * 3 * sizeof(float), ctx->stream));
    return RT_OK;
}

int rt_scene_set_pass_budget(rt_scene* sc, uint64_t bytes_per_lane) {
    if (!sc) return fail(RT_ERR_BAD_ARG, "rt_scene_set_pass_budget: null scene");
    RT_NOT_BUSY(sc, "rt_scene_set_pass_budget");
    sc->pass_budget = bytes_per_lane ? bytes_per_lane : RT_DEFAULT_PASS_BUDGET;
    return RT_OK;
}

int rt_render(rt_scene* sc, const rt_render_params* p, float* accum_dev, rt_stats* stats) {
    if (!sc || !accum_dev) return fail(RT_ERR_BAD_ARG, "rt_render: null argument");
    int rc = check_params(p);
    if (rc != RT_OK) return rc;
    RT_NOT_BUSY(sc, "rt_render");
    HIP_OK(hipSetDevice(sc->ctx->device));
    const uint32_t npix = p->width * p->height;
    const uint32_t spp = p->spp_end - p->spp_begin;
    if (stats) memset(stats, 0, sizeof *stats);
    if (spp == 0 || p->max_depth == 0) {  // depth <= 0: every sample returns 0 (integrator.rs:33-35)
        if (stats) stats->paths = (uint64_t)npix * spp;
        HIP_OK(hipStreamSynchronize(sc->ctx->stream));
        return RT_OK;
    }
    // Two passes are in flight at a time, each on its own stream with its own path buffers (a "lane"): every traversal launch
    // lasts as long as its slowest wave and the deep bounces carry few rays, so one pass alone leaves the machine part-idle
    // for ~1/8 of its time; the other lane's kernels fill those gaps.  The counting kernel runs on one lane (same passes, same sums).
    // Pass size.  samples_per_pass == 0: a pure function of the parameters and the scene's pass budget (never of free memory as
    // such or of the calls before): pass boundaries decide the order of the fp32 additions into the accumulator, so resumed,
    // chunked and sharded renders of the same samples must cut the same passes to stay bit-identical.  Budget: 80 GiB of path
    // state per lane (two lanes + scene fit the 288 GB; rt_scene_set_pass_budget), at most 128M paths, at most half the call's
    // samples so that both lanes have work.  If the buffers do not fit (a shared GPU), the budget steps down its ladder
    // (ws_ensure_default) and the call renders in smaller passes; rt_stats.samples_per_pass says which.
    uint32_t spass = p->samples_per_pass;
    const bool compact = !(p->integrator == RT_INT_MIS && sc->dev.n_lights > 0);   // as run_paths decides it
    if (spass == 0) {
        rc = ws_ensure_default(sc, sc->ws, npix, p->max_depth, spp, true, spass, compact);
        if (rc != RT_OK) return rc;
    } else {
        spass = std::max(1u, std::min(spass, spp));
        if ((uint64_t)npix * spass > 0x7FFFFF00ull) return fail(RT_ERR_BAD_ARG, "samples_per_pass * pixels too large");
        rc = ws_ensure(sc, sc->ws, npix * spass, p->max_depth, compact);
        if (rc != RT_OK) return rc;
    }
    const bool count_tests = (p->flags & RT_RENDER_COUNT_TESTS) != 0;
    const bool sort_rays = (p->flags & RT_RENDER_NO_SORT) == 0 && sc->sortable;
    const uint32_t n_pass = (spp + spass - 1u) / spass;
    int n_lanes = (n_pass >= 2 && !count_tests && (p->flags & RT_RENDER_ONE_LANE) == 0) ? 2 : 1;
    hipStream_t st = sc->ctx->stream;
    if (n_lanes == 2 && !sc->stream2) HIP_OK(hipStreamCreateWithFlags(&sc->stream2, hipStreamNonBlocking));
    if (n_lanes == 2 && (uint64_t)npix * spass >= sc->ws2_failed_cap) n_lanes = 1;      // did not fit before
    if (n_lanes == 2 && ws_ensure(sc, sc->ws2, npix * spass, p->max_depth, compact) != RT_OK) {  // no room for a second set of path buffers:
        (void)hipGetLastError();                                                         // same passes, one after the other
        sc->ws2_failed_cap = (uint64_t)npix * spass;
        n_lanes = 1;
    }

    struct LaneRun {
        Workspace* ws; hipStream_t st; EventTimer tm; uint64_t launches = 0, tail_bounces = 0; int rc = RT_OK; std::string err;
        LaneRun(Workspace* w, hipStream_t s, bool timed, EventPool* pool) : ws(w), st(s), tm(s, timed, pool) {}
    };
    LaneRun lane[2] = {LaneRun(&sc->ws, st, stats != nullptr, &sc->ev_pool[0]), LaneRun(&sc->ws2, sc->stream2, stats != nullptr, &sc->ev_pool[1])};
    // pass k adds its sums to the accumulator after pass k-1 did (fixed order: the frame does not depend on the lanes)
    EventScope scope;   // destroyed on every return below
    std::vector<hipEvent_t> added(n_pass, nullptr);
    for (auto& e : added) HIP_OK(scope.make(&e, hipEventDisableTiming));
    std::mutex mu; std::condition_variable cv; uint32_t n_recorded = 0; bool broken = false;
    hipEvent_t e0 = nullptr, e1 = nullptr, e_lane2 = nullptr;
    HIP_OK(scope.make(&e0, hipEventDefault)); HIP_OK(scope.make(&e1, hipEventDefault)); HIP_OK(scope.make(&e_lane2, hipEventDisableTiming));
    HIP_OK(hipEventRecord(e0, st));
    if (n_lanes == 2) HIP_OK(hipStreamWaitEvent(sc->stream2, e0, 0));   // work queued on the caller's stream before this call comes first
    auto run_lane = [&](int li) -> int {
        LaneRun& L = lane[li];
        Workspace& ws = *L.ws;
        HIP_OK(hipSetDevice(sc->ctx->device));
        HIP_OK(hipMemsetAsync(ws.stats, 0, RT_N_STATS * sizeof(unsigned long long), L.st));
        for (uint32_t k = (uint32_t)li; k < n_pass; k += (uint32_t)n_lanes) {
            const uint32_t s0 = p->spp_begin + k * spass;
            const uint32_t ns = std::min(spass, p->spp_end - s0);
            int r = run_paths(sc, ws, L.st, npix * ns, false, p->width, p->height, s0, p->max_depth, p->integrator, p->seed, count_tests, sort_rays, L.tm, L.launches, nullptr, (p->flags & RT_RENDER_LANE_WALK) != 0, false,
                              (p->flags & RT_RENDER_MATERIAL_BINS) != 0, (p->flags & (RT_RENDER_COUNT_STEPS | RT_RENDER_COUNT_TESTS)) != 0, &L.tail_bounces, (p->flags & RT_RENDER_SERIAL_SHADOW) != 0);
            if (r != RT_OK) return r;
            if (k > 0) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return n_recorded >= k || broken; });
                if (broken) return fail(RT_ERR_HIP, "the other render lane failed");
                lk.unlock();
                HIP_OK(hipStreamWaitEvent(L.st, added[k - 1], 0));
            }
            L.tm.begin(CAT_OTHER);
            hipLaunchKernelGGL(k_accumulate, grid_for(npix), dim3(256), 0, L.st, ws.pb.radiance, ws.pb.info, ws.cap, npix, ns, accum_dev);
            L.tm.end();
            L.launches += 1;
            HIP_OK(hipEventRecord(added[k], L.st));
            { std::lock_guard<std::mutex> lk(mu); n_recorded = k + 1; }
            cv.notify_all();
        }
        return RT_OK;
    };
    auto guarded = [&](int li) {
        lane[li].rc = run_lane(li);
        if (lane[li].rc != RT_OK) {
            lane[li].err = g_err;
            { std::lock_guard<std::mutex> lk(mu); broken = true; }
            cv.notify_all();
        }
    };
    if (n_lanes == 2) {
        std::thread t2;
        try { t2 = std::thread(guarded, 1); } catch (...) { n_lanes = 1; }  // no second host thread: same passes on one lane (nothing has been launched yet)
        guarded(0);
        if (t2.joinable()) t2.join();
    } else guarded(0);
    unsigned long long st4[RT_N_STATS] = {}, st4b[RT_N_STATS] = {};
    hipError_t he = hipSuccess;
    if (n_lanes == 2) {
        he = hipEventRecord(e_lane2, sc->stream2);
        if (he == hipSuccess) he = hipStreamWaitEvent(st, e_lane2, 0);
        if (he == hipSuccess) he = hipMemcpyAsync(st4b, sc->ws2.stats, sizeof st4b, hipMemcpyDeviceToHost, sc->stream2);
    }
    if (he == hipSuccess) he = hipEventRecord(e1, st);
    if (he == hipSuccess) he = hipMemcpyAsync(st4, sc->ws.stats, sizeof st4, hipMemcpyDeviceToHost, st);
    if (n_lanes == 2) (void)hipStreamSynchronize(sc->stream2);
    hipError_t hs = hipStreamSynchronize(st);
    if (he == hipSuccess) he = hs;
    if (he == hipSuccess) he = hipGetLastError();
    for (int li = 0; li < n_lanes && rc == RT_OK; ++li)
        if (lane[li].rc != RT_OK) { rc = lane[li].rc; g_err = lane[li].err; }
    if (rc == RT_OK && he != hipSuccess) rc = fail(RT_ERR_HIP, std::string("rt_render: ") + hipGetErrorString(he));
    if (stats && rc == RT_OK) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        double by_cat[CAT_N] = {}; uint64_t n_by_cat[CAT_N] = {};
        for (int li = 0; li < n_lanes; ++li) lane[li].tm.collect(by_cat, CAT_N, n_by_cat);
        for (int k = 0; k < RT_N_STATS; ++k) st4[k] += st4b[k];
        stats->rays = st4[0]; stats->box_tests = st4[1]; stats->tri_tests = st4[2]; stats->sphere_tests = st4[3];
        stats->paths = (uint64_t)npix * spp; stats->launches = lane[0].launches + lane[1].launches;
        stats->ms_total = ms; stats->ms_trace = by_cat[CAT_TRACE] + by_cat[CAT_SHADOW]; stats->ms_shade = by_cat[CAT_SHADE]; stats->ms_other = by_cat[CAT_OTHER];
        stats->trace_launches = n_by_cat[CAT_TRACE] + n_by_cat[CAT_SHADOW]; stats->filter_violations = st4[13]; stats->trace_wave_steps = st4[15];
        stats->ms_trace_shadow = by_cat[CAT_SHADOW]; stats->trace_shadow_launches = n_by_cat[CAT_SHADOW];
        stats->shadow_rays = st4[RT_STAT_SHADOW_RAYS]; stats->ms_tail = by_cat[CAT_TAIL]; stats->tail_bounces = lane[0].tail_bounces + lane[1].tail_bounces; stats->samples_per_pass = spass; stats->lanes = (uint32_t)n_lanes;
        stats->walk_run_steps = st4[4]; stats->walk_tri_steps = st4[5]; stats->walk_slow_steps = st4[6]; stats->walk_iterations = st4[7];
    } else {
        double by_cat[CAT_N] = {}; uint64_t n_by_cat[CAT_N] = {};
        for (int li = 0; li < n_lanes; ++li) lane[li].tm.collect(by_cat, CAT_N, n_by_cat);   // releases the events
    }
    return rc;
}

// One job = one rt_render call on ONE lane (own stream, own path buffers, own host thread for the per-bounce readbacks).
static void job_main(rt_job* J) {
    rt_scene* sc = J->sc;
    const rt_render_params& p = J->p;
    auto body = [&]() -> int {
        HIP_OK(hipSetDevice(sc->ctx->device));
        Workspace& ws = J->lane == 0 ? sc->ws : sc->ws2;
        hipStream_t st = J->lane == 0 ? sc->ctx->stream : sc->stream2;
        const uint32_t npix = p.width * p.height, spp = p.spp_end - p.spp_begin;
        EventTimer tm(st, true, &sc->ev_pool[J->lane]);
        uint64_t launches = 0, tail_bounces = 0;
        EventScope scope;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        HIP_OK(scope.make(&e0, hipEventDefault)); HIP_OK(scope.make(&e1, hipEventDefault));
        HIP_OK(hipEventRecord(e0, st));
        HIP_OK(hipMemsetAsync(ws.stats, 0, RT_N_STATS * sizeof(unsigned long long), st));
        const bool sort_rays = (p.flags & RT_RENDER_NO_SORT) == 0 && sc->sortable;
        bool waited = J->seq == 0;
        for (uint32_t s0 = p.spp_begin; s0 < p.spp_end; s0 += J->spass) {
            const uint32_t ns = std::min(J->spass, p.spp_end - s0);
            int r = run_paths(sc, ws, st, npix * ns, false, p.width, p.height, s0, p.max_depth, p.integrator, p.seed, false, sort_rays, tm, launches, nullptr,
                              (p.flags & RT_RENDER_LANE_WALK) != 0, false, (p.flags & RT_RENDER_MATERIAL_BINS) != 0, (p.flags & (RT_RENDER_COUNT_STEPS | RT_RENDER_COUNT_TESTS)) != 0, &tail_bounces, (p.flags & RT_RENDER_SERIAL_SHADOW) != 0);
            if (r != RT_OK) return r;
            if (!waited) {   // the job begun before this one adds its sums first (fixed order of the fp32 additions)
                std::unique_lock<std::mutex> lk(sc->job_mu);
                sc->job_cv.wait(lk, [&] { return sc->jobs_recorded >= J->seq; });
                lk.unlock();
                HIP_OK(hipStreamWaitEvent(st, sc->job_done[(J->seq - 1) % 4], 0));
                waited = true;
            }
            tm.begin(CAT_OTHER);
            hipLaunchKernelGGL(k_accumulate, grid_for(npix), dim3(256), 0, st, ws.pb.radiance, ws.pb.info, ws.cap, npix, ns, J->accum);
            tm.end();
            launches += 1;
        }
        HIP_OK(hipEventRecord(sc->job_done[J->seq % 4], st));
        { std::lock_guard<std::mutex> lk(sc->job_mu); sc->jobs_recorded = J->seq + 1; }
        sc->job_cv.notify_all();
        HIP_OK(hipEventRecord(e1, st));
        unsigned long long st4[RT_N_STATS] = {};
        HIP_OK(hipMemcpyAsync(st4, ws.stats, sizeof st4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipGetLastError());
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        double by_cat[CAT_N] = {}; uint64_t n_by_cat[CAT_N] = {};
        tm.collect(by_cat, CAT_N, n_by_cat);
        rt_stats& S = J->st;
        S.rays = st4[0]; S.paths = (uint64_t)npix * spp; S.launches = launches; S.ms_total = ms;
        S.ms_trace = by_cat[CAT_TRACE] + by_cat[CAT_SHADOW]; S.ms_shade = by_cat[CAT_SHADE]; S.ms_other = by_cat[CAT_OTHER];
        S.trace_launches = n_by_cat[CAT_TRACE] + n_by_cat[CAT_SHADOW]; S.trace_wave_steps = st4[15];
        S.ms_trace_shadow = by_cat[CAT_SHADOW]; S.trace_shadow_launches = n_by_cat[CAT_SHADOW]; S.shadow_rays = st4[RT_STAT_SHADOW_RAYS]; S.ms_tail = by_cat[CAT_TAIL]; S.tail_bounces = tail_bounces;
        S.samples_per_pass = J->spass; S.lanes = 1;
        S.walk_run_steps = st4[4]; S.walk_tri_steps = st4[5]; S.walk_slow_steps = st4[6]; S.walk_iterations = st4[7];
        return RT_OK;
    };
    J->rc = body();
    if (J->rc != RT_OK) {
        J->err = g_err;
        // the jobs behind this one must not wait for ever: count it as recorded (their sums are then meaningless, and their rt_render_end says why)
        { std::lock_guard<std::mutex> lk(sc->job_mu); sc->jobs_recorded = std::max(sc->jobs_recorded, J->seq + 1); }
        sc->job_cv.notify_all();
    }
}

static void lane_worker(rt_scene* sc, int lane) {
    for (;;) {
        rt_job* J = nullptr;
        {
            std::unique_lock<std::mutex> lk(sc->job_mu);
            sc->job_cv.wait(lk, [&] { return sc->workers_quit || (sc->lane_job[lane] && !sc->lane_job[lane]->started); });
            if (sc->workers_quit) return;
            J = sc->lane_job[lane];
            J->started = true;
        }
        job_main(J);
        { std::lock_guard<std::mutex> lk(sc->job_mu); J->finished = true; }
        sc->job_cv.notify_all();
    }
}

int rt_render_begin(rt_scene* sc, const rt_render_params* p, float* accum_dev, rt_job** out) {
    if (!sc || !accum_dev || !out) return fail(RT_ERR_BAD_ARG, "rt_render_begin: null argument");
    int rc = check_params(p);
    if (rc != RT_OK) return rc;
    if (p->flags & RT_RENDER_COUNT_TESTS) return fail(RT_ERR_BAD_ARG, "rt_render_begin: the counting kernel runs through rt_render");
    if (p->spp_end == p->spp_begin || p->max_depth == 0) return fail(RT_ERR_BAD_ARG, "rt_render_begin: empty sample range or depth 0 (nothing to render)");
    const int lane = (int)(sc->job_seq % 2);
    if (sc->lane_job[lane]) return fail(RT_ERR_BAD_ARG, "rt_render_begin: two jobs are in flight already (rt_render_end the older one first)");
    HIP_OK(hipSetDevice(sc->ctx->device));
    const uint32_t npix = p->width * p->height, spp = p->spp_end - p->spp_begin;
    if (lane == 1 && !sc->stream2) HIP_OK(hipStreamCreateWithFlags(&sc->stream2, hipStreamNonBlocking));
    for (auto& e : sc->job_done) if (!e) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    uint32_t spass = p->samples_per_pass;
    const bool compact = !(p->integrator == RT_INT_MIS && sc->dev.n_lights > 0);   // as run_paths decides it
    if (spass == 0) rc = ws_ensure_default(sc, lane == 0 ? sc->ws : sc->ws2, npix, p->max_depth, spp, false, spass, compact);   // one lane per job: not halved
    else {
        spass = std::max(1u, std::min(spass, spp));
        if ((uint64_t)npix * spass > 0x7FFFFF00ull) return fail(RT_ERR_BAD_ARG, "samples_per_pass * pixels too large");
        rc = ws_ensure(sc, lane == 0 ? sc->ws : sc->ws2, npix * spass, p->max_depth, compact);
    }
    if (rc != RT_OK) return rc;
    rt_job* J = new (std::nothrow) rt_job();
    if (!J) return fail(RT_ERR_OOM, "rt_render_begin: host allocation failed");
    J->sc = sc; J->p = *p; J->accum = accum_dev; J->lane = lane; J->spass = spass;
    if (sc->job_seq == 0 || (!sc->lane_job[0] && !sc->lane_job[1])) {   // nothing in flight: this job has no predecessor to wait for
        std::lock_guard<std::mutex> lk(sc->job_mu);
        sc->jobs_recorded = sc->job_seq;
    }
    J->seq = sc->job_seq;
    if (lane == 1) {   // work queued on the caller's stream before this call comes first (lane 0 IS that stream)
        hipEvent_t e = nullptr;
        hipError_t he = hipEventCreateWithFlags(&e, hipEventDisableTiming);
        if (he == hipSuccess) he = hipEventRecord(e, sc->ctx->stream);
        if (he == hipSuccess) he = hipStreamWaitEvent(sc->stream2, e, 0);
        if (e) (void)hipEventDestroy(e);
        if (he != hipSuccess) { delete J; return fail(RT_ERR_HIP, hipGetErrorString(he)); }
    }
    if (!sc->worker[lane].joinable()) {
        try { sc->worker[lane] = std::thread(lane_worker, sc, lane); } catch (...) { delete J; return fail(RT_ERR_OOM, "rt_render_begin: cannot start a host thread"); }
    }
    { std::lock_guard<std::mutex> lk(sc->job_mu); sc->lane_job[lane] = J; }
    sc->job_cv.notify_all();
    sc->job_seq += 1;
    *out = J;
    return RT_OK;
}

int rt_render_end(rt_job* J, rt_stats* stats) {
    if (!J) return fail(RT_ERR_BAD_ARG, "rt_render_end: null job");
    rt_scene* sc = J->sc;
    if (sc->lane_job[J->lane] != J) return fail(RT_ERR_BAD_ARG, "rt_render_end: not a job in flight on its scene");
    {
        std::unique_lock<std::mutex> lk(sc->job_mu);
        sc->job_cv.wait(lk, [&] { return J->finished; });
        sc->lane_job[J->lane] = nullptr;
    }
    int rc = J->rc;
    if (rc != RT_OK) g_err = J->err;
    else if (stats) *stats = J->st;
    delete J;
    return rc;
}

int rt_capture_rays(rt_scene* sc, const rt_render_params* p, uint32_t bounce, int shadow, uint32_t stride, uint32_t max_n, uint32_t* n_out, uint32_t* queue_len,
                    double* rays, uint32_t* prim_id, uint32_t* inst_id, double* t, double* t_light) {
    if (!sc || !n_out || !rays || !prim_id || !inst_id || !t || !t_light || stride == 0) return fail(RT_ERR_BAD_ARG, "rt_capture_rays: bad argument");
    int rc = check_params(p);
    if (rc != RT_OK) return rc;
    if (p->spp_end != p->spp_begin + 1) return fail(RT_ERR_BAD_ARG, "rt_capture_rays: exactly one sample index");
    RT_NOT_BUSY(sc, "rt_capture_rays");
    if ((int32_t)bounce >= p->max_depth) return fail(RT_ERR_BAD_ARG, "rt_capture_rays: bounce >= max_depth");
    HIP_OK(hipSetDevice(sc->ctx->device));
    const uint32_t npix = p->width * p->height;
    rc = ws_ensure(sc, sc->ws, npix, p->max_depth);
    if (rc != RT_OK) return rc;
    Capture cap; cap.bounce = bounce; cap.shadow = shadow != 0; cap.stride = stride; cap.max_n = max_n;
    EventTimer tm(sc->ctx->stream, false, nullptr);
    uint64_t launches = 0;
    HIP_OK(hipMemsetAsync(sc->ws.stats, 0, RT_N_STATS * sizeof(unsigned long long), sc->ctx->stream));
    rc = run_paths(sc, sc->ws, sc->ctx->stream, npix, false, p->width, p->height, p->spp_begin, p->max_depth, p->integrator, p->seed, false,
                   (p->flags & RT_RENDER_NO_SORT) == 0 && sc->sortable, tm, launches, &cap, (p->flags & RT_RENDER_LANE_WALK) != 0);
    HIP_OK(hipStreamSynchronize(sc->ctx->stream));
    if (rc != RT_OK) return rc;
    const uint32_t n = (uint32_t)cap.prim.size();
    *n_out = n; if (queue_len) *queue_len = cap.queue_len;
    memcpy(rays, cap.rays.data(), (size_t)n * 7 * sizeof(double));
    memcpy(t, cap.t.data(), (size_t)n * sizeof(double)); memcpy(t_light, cap.t_light.data(), (size_t)n * sizeof(double));
    for (uint32_t i = 0; i < n; ++i) {   // object ids, as rt_trace_closest reports them
        if (cap.prim[i] == RT_NONE) { prim_id[i] = RT_NONE; inst_id[i] = RT_NONE; continue; }
        const rt_prim& q = sc->host_prims[cap.prim[i]];
        prim_id[i] = q.object_id;
        inst_id[i] = cap.inst[i] != RT_NONE ? sc->host_prims[cap.inst[i]].inst_id : q.inst_id;
    }
    return RT_OK;
}

int rt_render_host(rt_scene* sc, const rt_render_params* p, float* accum_host, rt_stats* stats) {
    if (!sc || !accum_host) return fail(RT_ERR_BAD_ARG, "rt_render_host: null argument");
    int rc = check_params(p);
    if (rc != RT_OK) return rc;
    HIP_OK(hipSetDevice(sc->ctx->device));
    const size_t bytes = (size_t)p->width * p->height * 3 * sizeof(float);
    float* d = nullptr;
    HIP_OK(hipMalloc((void**)&d, bytes));
    hipError_t e = hipMemsetAsync(d, 0, bytes, sc->ctx->stream);
    if (e == hipSuccess) {
        rc = rt_render(sc, p, d, stats);
        if (rc == RT_OK) e = hipMemcpy(accum_host, d, bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d);
    if (rc != RT_OK) return rc;
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

int rt_resolve(rt_context* ctx, const float* accum_dev, uint32_t width, uint32_t height, uint32_t spp, int tonemap, float* out_dev) {
    if (!ctx || !accum_dev || !out_dev || spp == 0) return fail(RT_ERR_BAD_ARG, "rt_resolve: bad argument");
    if ((uint64_t)width * height > (1ull << 28)) return fail(RT_ERR_BAD_ARG, "rt_resolve: frame too large");
    HIP_OK(hipSetDevice(ctx->device));
    const uint32_t n3 = width * height * 3;
    hipLaunchKernelGGL(k_resolve, grid_for(n3), dim3(256), 0, ctx->stream, accum_dev, n3, 0.f, spp, tonemap, out_dev);
    HIP_OK(hipGetLastError());
    return RT_OK;
}

int rt_resolve_host(const float* accum_host, uint32_t width, uint32_t height, uint32_t spp, int tonemap, float* out_host) {
    if (!accum_host || !out_host || spp == 0) return fail(RT_ERR_BAD_ARG, "rt_resolve_host: bad argument");
    if ((uint64_t)width * height > (1ull << 28)) return fail(RT_ERR_BAD_ARG, "rt_resolve_host: frame too large");
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    float *a = nullptr, *o = nullptr;
    HIP_OK(hipMalloc((void**)&a, bytes));
    hipError_t e = hipMalloc((void**)&o, bytes);
    if (e == hipSuccess) e = hipMemcpy(a, accum_host, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const uint32_t n3 = width * height * 3;
        hipLaunchKernelGGL(k_resolve, grid_for(n3), dim3(256), 0, 0, a, n3, 0.f, spp, tonemap, o);
        e = hipMemcpy(out_host, o, bytes, hipMemcpyDeviceToHost);
    }
    (void)hipFree(a); (void)hipFree(o);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

int rt_sample_radiance_sig(rt_scene* sc, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_index, const uint32_t* sample_index,
                           int32_t max_depth, int32_t integrator, uint64_t seed, double* out_rgb, uint64_t* out_sig);
int rt_sample_radiance(rt_scene* sc, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_index, const uint32_t* sample_index,
                       int32_t max_depth, int32_t integrator, uint64_t seed, double* out_rgb) {
    return rt_sample_radiance_sig(sc, width, height, n, pixel_index, sample_index, max_depth, integrator, seed, out_rgb, nullptr);
}
int rt_sample_radiance_sig(rt_scene* sc, uint32_t width, uint32_t height, uint32_t n, const uint32_t* pixel_index, const uint32_t* sample_index,
                           int32_t max_depth, int32_t integrator, uint64_t seed, double* out_rgb, uint64_t* out_sig) {
    if (!sc || (n && (!pixel_index || !sample_index || !out_rgb))) return fail(RT_ERR_BAD_ARG, "rt_sample_radiance: null argument");
    rt_render_params p{}; p.width = width; p.height = height; p.max_depth = max_depth; p.integrator = integrator;
    int rc = check_params(&p);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    for (uint32_t i = 0; i < n; ++i)
        if (pixel_index[i] >= width * height) return fail(RT_ERR_BAD_ARG, "pixel index out of range");
    HIP_OK(hipSetDevice(sc->ctx->device));
    if (max_depth == 0) { memset(out_rgb, 0, (size_t)n * 3 * sizeof(double)); if (out_sig) memset(out_sig, 0, (size_t)n * sizeof(uint64_t)); return RT_OK; }
    RT_NOT_BUSY(sc, "rt_sample_radiance");
    rc = ws_ensure(sc, n, max_depth);
    if (rc != RT_OK) return rc;
    Workspace& ws = sc->ws;
    hipStream_t st = sc->ctx->stream;
    HIP_OK(hipMemcpyAsync(ws.in_pixel, pixel_index, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(ws.in_sample, sample_index, n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    EventTimer tm(st, false, nullptr);
    uint64_t launches = 0;
    if (out_sig && !ws.pb.sig) { rc = ws_alloc(ws, ws.pb.sig, ws.cap); if (rc != RT_OK) { ws.pb.sig = nullptr; return rc; } }
    rc = run_paths(sc, sc->ws, sc->ctx->stream, n, true, width, height, 0, max_depth, integrator, seed, false, sc->sortable, tm, launches, nullptr, false, out_sig != nullptr);
    if (rc != RT_OK) return rc;
    if (out_sig) HIP_OK(hipMemcpyAsync(out_sig, ws.pb.sig, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    std::vector<double> rad(3 * (size_t)ws.cap);
    HIP_OK(hipMemcpyAsync(rad.data(), ws.pb.radiance, rad.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    for (uint32_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) out_rgb[(size_t)i * 3 + c] = rad[(size_t)c * ws.cap + i];
    return RT_OK;
}

int rt_pdf_eval(rt_context* ctx, uint32_t n, uint32_t n_comp, const rt_pdf_desc* pdfs, const double* dirs, uint64_t seed, double* value, double* generated) {
    if (!ctx || (n && (!pdfs || !dirs || !value || !generated))) return fail(RT_ERR_BAD_ARG, "rt_pdf_eval: null argument");
    if (n_comp < 1 || n_comp > RT_PDF_MAX_COMP) return fail(RT_ERR_BAD_ARG, "rt_pdf_eval: n_comp must be 1..8");
    for (size_t k = 0; k < (size_t)n * n_comp; ++k)
        if (pdfs[k].kind > RT_PDF_CONE) return fail(RT_ERR_BAD_ARG, "rt_pdf_eval: unknown PDF kind");
    if (n == 0) return RT_OK;
    HIP_OK(hipSetDevice(ctx->device));
    const size_t b_desc = (size_t)n * n_comp * sizeof(rt_pdf_desc), b_dir = (size_t)n * 3 * sizeof(double), b_val = (size_t)n * sizeof(double);
    uint8_t* d = nullptr;
    HIP_OK(hipMalloc((void**)&d, b_desc + 2 * b_dir + b_val));
    rt_pdf_desc* d_desc = (rt_pdf_desc*)d; double* d_dir = (double*)(d + b_desc); double* d_gen = d_dir + (size_t)n * 3; double* d_val = d_gen + (size_t)n * 3;
    hipError_t e = hipMemcpy(d_desc, pdfs, b_desc, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_dir, dirs, b_dir, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_pdf_eval, grid_for(n), dim3(256), 0, ctx->stream, n, n_comp, d_desc, d_dir, seed, d_val, d_gen);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(value, d_val, b_val, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(generated, d_gen, b_dir, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

int rt_ieee_div_sqrt(rt_context* ctx, uint32_t n, const double* a, const double* b, double* q, double* s) {
    if (!ctx || (n && (!a || !b || !q || !s))) return fail(RT_ERR_BAD_ARG, "rt_ieee_div_sqrt: null argument");
    if (n == 0) return RT_OK;
    HIP_OK(hipSetDevice(ctx->device));
    double* d = nullptr;
    const size_t bytes = (size_t)n * sizeof(double);
    HIP_OK(hipMalloc((void**)&d, 4 * bytes));
    hipError_t e = hipMemcpy(d, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_div_sqrt, grid_for(n), dim3(256), 0, ctx->stream, n, d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(q, d + 2 * (size_t)n, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(s, d + 3 * (size_t)n, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(RT_ERR_HIP, hipGetErrorString(e));
    return RT_OK;
}

}  // extern "C"
