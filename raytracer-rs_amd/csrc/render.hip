// render.hip -- kernels and C ABI of the MI355X back end (gfx950 only).
// Entry points are declared in include/rt_mi355x.h, which cites the reference interface each replaces.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_device.h"
#include "rt_wave.h"
#include "rt_front.h"
#include "rt_pool.h"

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------
// error plumbing
static thread_local std::string g_last_error;
int rtm_internal_fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t e__ = (expr);                                                                                   \
        if (e__ != hipSuccess)                                                                                     \
            return rtm_internal_fail(RTM_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));          \
    } while (0)

static int g_device = -1;
int rtm_internal_device() { return g_device; }
static uint32_t* g_work_counter = nullptr;  // device word: next unclaimed tile of the frame in flight (rt_wave.h)
static int g_num_cus = 256;
static uint32_t g_t_test_frac = 44;  // env RTM_TEST_FRAC
static uint32_t g_front_wide = 0;  // test slots per step (0 = by scene); env RTM_FRONT_WIDE
static uint32_t g_t_test_frac_long = 20;  // ... in scenes with long cells (every cell a ray meets is opened: test sooner); env RTM_TEST_FRAC_LONG
static int g_kernel = 2;              // 0: rt_wave.h, 2: rt_front.h (default); env RTM_KERNEL=wave|front
#ifndef RTM_NOALLOC_SCRATCH_MIB
#define RTM_NOALLOC_SCRATCH_MIB 256  // size of the static work space of rtm_render_device_noalloc
#endif
constexpr int kFrontStreamsMax = 4;
static uint32_t g_front_streams = 2;  // chunks of a frame in flight at once (0 = by frame size: 3 from 12 Mi sample paths, else 2); env RTM_FRONT_STREAMS / rtm_set_tuning("front_streams")
static hipStream_t g_front_stream[kFrontStreamsMax] = {};
static hipEvent_t g_front_fork = nullptr, g_front_join[kFrontStreamsMax] = {};
static bool g_front_trace_wgs_env = false;
static uint32_t g_front_trace_wgs_per_cu = RTM_FRONT_OCC;  // trace workgroups per CU (RTM_FRONT_TRACE_WGS)
static bool g_front_debug_rounds = false;
// rtm_set_tuning("front_timed_events", 1): ordinary frames carry a HIP event pair around every trace launch (on the
// stream it is launched on) until rtm_timed_events_collect() reads them: bench.py's roofline line times the dominant
// kernel in the same regime as the frames it reports.
static bool g_front_timed_events = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_front_ev_pool;
static size_t g_front_ev_used = 0;
static uint32_t g_front_pool = 0;  // trace kernel: 1 = rays parked in LDS pools (rt_pool.h) when the scene's bitmaps leave room for them, 0 = one ray per lane (rt_front.h); env RTM_FRONT_POOL
static uint32_t g_pool_refill = 40, g_pool_walk_more = 40, g_pool_test_frac = 32;  // env RTM_POOL_REFILL / RTM_POOL_WALK_MORE / RTM_POOL_TEST_FRAC
static int g_pool_rays = 96;  // rays per wave of the pool kernel (72, with more waves per CU, was measured no better and is not built)
static uint32_t g_pool_plain = 0, g_pool_waves = 0;  // env RTM_POOL_PLAIN=1: stage the plain occupancy only; RTM_POOL_WAVES: cap on waves per workgroup
static uint32_t g_front_inlane_lights = 1;  // env RTM_FRONT_INLANE_LIGHTS
static bool g_front_inlane_force = false;  // the knob was set by hand: also for scenes with many analytic primitives
static uint32_t g_front_inlane = 24;  // shade kernel: paths advance in the lane while no query of theirs needs the trace kernel, if at least this many lanes of the wave do (0 = never); env RTM_FRONT_INLANE
static uint32_t g_front_sph_screen = 1;  // shade kernel, scenes with more than four spheres: packed-f32 screen in front of the f64 sphere tests; env RTM_FRONT_SPH_SCREEN
static uint32_t g_front_rehit = 1;  // shade kernel: scenes with at most four spheres + planes recompute a path query's analytic hit instead of carrying it; env RTM_FRONT_REHIT
static uint32_t g_front_coop = 1;  // trace kernel: idle lanes test the entries of the last one or two testing rays of a wave
static uint32_t g_front_claim = 128, g_front_steal = 8;  // trace kernel: items per claim; other waves' sub-queues tried
static uint32_t g_front_refill = 24, g_front_cull = 1, g_front_shade_waves_per_cu = 24;  // shade waves per CU: a multiple of 4
static bool g_front_shade_waves_env = false;
struct FrontWorkspace {  // rt_front.h path state and queues; grow-only, one per process (the library renders on one device, rtm_init)
    char* base = nullptr;
    size_t bytes = 0;
};
static FrontWorkspace g_front_ws;
// One frame at a time uses the process-wide state above (work space, internal streams, fork/join events, the wave kernel's
// tile counter).  Frames queued from different caller streams are ordered by g_frame_done (recorded on the caller's stream
// behind a frame's last kernel, waited for by the next frame's first); a second host thread entering a render entry point
// while another is inside gets RTM_ERR_STATE.
static std::atomic<bool> g_render_busy{false};
static hipEvent_t g_frame_done = nullptr;
static bool g_frame_done_valid = false;
static size_t g_front_max_paths = 0;  // sample paths in flight at once over the chunks of a frame (~0.5 KB of work space each); 0 = by frame size; env RTM_FRONT_MAX_PATHS
static uint32_t g_s_threshold = 24, g_t_min_lanes = 32;  // wave scheduler knobs (rt_wave.h); env RTM_S_THRESHOLD / RTM_T_MIN_LANES

// ------------------------------------------------------------------------------------------------
// kernels
using namespace rtd;

RTM_DEV uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, off, 64);
        const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

// hit() of caller-supplied rays (differential tests)
__global__ __launch_bounds__(256) void trace_kernel(RtmDevScene S, uint64_t n, const double* __restrict__ rays, uint32_t* obj,
                                                     uint32_t* tri, double* t_out, double* nrm_out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o = ld3(rays + i * 6), d = ld3(rays + i * 6 + 3);
    Counters cnt;
    Hit h;
    const bool got = scene_hit<false>(S, o, d, h, cnt);
    V3 nn = mk(0, 0, 0);
    if (got) nn = hit_normal(S, h, o, d, ray_at(o, d, h.t));
    obj[i] = got ? h.obj : kNoId;
    tri[i] = got ? h.tri : kNoId;
    t_out[i] = got ? h.t : __builtin_inf();
    nrm_out[i * 3] = nn.x; nrm_out[i * 3 + 1] = nn.y; nrm_out[i * 3 + 2] = nn.z;
}

// main.rs:33-48,176-185: optional gamma, then (c*255.0) as u8 (truncating, saturating, NaN -> 0)
__global__ void tonemap_kernel(const double* __restrict__ rgb, uint64_t n, int gamma, uint8_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = rgb[i];
    if (gamma) v = v > 0.0 ? pow(v, 1.0 / 2.2) : 0.0;
    v = v * 255.0;
    out[i] = !(v > 0.0) ? 0 : (v >= 255.0 ? 255 : (uint8_t)v);
}

__global__ void f64_ops_kernel(uint64_t n, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = a[i], y = b[i];
    out[i] = x / y;
    out[n + i] = __builtin_sqrt(fabs(x));
    out[2 * n + i] = fmin(x, y);
    out[3 * n + i] = fmax(x, y);
    out[4 * n + i] = x * y + x;
    out[5 * n + i] = (double)cell_coord(x, 32);
}

// block g of d_blocks = rows y = g + k*G, densely packed -> row-major image
__global__ void deinterleave_kernel(const double* __restrict__ blocks, uint32_t W, uint32_t H, uint32_t G, double* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over H*W*3 doubles of the output
    const uint64_t row_len = (uint64_t)W * 3;
    if (i >= (uint64_t)H * row_len) return;
    const uint32_t y = (uint32_t)(i / row_len);
    const uint64_t within = i % row_len;
    const uint32_t g = y % G, k = y / G;
    // rows owned by ranks < g: ceil((H - r)/G) each
    uint64_t base_rows = 0;
    for (uint32_t r = 0; r < g; ++r) base_rows += (H - r + G - 1) / G;
    out[i] = blocks[(base_rows + k) * row_len + within];
}

// ------------------------------------------------------------------------------------------------
// device scene
struct rtm_scene_dev {
    RtmDevScene view;
    std::vector<void*> allocs;
    std::vector<uint32_t> occ_words;  // per model: words of its cell-occupancy bitmap
    uint32_t max_n = 0;               // largest cells_per_side
    uint64_t max_list = 0;            // largest n_tri_idx
    uint64_t n_long_cells = 0;        // cells with more than 32 entries, over all models
    uint64_t n_nonempty_cells = 0;    // cells with any entry, over all models
    uint32_t occ7_words = 0;          // words of the seven scene-wide bitmaps (rt_front.h)
    bool front_ok = true;             // the scene fits the packed indices of the ray-front path
    int device;
};

namespace {

template <typename T>
int upload(rtm_scene_dev* sd, const std::vector<T>& host, const T** out) {
    void* p = nullptr;
    const size_t bytes = (host.size() ? host.size() : 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    sd->allocs.push_back(p);
    if (!host.empty()) HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return RTM_OK;
}

int check_scene(const rtm_scene* s) {
    if (!s) return rtm_internal_fail(RTM_ERR_INVALID, "scene is null");
    if (s->n_spheres && !(s->sph_cx && s->sph_cy && s->sph_cz && s->sph_r && s->sph_mat && s->sph_obj))
        return rtm_internal_fail(RTM_ERR_INVALID, "sphere arrays missing");
    if (s->n_planes && !(s->pl_nx && s->pl_ny && s->pl_nz && s->pl_d && s->pl_mat && s->pl_obj))
        return rtm_internal_fail(RTM_ERR_INVALID, "plane arrays missing");
    if (s->n_lights && !(s->li_px && s->li_py && s->li_pz && s->li_r && s->li_g && s->li_b && s->li_radius))
        return rtm_internal_fail(RTM_ERR_INVALID, "light arrays missing");
    if (s->n_materials && !(s->mat_r && s->mat_g && s->mat_b && s->mat_type))
        return rtm_internal_fail(RTM_ERR_INVALID, "material arrays missing");
    if (s->n_models && !s->models) return rtm_internal_fail(RTM_ERR_INVALID, "model array missing");
    for (uint32_t i = 0; i < s->n_spheres; ++i)
        if (s->sph_mat[i] >= s->n_materials || s->sph_obj[i] >= s->n_objects)
            return rtm_internal_fail(RTM_ERR_INVALID, "sphere material/object index out of range");
    for (uint32_t i = 0; i < s->n_planes; ++i)
        if (s->pl_mat[i] >= s->n_materials || s->pl_obj[i] >= s->n_objects)
            return rtm_internal_fail(RTM_ERR_INVALID, "plane material/object index out of range");
    for (uint32_t m = 0; m < s->n_models; ++m) {
        const rtm_model& M = s->models[m];
        if (M.cells_per_side == 0 || M.cells_per_side > 1024) return rtm_internal_fail(RTM_ERR_INVALID, "cells_per_side out of range");
        if (!M.offsets || (M.n_tri_idx && !M.tri_idx)) return rtm_internal_fail(RTM_ERR_INVALID, "grid arrays missing");
        if (M.mat_idx >= s->n_materials || M.obj_idx >= s->n_objects)
            return rtm_internal_fail(RTM_ERR_INVALID, "model material/object index out of range");
        for (int a = 0; a < 9; ++a)
            if (M.n_tris && (!M.pos[a] || !M.nrm[a])) return rtm_internal_fail(RTM_ERR_INVALID, "triangle arrays missing");
        const uint64_t n3 = (uint64_t)M.cells_per_side * M.cells_per_side * M.cells_per_side;
        uint32_t prev = 0;
        for (uint64_t c = 0; c < n3; ++c) {  // the kernels index tri_idx by these without further checks
            if (M.offsets[c] < prev) return rtm_internal_fail(RTM_ERR_INVALID, "grid offsets are not non-decreasing");
            prev = M.offsets[c];
        }
        if (prev != M.n_tri_idx) return rtm_internal_fail(RTM_ERR_INVALID, "grid offsets do not end at n_tri_idx");
        for (uint64_t k = 0; k < M.n_tri_idx; ++k)
            if (M.tri_idx[k] >= M.n_tris) return rtm_internal_fail(RTM_ERR_INVALID, "triangle index out of range");
    }
    return RTM_OK;
}

int check_params(const rtm_params* p) {
    if (!p) return rtm_internal_fail(RTM_ERR_INVALID, "params is null");
    if (p->width == 0 || p->height == 0 || p->width > 65536 || p->height > 65536)
        return rtm_internal_fail(RTM_ERR_INVALID, "width/height out of range");
    if (p->spp == 0) return rtm_internal_fail(RTM_ERR_INVALID, "spp must be >= 1");
    if (p->max_bounces == 0 || p->max_bounces > 64) return rtm_internal_fail(RTM_ERR_INVALID, "max_bounces must be 1..64");
    if (p->row_stride == 0 || p->row_begin >= p->height) return rtm_internal_fail(RTM_ERR_INVALID, "bad row_begin/row_stride");
    if ((uint64_t)p->width * p->height > 0xFFFFFFFFull) return rtm_internal_fail(RTM_ERR_INVALID, "image too large");
    if (p->flags & ~(RTM_FLAG_ZERO_JITTER | RTM_FLAG_KERNEL_WAVE)) return rtm_internal_fail(RTM_ERR_INVALID, "unknown flag (2 was RTM_FLAG_KERNEL_V1, the per-lane kernel retired in round 3)");
    return RTM_OK;
}

int need_init() {
    if (g_device < 0) return rtm_internal_fail(RTM_ERR_STATE, "rtm_init has not been called");
    return RTM_OK;
}

}  // namespace

extern "C" {

// The packed indices of the ray-front path (rt_front.h): a list position times ten (doubles per record) is formed in 32 bits,
// and a scene cell number shares a word with three bits of arrival code.  Scenes beyond either run the single-kernel formulation.
int rtm_front_scene_fits(uint64_t list_entries, uint64_t cells) {
    return list_entries <= (uint64_t)rtf::kMaxListEntries && cells < (1ull << 27) ? 1 : 0;
}

uint32_t rtm_local_rows(const rtm_params* p) {
    if (!p || p->row_stride == 0 || p->row_begin >= p->height) return 0;
    return (p->height - p->row_begin + p->row_stride - 1) / p->row_stride;
}

static int scratch_bind(int device);  // (defined beside the scratch array, further down)

// Everything the library holds on the current device (work space, streams, events, counters).  Used by rtm_shutdown
// and by rtm_init when the process moves to another device: none of it may follow to the new one.
static void release_device_state() {
    if (g_device >= 0) (void)hipSetDevice(g_device);
    if (g_frame_done_valid) (void)hipEventSynchronize(g_frame_done);
    if (g_work_counter) (void)hipFree(g_work_counter);
    g_work_counter = nullptr;
    if (g_front_fork) {
        (void)hipEventDestroy(g_front_fork);
        g_front_fork = nullptr;
        for (int i = 0; i < kFrontStreamsMax; ++i) {
            (void)hipStreamDestroy(g_front_stream[i]);
            (void)hipEventDestroy(g_front_join[i]);
            g_front_stream[i] = nullptr;
            g_front_join[i] = nullptr;
        }
    }
    if (g_frame_done) (void)hipEventDestroy(g_frame_done);
    g_frame_done = nullptr;
    g_frame_done_valid = false;
    for (auto& pr : g_front_ev_pool) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    g_front_ev_pool.clear();
    g_front_ev_used = 0;
    if (g_front_ws.base) (void)hipFree(g_front_ws.base);
    g_front_ws.base = nullptr;
    g_front_ws.bytes = 0;
}

int rtm_init(int device) {
    int count = 0;
    HIP_TRY(hipGetDeviceCount(&count));
    if (count <= 0) return rtm_internal_fail(RTM_ERR_DEVICE, "no HIP device visible");
    if (device < 0) {
        const char* lr = std::getenv("LOCAL_RANK");
        device = lr ? std::atoi(lr) % count : 0;
    }
    if (device >= count) return rtm_internal_fail(RTM_ERR_INVALID, "device index out of range");
    if (g_device >= 0 && g_device != device) release_device_state();  // (scenes uploaded to the old device stay there: free them first)
    HIP_TRY(hipSetDevice(device));
    g_device = device;
    if (!g_work_counter) HIP_TRY(hipMalloc((void**)&g_work_counter, 256));
    if (!g_frame_done) HIP_TRY(hipEventCreateWithFlags(&g_frame_done, hipEventDisableTiming));
    if (int rc = scratch_bind(device)) return rc;
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (const char* e = std::getenv("RTM_KERNEL")) g_kernel = std::string(e) == "wave" ? 0 : 2;
    if (const char* e = std::getenv("RTM_FRONT_REFILL")) g_front_refill = (uint32_t)std::max(1, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("RTM_FRONT_CLAIM")) g_front_claim = (uint32_t)std::max(16, std::atoi(e));
    if (const char* e = std::getenv("RTM_FRONT_COOP")) g_front_coop = std::atoi(e) ? 1u : 0u;
    if (const char* e = std::getenv("RTM_FRONT_POOL")) g_front_pool = std::atoi(e) ? 1u : 0u;
    if (const char* e = std::getenv("RTM_POOL_REFILL")) g_pool_refill = (uint32_t)std::max(1, std::min(128, std::atoi(e)));
    if (const char* e = std::getenv("RTM_POOL_PLAIN")) g_pool_plain = std::atoi(e) ? 1u : 0u;
    if (const char* e = std::getenv("RTM_POOL_WAVES")) g_pool_waves = (uint32_t)std::max(0, std::min(16, std::atoi(e)));
    if (const char* e = std::getenv("RTM_POOL_WALK_MORE")) g_pool_walk_more = (uint32_t)std::max(1, std::min(65, std::atoi(e)));
    if (const char* e = std::getenv("RTM_POOL_TEST_FRAC")) g_pool_test_frac = (uint32_t)std::max(0, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("RTM_FRONT_INLANE_LIGHTS")) g_front_inlane_lights = std::atoi(e) ? 1u : 0u;
    if (const char* e = std::getenv("RTM_FRONT_INLANE")) { g_front_inlane = (uint32_t)std::max(0, std::min(64, std::atoi(e))); g_front_inlane_force = true; }
    if (const char* e = std::getenv("RTM_FRONT_SPH_SCREEN")) g_front_sph_screen = std::atoi(e) ? 1u : 0u;
    if (const char* e = std::getenv("RTM_FRONT_REHIT")) g_front_rehit = std::atoi(e) ? 1u : 0u;
    if (const char* e = std::getenv("RTM_FRONT_STEAL")) g_front_steal = (uint32_t)std::max(0, std::atoi(e));
    g_front_shade_waves_env = std::getenv("RTM_FRONT_SHADE_WAVES") != nullptr;
    if (const char* e = std::getenv("RTM_FRONT_SHADE_WAVES")) g_front_shade_waves_per_cu = (uint32_t)std::max(4, std::min(64, std::atoi(e) & ~3));
    g_front_debug_rounds = std::getenv("RTM_FRONT_DEBUG_ROUNDS") != nullptr;
    if (const char* e = std::getenv("RTM_FRONT_STREAMS")) g_front_streams = (uint32_t)std::max(0, std::min(kFrontStreamsMax, std::atoi(e)));
    if (!g_front_fork) {
        HIP_TRY(hipEventCreateWithFlags(&g_front_fork, hipEventDisableTiming));
        for (int i = 0; i < kFrontStreamsMax; ++i) {
            HIP_TRY(hipStreamCreateWithFlags(&g_front_stream[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&g_front_join[i], hipEventDisableTiming));
        }
    }
    g_front_trace_wgs_env = std::getenv("RTM_FRONT_TRACE_WGS") != nullptr;
    if (const char* e = std::getenv("RTM_FRONT_TRACE_WGS")) g_front_trace_wgs_per_cu = (uint32_t)std::max(1, std::min(RTM_FRONT_OCC, std::atoi(e)));
    if (const char* e = std::getenv("RTM_FRONT_CULL")) g_front_cull = std::atoi(e) ? 1u : 0u;
    if (const char* e = std::getenv("RTM_FRONT_MAX_PATHS")) g_front_max_paths = (size_t)std::max(64ll, std::atoll(e));
    if (const char* e = std::getenv("RTM_S_THRESHOLD")) g_s_threshold = (uint32_t)std::max(1, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("RTM_TEST_FRAC")) g_t_test_frac = g_t_test_frac_long = (uint32_t)std::max(0, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("RTM_FRONT_WIDE")) g_front_wide = (uint32_t)std::max(0, std::min(RTM_FTRI_WIDE, std::atoi(e)));
    if (const char* e = std::getenv("RTM_TEST_FRAC_LONG")) g_t_test_frac_long = (uint32_t)std::max(0, std::min(64, std::atoi(e)));
    if (const char* e = std::getenv("RTM_T_MIN_LANES")) g_t_min_lanes = (uint32_t)std::max(0, std::min(64, std::atoi(e)));
    return RTM_OK;
}

int rtm_set_tuning(const char* key, long long value) {
    if (!key) return rtm_internal_fail(RTM_ERR_INVALID, "key is null");
    const std::string k(key);
    if (k == "front_max_paths" && (value == 0 || (value >= 64 && value <= (1ll << 31)))) { g_front_max_paths = (size_t)value; return RTM_OK; }  // 0 = by frame size
    if (k == "front_cull" && (value == 0 || value == 1)) { g_front_cull = (uint32_t)value; return RTM_OK; }
    if (k == "front_refill" && value >= 1 && value <= 64) { g_front_refill = (uint32_t)value; return RTM_OK; }
    if (k == "front_claim" && value >= 16 && value <= (1 << 20)) { g_front_claim = (uint32_t)value; return RTM_OK; }
    if (k == "front_coop" && (value == 0 || value == 1)) { g_front_coop = (uint32_t)value; return RTM_OK; }
    if (k == "front_sph_screen" && (value == 0 || value == 1)) { g_front_sph_screen = (uint32_t)value; return RTM_OK; }
    if (k == "front_rehit" && (value == 0 || value == 1)) { g_front_rehit = (uint32_t)value; return RTM_OK; }
    if (k == "front_inlane_lights" && (value == 0 || value == 1)) { g_front_inlane_lights = (uint32_t)value; return RTM_OK; }
    if (k == "front_inlane" && value == -1) { g_front_inlane = 24; g_front_inlane_force = false; return RTM_OK; }  // back to the default
    if (k == "front_inlane" && value >= 0 && value <= 64) { g_front_inlane = (uint32_t)value; g_front_inlane_force = true; return RTM_OK; }
    if (k == "front_pool" && (value == 0 || value == 1)) { g_front_pool = (uint32_t)value; return RTM_OK; }
    if (k == "pool_refill" && value >= 1 && value <= 128) { g_pool_refill = (uint32_t)value; return RTM_OK; }
    if (k == "pool_walk_more" && value >= 1 && value <= 65) { g_pool_walk_more = (uint32_t)value; return RTM_OK; }
    if (k == "pool_test_frac" && value >= 0 && value <= 64) { g_pool_test_frac = (uint32_t)value; return RTM_OK; }
    if (k == "front_timed_events" && (value == 0 || value == 1)) { g_front_timed_events = value != 0; return RTM_OK; }
    if (k == "front_steal" && value >= 0 && value <= 4096) { g_front_steal = (uint32_t)value; return RTM_OK; }
    if (k == "front_streams" && value >= 0 && value <= kFrontStreamsMax) { g_front_streams = (uint32_t)value; return RTM_OK; }
    return rtm_internal_fail(RTM_ERR_INVALID, "unknown tuning key or value out of range: " + k);
}

int rtm_shutdown(void) {
    release_device_state();
    g_device = -1;
    return RTM_OK;
}

const char* rtm_last_error(void) { return g_last_error.c_str(); }
const char* rtm_version(void) { return "raytracer-rs_amd 0.1 (gfx950, f64, contract=off)"; }

int rtm_scene_upload(const rtm_scene* s, rtm_scene_dev** out) {
    if (!out) return rtm_internal_fail(RTM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (int rc = need_init()) return rc;
    if (int rc = check_scene(s)) return rc;
    auto* sd = new rtm_scene_dev();
    sd->device = g_device;
    auto fail = [&](int rc) {
        rtm_scene_free(sd);
        return rc;
    };
    RtmDevScene& V = sd->view;
    std::memset(&V, 0, sizeof V);
    std::memcpy(V.cam_pos, s->camera.position, 24);
    std::memcpy(V.cam_fwd, s->camera.forward, 24);
    std::memcpy(V.cam_up, s->camera.up, 24);
    std::memcpy(V.cam_right, s->camera.right, 24);
    V.cam_zoom = s->camera.zoom;
    V.n_objects = s->n_objects; V.n_spheres = s->n_spheres; V.n_planes = s->n_planes;
    V.n_models = s->n_models; V.n_lights = s->n_lights; V.n_materials = s->n_materials;

    std::vector<double> sph((size_t)s->n_spheres * 4), pl((size_t)s->n_planes * 4), li((size_t)s->n_lights * 8, 0.0),
        mat((size_t)s->n_materials * 4);
    std::vector<uint32_t> sph_ids((size_t)s->n_spheres * 2), pl_ids((size_t)s->n_planes * 2);
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        sph[i * 4] = s->sph_cx[i]; sph[i * 4 + 1] = s->sph_cy[i]; sph[i * 4 + 2] = s->sph_cz[i]; sph[i * 4 + 3] = s->sph_r[i];
        sph_ids[i * 2] = s->sph_mat[i]; sph_ids[i * 2 + 1] = s->sph_obj[i];
    }
    for (uint32_t i = 0; i < s->n_planes; ++i) {
        pl[i * 4] = s->pl_nx[i]; pl[i * 4 + 1] = s->pl_ny[i]; pl[i * 4 + 2] = s->pl_nz[i]; pl[i * 4 + 3] = s->pl_d[i];
        pl_ids[i * 2] = s->pl_mat[i]; pl_ids[i * 2 + 1] = s->pl_obj[i];
    }
    for (uint32_t i = 0; i < s->n_lights; ++i) {
        const double v[7] = {s->li_px[i], s->li_py[i], s->li_pz[i], s->li_r[i], s->li_g[i], s->li_b[i], s->li_radius[i]};
        std::memcpy(&li[i * 8], v, sizeof v);
    }
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        mat[i * 4] = s->mat_r[i]; mat[i * 4 + 1] = s->mat_g[i]; mat[i * 4 + 2] = s->mat_b[i];
        mat[i * 4 + 3] = s->mat_type[i] ? 1.0 : 0.0;
    }
    // the spheres in f32 for the shade kernel's screen (rt_front.h analytic(), MANY)
    std::vector<float> sph_scr((((size_t)s->n_spheres + 31) / 32) * 32 * 4, 0.0f);
    {
        double lc = 0.0, r2max = 0.0;
        bool ok = true;
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            const double c3[3] = {s->sph_cx[i], s->sph_cy[i], s->sph_cz[i]}, r2 = s->sph_r[i] * s->sph_r[i];
            float* const rec = &sph_scr[(size_t)(i / 2) * 8 + (i & 1)];
            for (int c = 0; c < 3; ++c) { rec[c * 2] = (float)c3[c]; lc = std::max(lc, std::fabs(c3[c])); ok = ok && std::fabs(c3[c]) < 1e8; }
            rec[6] = (float)r2;
            r2max = std::max(r2max, r2);
            ok = ok && r2 < 1e16;  // (NaN fails both)
        }
        V.sph_lc = std::nextafter((float)lc, INFINITY); V.sph_r2max = std::nextafter((float)r2max, INFINITY);
        V.sph_screen_ok = ok ? 1u : 0u;
        if (!ok) { V.sph_lc = 0.0f; V.sph_r2max = 0.0f; std::fill(sph_scr.begin(), sph_scr.end(), 0.0f); }
    }
    int rc;
    if ((rc = upload(sd, sph_scr, &V.sph_screen))) return fail(rc);
    if ((rc = upload(sd, sph, &V.spheres)) || (rc = upload(sd, sph_ids, &V.sph_ids)) || (rc = upload(sd, pl, &V.planes)) ||
        (rc = upload(sd, pl_ids, &V.pl_ids)) || (rc = upload(sd, li, &V.lights)) || (rc = upload(sd, mat, &V.materials)))
        return fail(rc);

    std::vector<RtmDevModel> models(s->n_models);
    // The ray-front path's view of the grids (rt_front.h): ONE array of cell records, list records, boxes, frames and
    // bitmaps over the cells of all models (model m's cells start at cell_base, a multiple of 32; list positions are global),
    // so that the lanes of a wave traverse whichever model their ray is in side by side -- what differs per model is one
    // word of per-lane state (cells per side, cell_base), not a set of wave-uniform pointers.
    std::vector<double> g_rec, g_frames;
    std::vector<uint32_t> g_info, g_box8;
    std::vector<std::vector<uint32_t>> occ7_of_model;
    uint64_t total_cells = 0;
    for (uint32_t m = 0; m < s->n_models; ++m) {
        const rtm_model& M = s->models[m];
        RtmDevModel& D = models[m];
        std::memset(&D, 0, sizeof D);
        const double n = (double)M.cells_per_side;
        for (int c = 0; c < 3; ++c) {
            // Pure functions of the Box3 the reference re-evaluates per call; x86-64 SSE2 and gfx950 f64
            // +,-,/,* are both IEEE, so evaluating them once here yields the same bits.
            D.bmin[c] = M.bbox_center[c] - M.bbox_half[c];
            D.bmax[c] = M.bbox_center[c] + M.bbox_half[c];
            D.cell[c] = (D.bmax[c] - D.bmin[c]) / n;
            D.hc[c] = M.bbox_half[c] / n;
            D.hc2[c] = D.hc[c] * 2.0;
        }
        sd->max_n = std::max(sd->max_n, M.cells_per_side);
        sd->max_list = std::max<uint64_t>(sd->max_list, M.n_tri_idx);
        D.n = M.cells_per_side; D.n_tris = M.n_tris; D.mat = M.mat_idx; D.obj = M.obj_idx;
        const size_t n3 = (size_t)M.cells_per_side * M.cells_per_side * M.cells_per_side;
        std::vector<uint32_t> start(n3 + 2, 0), occ((n3 + 31) / 32, 0);
        for (size_t c = 0; c < n3; ++c) {
            start[c + 1] = M.offsets[c];
            if (start[c + 1] > start[c]) occ[c >> 5] |= 1u << (c & 31);
        }
        start[n3 + 1] = start[n3];
        sd->occ_words.push_back((uint32_t)occ.size());
        std::vector<uint32_t> idx(M.tri_idx, M.tri_idx + M.n_tri_idx);
        std::vector<double> pos((size_t)M.n_tris * 9), nrm((size_t)M.n_tris * 9);
        for (uint32_t t = 0; t < M.n_tris; ++t)
            for (int a = 0; a < 9; ++a) {
                // v0, then the edges v1-v0 and v2-v0: the same IEEE subtraction the reference performs per test
                pos[(size_t)t * 9 + a] = a < 3 ? M.pos[a][t] : M.pos[a][t] - M.pos[a % 3][t];
                nrm[(size_t)t * 9 + a] = M.nrm[a][t];
            }
        // the lists again, as records in list order (rt_front.h): a cell's entries are contiguous 80-byte records, so the
        // traversal reads them without the indirection through tri_idx and neighbouring entries share cache lines
        const uint64_t entry_base = g_rec.size() / 10, cell_base = total_cells;
        if (!rtm_front_scene_fits(entry_base + M.n_tri_idx, cell_base + n3)) {
            sd->front_ok = false;  // (beyond the packed indices of rt_front.h: the single-kernel formulation renders such scenes)
        }
        D.cell_base = (uint32_t)cell_base;
        total_cells += (n3 + 31) & ~(size_t)31;
        g_rec.resize((size_t)(entry_base + M.n_tri_idx) * 10);
        double* const rec = g_rec.data() + (size_t)entry_base * 10;
        for (uint64_t e = 0; e < M.n_tri_idx; ++e) {
            const uint32_t t = idx[e];
            for (int a = 0; a < 9; ++a) rec[e * 10 + a] = pos[(size_t)t * 9 + a];
            const uint64_t bits = t;
            std::memcpy(&rec[e * 10 + 9], &bits, 8);
        }
        // Per cell: [start, end) and, for lists of at most 32 entries, one mask per arrival direction.  The reference tests
        // every entry of every cell it visits; an entry whose triangle is also listed in the cell the DDA step came from
        // was tested one step earlier with the same ray and was not accepted (the walk would have ended there,
        // core.rs:147-149), so testing it again cannot change anything: its bit is cleared.
        {
            const uint32_t n = M.cells_per_side;
            std::vector<uint32_t> info(n3 * 8);
            std::vector<uint32_t> sorted;
            for (size_t c = 0; c < n3; ++c) {
                const uint32_t b = start[c], e = start[c + 1], len = e - b;
                info[c * 8] = b;
                info[c * 8 + 1] = e;
                const uint32_t all = len >= 32 ? 0xFFFFFFFFu : ((1u << len) - 1u);
                const uint32_t cx = (uint32_t)(c % n), cy = (uint32_t)((c / n) % n), cz = (uint32_t)(c / ((size_t)n * n));
                for (int a = 0; a < 6; ++a) {
                    uint32_t mask = all;
                    // arrival code a = axis * 2 + (step > 0): the previous cell is one step back along that axis
                    const int axis = a >> 1, back = (a & 1) ? -1 : 1;
                    const long px = (long)cx + (axis == 0 ? back : 0), py = (long)cy + (axis == 1 ? back : 0), pz = (long)cz + (axis == 2 ? back : 0);
                    if (len > 0 && len <= 32 && px >= 0 && py >= 0 && pz >= 0 && px < (long)n && py < (long)n && pz < (long)n) {
                        const size_t pc = ((size_t)pz * n + (size_t)py) * n + (size_t)px;
                        const uint32_t pb = start[pc], pe = start[pc + 1];
                        if (pe > pb) {
                            sorted.assign(idx.begin() + pb, idx.begin() + pe);
                            std::sort(sorted.begin(), sorted.end());
                            for (uint32_t i = 0; i < len; ++i)
                                if (std::binary_search(sorted.begin(), sorted.end(), idx[b + i])) mask &= ~(1u << i);
                        }
                    }
                    info[c * 8 + 2 + a] = mask;
                }
            }
            // Long cells (more than 32 entries: no arrival masks): every entry gets its triangle's bounding box, quantised to 8
            // bits per coordinate in a frame of the cell -- the region that bounds all its triangles, mapped to [2, 253] --
            // and rounded OUTWARDS by at least one quantum.  The trace kernel tests a ray against these 8-byte boxes before it
            // reads an entry's 80-byte record: a ray whose line misses the enlarged box cannot pass the reference's
            // triangle test (model.rs:43-74 accepts only points of the triangle, give or take rounding that is many orders
            // below a quantum), so skipping the entry leaves every accepted hit, hence the result, unchanged.
            {
                std::vector<double>& frames = g_frames;
                const size_t frames_before = frames.size();
                std::vector<uint32_t>& box8 = g_box8;  // the boxes of the long cells only, each cell's from a 16-byte boundary
                for (size_t c = 0; c < n3; ++c) {
                    const uint32_t b = start[c], e = start[c + 1];
                    if (e - b <= 32u) continue;
                    double rlo[3] = {1e300, 1e300, 1e300}, rhi[3] = {-1e300, -1e300, -1e300};
                    auto tri_box = [&](uint32_t t, double lo[3], double hi[3]) {
                        for (int a = 0; a < 3; ++a) {
                            const double v0 = M.pos[a][t], v1 = M.pos[3 + a][t], v2 = M.pos[6 + a][t];
                            lo[a] = std::min(v0, std::min(v1, v2));
                            hi[a] = std::max(v0, std::max(v1, v2));
                        }
                    };
                    for (uint32_t k = b; k < e; ++k) {
                        double lo[3], hi[3];
                        tri_box(idx[k], lo, hi);
                        for (int a = 0; a < 3; ++a) { rlo[a] = std::min(rlo[a], lo[a]); rhi[a] = std::max(rhi[a], hi[a]); }
                    }
                    double ctr[3], iq[3];
                    bool usable = true;
                    for (int a = 0; a < 3; ++a) {
                        const double ext = (rhi[a] - rlo[a]) * 1.001 + 1e-9 * (1.0 + std::fabs(rlo[a]) + std::fabs(rhi[a]));
                        ctr[a] = 0.5 * (rlo[a] + rhi[a]);
                        iq[a] = 251.0 / ext;
                        usable = usable && std::isfinite(ctr[a]) && std::isfinite(iq[a]) && iq[a] > 0.0;
                    }
                    const uint32_t frame = (uint32_t)(frames.size() / 8);
                    for (int a = 0; a < 3; ++a) frames.push_back(ctr[a]);
                    for (int a = 0; a < 3; ++a) frames.push_back(iq[a]);
                    {   // slot 6: (first box of this cell) - (global list position of its first entry), modulo 2^32, as raw bits
                        if (box8.size() % 4u) box8.resize(box8.size() + 2, 0u);
                        const uint64_t bits = (uint32_t)(box8.size() / 2 - (entry_base + b));
                        double as_double;
                        std::memcpy(&as_double, &bits, 8);
                        frames.push_back(as_double);
                        frames.push_back(0.0);
                    }
                    info[c * 8 + 2] = frame;
                    for (uint32_t k = b; k < e; ++k) {
                        double lo[3], hi[3];
                        tri_box(idx[k], lo, hi);
                        uint32_t q0[3], q1[3];
                        for (int a = 0; a < 3; ++a) {
                            const double l = (lo[a] - ctr[a]) * iq[a] + 127.5, h = (hi[a] - ctr[a]) * iq[a] + 127.5;
                            // (non-finite vertices, or a frame that cannot be formed: the whole range, i.e. never rejected)
                            q0[a] = (usable && l >= 1.0 && l <= 255.0) ? (uint32_t)std::floor(l) - 1u : 0u;
                            q1[a] = (usable && h >= 0.0 && h <= 254.0) ? (uint32_t)std::ceil(h) + 1u : 255u;
                        }
                        box8.push_back(q0[0] | (q0[1] << 8) | (q0[2] << 16) | (q1[0] << 24));
                        box8.push_back(q1[1] | (q1[2] << 8));
                    }
                }
                sd->n_long_cells += (frames.size() - frames_before) / 8;
            }
            // ... and the same knowledge as bitmaps for the walk: arriving along direction a, a cell whose entries were all
            // tested one step earlier is as good as empty
            const size_t words = occ.size();
            std::vector<uint32_t> occ7(words * 7, 0u);
            for (size_t c = 0; c < n3; ++c) {
                const uint32_t len = info[c * 8 + 1] - info[c * 8];
                if (!len) continue;
                sd->n_nonempty_cells++;
                for (int a = 0; a < 6; ++a)
                    if (len > 32 || info[c * 8 + 2 + a] != 0u) occ7[(size_t)a * words + (c >> 5)] |= 1u << (c & 31);
                occ7[6 * words + (c >> 5)] |= 1u << (c & 31);
            }
            occ7_of_model.push_back(std::move(occ7));
            // into the scene-wide array: list positions become global
            for (size_t c = 0; c < n3; ++c) { info[c * 8] += (uint32_t)entry_base; info[c * 8 + 1] += (uint32_t)entry_base; }
            g_info.resize((size_t)total_cells * 8, 0u);
            std::memcpy(g_info.data() + (size_t)cell_base * 8, info.data(), info.size() * 4);
        }
        if ((rc = upload(sd, start, &D.cell_start)) || (rc = upload(sd, idx, &D.tri_idx)) || (rc = upload(sd, pos, &D.tri_pos)) ||
            (rc = upload(sd, nrm, &D.tri_nrm)) || (rc = upload(sd, occ, &D.occupancy)))
            return fail(rc);
    }
    {   // the seven bitmaps over the cells of all models, interleaved: the words of cells [32 w, 32 w + 32) are words 8 w .. 8 w + 6
        // (one shift and one OR from cell and arrival direction to the word, no multiply in the walk)
        const size_t W = (size_t)(total_cells / 32);
        std::vector<uint32_t> occ7_all(W * 8 + 1, 0u);
        for (uint32_t m = 0; m < s->n_models; ++m) {
            const size_t wm = occ7_of_model[m].size() / 7, wbase = models[m].cell_base / 32;
            for (int a = 0; a < 7; ++a)
                for (size_t w = 0; w < wm; ++w) occ7_all[(wbase + w) * 8 + (size_t)a] = occ7_of_model[m][(size_t)a * wm + w];
        }
        V.g_occ_dir_words = (uint32_t)W;
        sd->occ7_words = (uint32_t)std::min<size_t>(W * 8, 0xFFFFFFFFu);
        g_box8.resize(((g_box8.size() + 3) & ~(size_t)3) + 2 * 64 + 8, 0u);  // (the screen reads a whole 64-entry pass from any cell's first box)
        if ((rc = upload(sd, occ7_all, &V.g_occ7)) || (rc = upload(sd, g_info, &V.g_cell_info)) || (rc = upload(sd, g_rec, &V.g_cell_rec)) ||
            (rc = upload(sd, g_frames, &V.g_cell_frame)) || (rc = upload(sd, g_box8, &V.g_cell_box8)))
            return fail(rc);
    }
    if ((rc = upload(sd, models, &V.models))) return fail(rc);
    *out = sd;
    return RTM_OK;
}

int rtm_scene_free(rtm_scene_dev* sd) {
    if (!sd) return RTM_OK;
    for (void* p : sd->allocs) (void)hipFree(p);
    delete sd;
    return RTM_OK;
}

// ---- rt_front.h orchestration ---------------------------------------------------------------------
namespace {

bool front_supported(const rtm_scene_dev* sd, const rtm_params* p) {
    // packed cell coordinates and model indices (rt_front.h); one 8-row band of paths must fit the workspace budget
    const uint64_t band_paths = (uint64_t)((p->width + 7) / 8) * 64u * p->spp;
    return sd->front_ok && sd->occ_words.size() <= 0xFFFFu && sd->max_n <= rtf::kMaxCellsPerSide && band_paths <= (64ull << 20) &&
           p->max_bounces <= rtf::kMaxFrontBounces;  // (the bounce count shares a word with the path's flags)
}

// words of the seven scene-wide bitmaps when they fit the trace kernel's LDS budget (all of them or none: a probe that
// may go either way would wait for both memories)
uint32_t front_occ_words(const rtm_scene_dev* sd) {
    if (sd->occ7_words <= (uint32_t)rtw::kOccLdsWordsMax) return sd->occ7_words;
    if (sd->occ7_words / 8u <= (uint32_t)rtw::kOccLdsWordsMax) return sd->occ7_words / 8u;  // the plain occupancy alone
    return 0u;
}

// One frame as bands of rows; per band: round 0 (camera rays), then shade/trace rounds until no path is alive.
struct FrontProfile {  // rtm_profile_render_device: event pairs per launch (kind 0 shade, 1 trace, 2 resolve), item counts
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[3];
    unsigned long long* d_items = nullptr;
    bool keep_cull = false;  // counting run under the production culling setting
};

__global__ void front_sum_items_kernel(const uint32_t* ctr, uint32_t nq, unsigned long long* out) {
    unsigned long long v = 0;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nq; j += gridDim.x * blockDim.x) {
        const uint32_t* const c = ctr + (size_t)j * rtf::CT_STRIDE + rtf::CT_ITEMS0;
        v += (unsigned long long)c[0] + c[1] + c[2];
    }
    v = wave_sum_u64(v);
    if ((threadIdx.x & 63u) == 0u && v) atomicAdd(out, v);
}

struct FrontFixedWs {  // rtm_render_device_noalloc: a work space that is already there (the static scratch array); nothing is
    char* base;        // allocated, the chunks of the frame run one after the other on the caller's stream, no events
    size_t bytes;
};

struct FrontLayout {  // byte offsets of the work space of one chunk
    size_t o_D0, o_D1, o_U0, o_U1, o_fin, o_res, o_itD, o_itU, o_ctr, bytes;
};
FrontLayout front_layout(size_t np_max, uint32_t nq, uint32_t pf_n, uint32_t pu_n) {
    FrontLayout L{};
    const uint32_t blocks_max = (uint32_t)((np_max + 63) / 64);
    const uint32_t cap_l_max = ((blocks_max + nq - 1) / nq) * 64u;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
    const size_t items_max = (size_t)nq * cap_l_max * 3u;
    const size_t slots_max = (size_t)nq * cap_l_max;  // path slots (>= np_max: every wave's share is rounded up to whole blocks)
    L.o_D0 = take(slots_max * pf_n * 8); L.o_D1 = take(slots_max * pf_n * 8);
    L.o_U0 = take(slots_max * pu_n * 4); L.o_U1 = take(slots_max * pu_n * 4);
    L.o_fin = take(slots_max * 3 * 8); L.o_res = take(slots_max * 3 * sizeof(rtf::FrontResult));
    L.o_itD = take(items_max * rtf::IT_DCOUNT * 8); L.o_itU = take(items_max * rtf::IU_COUNT * 4);
    L.o_ctr = take((size_t)nq * rtf::CT_STRIDE * 4);
    L.bytes = (off + 4095) & ~(size_t)4095;
    return L;
}

int render_front_queue(const rtm_scene_dev* sd, const rtm_params* p, uint32_t n_rows, uint32_t occ_words, double* d_rgb, uint32_t* d_obj,
                       uint32_t* d_tri, unsigned long long* d_cnt, hipStream_t stream, FrontProfile* prof, const FrontFixedWs* fixed);

// An error in the middle of a frame (a failed launch, an event that cannot be recorded) leaves kernels queued on the internal
// streams with nothing ordering the caller's stream, or the next frame's work space, after them: wait for them before the error
// is reported, so that the caller may free or reuse its outputs.
int render_front(const rtm_scene_dev* sd, const rtm_params* p, uint32_t n_rows, uint32_t occ_words, double* d_rgb, uint32_t* d_obj,
                 uint32_t* d_tri, unsigned long long* d_cnt, hipStream_t stream, FrontProfile* prof = nullptr,
                 const FrontFixedWs* fixed = nullptr) {
    const int rc = render_front_queue(sd, p, n_rows, occ_words, d_rgb, d_obj, d_tri, d_cnt, stream, prof, fixed);
    if (rc != RTM_OK && !fixed) {
        const std::string msg = g_last_error;
        for (int i = 0; i < kFrontStreamsMax; ++i)
            if (g_front_stream[i]) (void)hipStreamSynchronize(g_front_stream[i]);
        (void)hipStreamSynchronize(stream);
        g_last_error = msg;
    }
    return rc;
}

int render_front_queue(const rtm_scene_dev* sd, const rtm_params* p, uint32_t n_rows, uint32_t occ_words, double* d_rgb, uint32_t* d_obj,
                       uint32_t* d_tri, unsigned long long* d_cnt, hipStream_t stream, FrontProfile* prof, const FrontFixedWs* fixed) {
    const bool stats = d_cnt != nullptr;
    int mark_rc = RTM_OK;
    auto mark = [&](int kind, bool end, hipStream_t on) {  // event before / after a launch (profile runs only)
        auto ok = [&](hipError_t e, const char* what) {
            if (e != hipSuccess && mark_rc == RTM_OK) mark_rc = rtm_internal_fail(RTM_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
            return e == hipSuccess;
        };
        if (fixed) return;
        if (!prof && g_front_timed_events && kind == 1) {
            if (!end) {
                if (g_front_ev_used == g_front_ev_pool.size()) {
                    hipEvent_t a = nullptr, b = nullptr;
                    if (!ok(hipEventCreate(&a), "hipEventCreate")) return;
                    if (!ok(hipEventCreate(&b), "hipEventCreate")) { (void)hipEventDestroy(a); return; }
                    g_front_ev_pool.push_back({a, b});
                }
                ok(hipEventRecord(g_front_ev_pool[g_front_ev_used].first, on), "hipEventRecord");
            } else if (g_front_ev_used < g_front_ev_pool.size()) {
                ok(hipEventRecord(g_front_ev_pool[g_front_ev_used++].second, on), "hipEventRecord");
            }
            return;
        }
        if (!prof || prof->keep_cull) return;
        if (!end) {
            hipEvent_t a = nullptr, b = nullptr;
            if (!ok(hipEventCreate(&a), "hipEventCreate")) return;
            if (!ok(hipEventCreate(&b), "hipEventCreate")) { (void)hipEventDestroy(a); return; }
            prof->ev[kind].push_back({a, b});
            ok(hipEventRecord(a, on), "hipEventRecord");
        } else if (!prof->ev[kind].empty()) {
            ok(hipEventRecord(prof->ev[kind].back().second, on), "hipEventRecord");
        }
    };
    // what travels with a path (rt_front.h PF_*, PU_*): the fields behind PF_EFF / PU_HOME exist only for scenes with more than two lights
    const uint32_t pf_n = sd->view.n_lights > 2u ? (uint32_t)rtf::PF_COUNT : (uint32_t)rtf::PF_NRM;
    const uint32_t pu_n = sd->view.n_lights > 2u ? (uint32_t)rtf::PU_COUNT : (uint32_t)rtf::PU_LI;
    const uint32_t tiles_x = (p->width + 7) / 8;
    const size_t band_paths = (size_t)tiles_x * 64u * p->spp;  // paths of one 8-row band
    if (band_paths > 0x3FFFFFFFu / 3u) return rtm_internal_fail(RTM_ERR_INVALID, "width x spp too large for one band of paths");
    const uint32_t bands_total = (n_rows + 7) / 8;
    // The frame is cut into chunks of whole 8-row bands; chunk c runs on internal stream c % n_conc with its own work space,
    // so the late rounds of one chunk (few paths alive: launch- and latency-bound) overlap the heavy rounds of another.
    // (0: three chunks in flight win 1-2 % on the whole bunny frame and lose 4 % on suzanne and on small rank shares: not the default)
    const uint32_t want_conc = g_front_streams ? g_front_streams : (band_paths * bands_total >= ((size_t)12 << 20) ? 3u : 2u);
    const uint32_t n_conc = fixed ? 1u : std::max<uint32_t>(1, std::min<uint32_t>({want_conc, (uint32_t)kFrontStreamsMax, bands_total}));
    uint32_t bands_per_chunk;
    if (g_front_max_paths) {
        bands_per_chunk = (uint32_t)std::max<size_t>(1, g_front_max_paths / n_conc / band_paths);
        bands_per_chunk = std::min(bands_per_chunk, (bands_total + n_conc - 1) / n_conc);
    } else {
        // By frame size: one chunk per internal stream where that is at most 64 Mi sample paths (fewer, larger launches win: the
        // bunny frame takes 31.9 ms as two chunks, 34.6 ms as four, profiles/r03_chunks.log), else chunks of 64 Mi -- config 5's
        // 9.4e8 paths run as 15 chunks over ~100 GB of the 288 GB -- and never chunks below 4 Mi paths.
        const size_t total = band_paths * bands_total;
        const size_t per_chunk = std::min<size_t>((size_t)64 << 20, std::max<size_t>((size_t)4 << 20, (total + n_conc - 1) / n_conc));
        uint32_t n_chunks_want = (uint32_t)std::max<size_t>(n_conc, (total + per_chunk - 1) / per_chunk);
        n_chunks_want = (n_chunks_want + n_conc - 1) / n_conc * n_conc;
        bands_per_chunk = std::max<uint32_t>(1, (bands_total + n_chunks_want - 1) / n_chunks_want);
    }
    {   // (pixel of the tile, tile column, tile row of the chunk, sample) of a path travel packed in one word (rt_front.h PU_HOME):
        // width x spp is bounded by front_supported(), the rows of a chunk are bounded here
        auto bits = [](uint32_t n) { uint32_t b = 1; while ((1u << b) < n) ++b; return b; };
        const uint32_t used = 6u + bits(tiles_x) + bits(p->spp);
        if (used > 31u) return rtm_internal_fail(RTM_ERR_INVALID, "width x spp too large for one band of paths");
        bands_per_chunk = std::min(bands_per_chunk, 1u << std::min(20u, 32u - used - 1u));
    }
    // sub-queues (rt_front.h): the shade grid has exactly nq waves, one sub-list per shade wave.  (The fixed work space
    // uses a third of them: every sub-queue owns whole 64-path blocks, which is what a small work space is spent on.)
    // (small chunks -- a rank's share of a frame split over 8 GPUs: under 4 Mi paths -- run a little faster with 16 shade waves per CU
    // than with 24: 6.0 against 6.15 ms for an eighth of the bunny frame)
    const bool small_chunk = band_paths * bands_per_chunk < ((size_t)4 << 20);
    const uint32_t nq = (uint32_t)g_num_cus * (fixed ? 8u : (g_front_shade_waves_env || !small_chunk) ? g_front_shade_waves_per_cu : 16u);
    if (fixed) {  // as many bands at a time as the work space holds
        while (bands_per_chunk > 1 && front_layout(band_paths * bands_per_chunk, nq, pf_n, pu_n).bytes > fixed->bytes) --bands_per_chunk;
        if (front_layout(band_paths * bands_per_chunk, nq, pf_n, pu_n).bytes > fixed->bytes)
            return rtm_internal_fail(RTM_ERR_INVALID, "frame too wide for the static scratch: one 8-row band of width x spp = " +
                                                          std::to_string((unsigned long long)p->width * p->spp) + " sample paths per row needs " +
                                                          std::to_string(front_layout(band_paths, nq, pf_n, pu_n).bytes >> 20) + " MiB of " +
                                                          std::to_string(fixed->bytes >> 20) + " MiB (lower width or spp, or use rtm_render_device)");
    }
    const size_t np_max = band_paths * bands_per_chunk;
    if (np_max + (size_t)64 * 64 * 1024 >= (1ull << 30)) return rtm_internal_fail(RTM_ERR_INVALID, "band of paths too large");
    const FrontLayout L = front_layout(np_max, nq, pf_n, pu_n);
    const size_t o_D0 = L.o_D0, o_D1 = L.o_D1, o_U0 = L.o_U0, o_U1 = L.o_U1, o_fin = L.o_fin, o_res = L.o_res, o_itD = L.o_itD,
                 o_itU = L.o_itU, o_ctr = L.o_ctr, ws_one = L.bytes;
    char* ws_base = fixed ? fixed->base : nullptr;
    if (!fixed) {
        if (ws_one * n_conc > g_front_ws.bytes) {
            // (frames queued earlier, on whatever stream, may still be using the old work space)
            if (g_frame_done_valid) HIP_TRY(hipEventSynchronize(g_frame_done));
            if (g_front_ws.base) HIP_TRY(hipFree(g_front_ws.base));
            g_front_ws.base = nullptr;
            g_front_ws.bytes = 0;
            HIP_TRY(hipMalloc((void**)&g_front_ws.base, ws_one * n_conc));
            g_front_ws.bytes = ws_one * n_conc;
        }
        ws_base = g_front_ws.base;
    }
    // fork: the internal streams start after whatever the caller queued on `stream`
    hipStream_t lanes[kFrontStreamsMax];
    for (uint32_t i = 0; i < n_conc; ++i) lanes[i] = n_conc == 1 ? stream : g_front_stream[i];
    if (n_conc > 1) {
        HIP_TRY(hipEventRecord(g_front_fork, stream));
        for (uint32_t i = 0; i < n_conc; ++i) HIP_TRY(hipStreamWaitEvent(lanes[i], g_front_fork, 0));
    }
    const uint32_t B_rounds = std::max<uint32_t>(1, (sd->view.n_lights + 1) / 2);
    // paths advance in the lane (rt_front.h) when at least `inlane` lanes of their wave can; 65 = never
    const uint32_t inlane_on = (g_front_inlane && (g_front_inlane_force || sd->view.n_spheres + sd->view.n_planes <= 16u)) ? g_front_inlane : 65u;
    // A scene without models has nothing to trace: no trace launches; and a small frame of such a scene (config 1: 1200^2 at 1 spp,
    // 0.63 -> 0.28 ms) is bound by its 22 launches, so every path runs to its end inside round 0's shade kernel (threshold 1: a
    // lane goes on whatever the others do).  Larger frames keep the rounds: their compaction is worth more (20 spp: 3.7 against 4.0 ms).
    const bool no_models = sd->view.n_models == 0u;
    const bool one_launch = no_models && inlane_on <= 64u && !g_front_inlane_force && band_paths * bands_total <= ((size_t)4 << 20);
    const uint32_t inlane = one_launch ? 1u : inlane_on;
    const uint32_t rounds = one_launch ? 1u : 2 + std::max<uint32_t>(p->max_bounces, 1) * B_rounds;
    const dim3 block(256);
    const uint32_t shade_wgs = nq / 4u;
    const bool dense = sd->n_long_cells * 2 >= sd->n_nonempty_cells && sd->n_long_cells > 0;  // most non-empty cells are long (config 5's soup)
    // (three trace workgroups per CU, not the four that fit: alone the kernel is 9 % slower with three waves per SIMD, but the
    // fourth slot lets the other chunk's shade waves run beside it -- the frame is 1 % faster with two chunks in flight; measured)
    const uint32_t trace_wgs = (uint32_t)g_num_cus * (g_front_trace_wgs_env ? g_front_trace_wgs_per_cu : (n_conc > 1 && !dense ? 3u : (uint32_t)RTM_FRONT_OCC));  // (the soup's trace launches are long and bound by issue: four)
    const bool all_lds = occ_words != 0u || sd->occ7_words == 0u;  // the bitmaps the walk probes are staged in LDS
    bool occ_plain_only = all_lds && occ_words != sd->occ7_words;
    // The pool trace kernel (rt_pool.h): one 512-thread workgroup per CU holds the bitmaps once and a pool of kPoolRays rays per
    // wave in LDS.  It needs the probed bitmaps in LDS; when all seven do not fit beside the pools, the plain occupancy alone is staged.
    const bool pool_multi = sd->view.n_models > 1, pool_long = sd->n_long_cells != 0;
    constexpr size_t kLdsDynMax = 160u * 1024u - rtf::kModelTab * 4u;
    bool use_pool = g_front_pool != 0u && all_lds && sd->view.n_models > 0;
    uint32_t pool_waves = 0;
    auto waves_for = [&](uint32_t words) {
        const size_t bm = ((size_t)words * 4 + 15) & ~(size_t)15;
        if (bm >= kLdsDynMax) return 0u;
        const uint32_t w = (uint32_t)std::min<size_t>(16, (kLdsDynMax - bm) / rtf::pool_wave_bytes(g_pool_rays, pool_long, pool_multi));
        return g_pool_waves ? std::min(w, g_pool_waves) : w;
    };
    if (use_pool) {
        pool_waves = g_pool_plain ? 0u : waves_for(occ_words);
        if (pool_waves < 8u && sd->occ7_words) {  // the plain occupancy alone leaves room for more waves
            const uint32_t w2 = waves_for(sd->occ7_words / 8u);
            if (w2 > pool_waves) { pool_waves = w2; occ_words = sd->occ7_words / 8u; occ_plain_only = true; }
        }
        if (pool_waves < 4u) use_pool = false;
    }
    const size_t pool_lds = use_pool ? rtf::pool_lds_bytes(occ_words, g_pool_rays, pool_long, pool_multi, pool_waves) : 0;
    // The chunks run n_conc at a time, each on its own stream and work-space slot, and their launches are ENQUEUED round by
    // round across the group: queued chunk after chunk, the second stream's first kernel reached the GPU only after the
    // host had queued all ~24 launches of the first chunk (0.9 ms into a 6.5 ms frame share).
    const uint32_t shade_wgs0 = shade_wgs;
    const uint32_t n_chunks = (bands_total + bands_per_chunk - 1) / bands_per_chunk;
    for (uint32_t chunk0 = 0; chunk0 < n_chunks; chunk0 += n_conc) {
        const uint32_t n_group = std::min(n_conc, n_chunks - chunk0);
        rtf::FrontArgs args_of[kFrontStreamsMax];
        for (uint32_t gi = 0; gi < n_group; ++gi) {
            const uint32_t chunk = chunk0 + gi, band0 = chunk * bands_per_chunk;
            const uint32_t nb = std::min(bands_per_chunk, bands_total - band0);
            char* const B = ws_base + (size_t)(chunk % n_conc) * ws_one;
            rtf::FrontArgs A{};
            A.W = p->width; A.H = p->height; A.spp = p->spp; A.max_bounces = p->max_bounces; A.seed = p->seed;
            A.row_begin = p->row_begin; A.row_stride = p->row_stride; A.flags = p->flags;
            A.row0 = band0 * 8; A.rows = std::min(nb * 8, n_rows - band0 * 8);
            A.tiles_x = tiles_x; A.n_tiles = tiles_x * nb; A.npix = A.n_tiles * 64u; A.np = A.npix * p->spp;
            A.pf_n = pf_n; A.pu_n = pu_n;
            A.cull = (stats && !(prof && prof->keep_cull) && !std::getenv("RTM_DIAG_KEEP_CULL")) ? 0u : g_front_cull;
            // Scenes where most non-empty cells are long (config 5's soup): the box screen leaves about two entries per cell and every
            // cell a ray meets is opened, so two test slots per step and an earlier switch from walking to testing (soup -12 %);
            // a scene with a few long cells (suzanne: lists up to 37) keeps the settings of short lists.
            A.pool_refill = g_pool_refill; A.pool_walk_more = g_pool_walk_more;
            A.refill = g_front_refill; A.t_test_frac = use_pool ? g_pool_test_frac : (dense ? g_t_test_frac_long : g_t_test_frac);
            A.wide = g_front_wide ? g_front_wide : (dense ? 2u : (uint32_t)RTM_FTRI_WIDE); A.t_min_lanes = g_t_min_lanes;
            // (scenes with many analytic primitives -- multisphere: 125 discriminants per query -- are bound by that arithmetic: they gain
            // nothing from skipping the state's round trip and lose the compaction: measured 42 -> 44-68 ms, profiles/r03_sweep_inlane.log)
            // (the light batches in the lane only where the general in-lane advance is off: multisphere 36.9 -> 35.7 ms; simple_sphere at 20 spp loses 7 % with it)
            A.inlane = inlane; A.inlane_lights = inlane_on > 64u ? g_front_inlane_lights : 0u;
            A.claim = std::max<uint32_t>(16u, g_front_claim);  // (a claim of nothing would spin the trace kernel for ever)
            A.sph_screen = g_front_sph_screen;
            A.steal = g_front_steal; A.coop = g_front_coop; A.att_replay = (p->max_bounces <= 16u && sd->view.n_materials <= 0xFFFFu) ? g_front_rehit : 0u; A.rehit = (sd->view.n_spheres + sd->view.n_planes <= 4u) ? g_front_rehit : 0u;
            A.occ_words = occ_words;
            A.occ_plain = occ_plain_only ? 1u : 0u;
            // which bitmap word the walk probes (rt_front.h probe()): the arrival direction's, the plain occupancy when
            // nothing may be skipped (counting build), or the one staged bitmap
            A.occ_shift = occ_plain_only ? 0u : 3u;
            A.occ_m1 = (!occ_plain_only && A.cull) ? 7u : 0u;
            A.occ_m2 = (!occ_plain_only && !A.cull) ? 6u : 0u;
            A.D[0] = (double*)(B + o_D0); A.D[1] = (double*)(B + o_D1); A.U[0] = (uint32_t*)(B + o_U0); A.U[1] = (uint32_t*)(B + o_U1);
            A.fin = (double*)(B + o_fin); A.res = (rtf::FrontResult*)(B + o_res);
            A.itD = (double*)(B + o_itD); A.itU = (uint32_t*)(B + o_itU);
            {   // PU_HOME's fields (rt_front.h): 6 bits pixel of the tile, then tile column, tile row of the chunk, sample
                auto bits = [](uint32_t n) { uint32_t b = 1; while ((1u << b) < n) ++b; return b; };
                const uint32_t bx = bits(tiles_x), by = bits(bands_per_chunk);
                A.home_sy = 6u + bx; A.home_ss = 6u + bx + by; A.home_mx = (1u << bx) - 1u; A.home_my = (1u << by) - 1u;
            }
            A.nq = nq; A.cap_l = ((((A.np + 63u) / 64u) + nq - 1) / nq) * 64u; A.cap_i = A.cap_l * 3u; A.cap_items = nq * A.cap_i;
            A.ctr = (uint32_t*)(B + o_ctr);
            args_of[gi] = A;
        }
#ifdef RTM_WAVE_TIMES
        static unsigned long long* wt_buf = nullptr;  // diagnostics build: per-wave times of every trace launch of chunk 0
        if (!wt_buf) HIP_TRY(hipMalloc((void**)&wt_buf, (size_t)64 * 8192 * 4 * 8));
        HIP_TRY(hipMemsetAsync(wt_buf, 0, (size_t)64 * 8192 * 4 * 8, lanes[0]));
#endif
        for (uint32_t r = 0; r < rounds; ++r) {
            for (uint32_t gi = 0; gi < n_group; ++gi) {
#ifdef RTM_WAVE_TIMES
                args_of[gi].dbg = (gi == 0 && chunk0 == 0 && r < 64) ? wt_buf + (size_t)r * 8192 * 4 : nullptr;
#endif
                const rtf::FrontArgs& A = args_of[gi];
                hipStream_t stream = lanes[(chunk0 + gi) % n_conc];  // (shadows the caller's stream inside the loop)
                unsigned long long* shade_cnt = d_cnt;
#ifdef RTM_SHADE_TIMES
                static unsigned long long* st_buf = nullptr;  // diagnostics build: cycles per phase of the shade kernel's iterations
                if (!st_buf) { HIP_TRY(hipMalloc((void**)&st_buf, 40 * 8)); }
                if (!d_cnt) { HIP_TRY(hipMemsetAsync(st_buf, 0, 40 * 8, stream)); shade_cnt = st_buf; }
#endif
                mark(0, false, stream);
                {
                    const bool tabs = sd->view.n_lights <= (uint32_t)rtf::kShadeLights && sd->view.n_materials <= (uint32_t)rtf::kShadeMats &&
                                      sd->view.n_models <= (uint32_t)rtf::kShadeModels && sd->view.n_spheres <= (uint32_t)rtf::kShadeSpheres &&
                                      sd->view.n_planes <= (uint32_t)rtf::kShadePlanes;
                    const bool many = sd->view.n_spheres > 4u;
                    const size_t slds = (tabs ? rtf::kShadeLdsBytes : 0) + (many ? rtf::kShadePackBytes : 0) + (many && tabs ? rtf::kShadeSphBytes : 0);
                    switch ((many ? 8 : 0) | (stats ? 4 : 0) | (r == 0 ? 2 : 0) | (tabs ? 1 : 0)) {
#define RTM_SHADE_CASE(v, S_, R_, T_) \
    case v: hipLaunchKernelGGL((rtf::front_shade_kernel<S_, R_, T_, false>), dim3(shade_wgs0), block, slds, stream, sd->view, A, r, d_obj, d_tri, shade_cnt); break; \
    case v + 8: hipLaunchKernelGGL((rtf::front_shade_kernel<S_, R_, T_, true>), dim3(shade_wgs0), block, slds, stream, sd->view, A, r, d_obj, d_tri, shade_cnt); break;
                        RTM_SHADE_CASE(0, false, false, false) RTM_SHADE_CASE(1, false, false, true)
                        RTM_SHADE_CASE(2, false, true, false) RTM_SHADE_CASE(3, false, true, true)
                        RTM_SHADE_CASE(4, true, false, false) RTM_SHADE_CASE(5, true, false, true)
                        RTM_SHADE_CASE(6, true, true, false) RTM_SHADE_CASE(7, true, true, true)
#undef RTM_SHADE_CASE
                    }
                }
                mark(0, true, stream);
#ifdef RTM_SHADE_TIMES
                if (!d_cnt && gi == 0 && std::getenv("RTM_SHADE_TIMES")) {
                    unsigned long long h5[5];
                    HIP_TRY(hipStreamSynchronize(stream));
                    HIP_TRY(hipMemcpy(h5, st_buf + 34, sizeof h5, hipMemcpyDeviceToHost));
                    if (h5[4])
                        std::fprintf(stderr, "[rtm shade times] round %u: %llu wave iterations; cycles per iteration: state loads %.0f, answers %.0f, compute %.0f, stores %.0f\n",
                                     r, h5[4], (double)h5[0] / h5[4], (double)h5[1] / h5[4], (double)h5[2] / h5[4], (double)h5[3] / h5[4]);
                }
#endif
                if (prof && prof->d_items && !prof->keep_cull)
                    hipLaunchKernelGGL(front_sum_items_kernel, dim3(32), dim3(256), 0, stream, A.ctr, nq, prof->d_items);
                if (r + 1 < rounds && !no_models) {  // the last round only finalises; scenes without models have no trace items
                    mark(1, false, stream);
                    const size_t lds = (size_t)occ_words * 4;
                    const int variant = (stats ? 4 : 0) | (all_lds ? 2 : 0) | (sd->n_long_cells ? 1 : 0);
                    if (use_pool) {
                        const int pv = (stats ? 4 : 0) | (pool_long ? 2 : 0) | (pool_multi ? 1 : 0);
                        switch (pv) {
#define RTM_POOL_LAUNCH(v, S_, L_, M_, R_)                                                                                                  \
    {                                                                                                                                       \
        static bool attr_set = false;                                                                                                       \
        if (!attr_set) {                                                                                                                    \
            HIP_TRY(hipFuncSetAttribute((const void*)&rtf::pool_trace_kernel<S_, L_, M_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                        (int)kLdsDynMax));                                                                                  \
            attr_set = true;                                                                                                                \
        }                                                                                                                                   \
        hipLaunchKernelGGL((rtf::pool_trace_kernel<S_, L_, M_, R_>), dim3((uint32_t)g_num_cus), dim3(pool_waves * 64u), pool_lds, stream,   \
                           sd->view, A, r, d_cnt);                                                                                          \
    }
#define RTM_POOL_CASE(v, S_, L_, M_)                                                                                                        \
    case v:                                                                                                                                 \
        RTM_POOL_LAUNCH(v, S_, L_, M_, 96)                                   \
        break;
                            RTM_POOL_CASE(0, false, false, false) RTM_POOL_CASE(1, false, false, true)
                            RTM_POOL_CASE(2, false, true, false) RTM_POOL_CASE(3, false, true, true)
                            RTM_POOL_CASE(4, true, false, false) RTM_POOL_CASE(5, true, false, true)
                            RTM_POOL_CASE(6, true, true, false) RTM_POOL_CASE(7, true, true, true)
#undef RTM_POOL_LAUNCH
#undef RTM_POOL_CASE
                        }
                    } else
                    switch (variant) {
#define RTM_TRACE_CASE(v, S_, L_, G_) \
    case v: hipLaunchKernelGGL((rtf::front_trace_kernel<S_, L_, G_>), dim3(trace_wgs), block, lds, stream, sd->view, A, r, d_cnt); break;
                        RTM_TRACE_CASE(0, false, false, false) RTM_TRACE_CASE(1, false, false, true)
                        RTM_TRACE_CASE(2, false, true, false) RTM_TRACE_CASE(3, false, true, true)
                        RTM_TRACE_CASE(4, true, false, false) RTM_TRACE_CASE(5, true, false, true)
                        RTM_TRACE_CASE(6, true, true, false) RTM_TRACE_CASE(7, true, true, true)
#undef RTM_TRACE_CASE
                    }
                    mark(1, true, stream);
                }
                HIP_TRY(hipGetLastError());
                if (g_front_debug_rounds && !fixed) {  // RTM_FRONT_DEBUG_ROUNDS=1: list lengths after every round (synchronises; diagnostics only)
                    std::vector<uint32_t> h((size_t)nq * rtf::CT_STRIDE);
                    HIP_TRY(hipStreamSynchronize(stream));
                    HIP_TRY(hipMemcpy(h.data(), A.ctr, h.size() * 4, hipMemcpyDeviceToHost));
                    uint64_t live = 0, items = 0;
                    uint32_t live_max = 0, items_max = 0;
                    for (uint32_t j = 0; j < nq; ++j) {
                        const uint32_t l = h[(size_t)j * rtf::CT_STRIDE + ((r + 1u) & 1u)];
                        const uint32_t* const ci = &h[(size_t)j * rtf::CT_STRIDE + rtf::CT_ITEMS0];
                        const uint32_t it = ci[0] + ci[1] + ci[2];
                        live += l; items += it;
                        live_max = std::max(live_max, l); items_max = std::max(items_max, it);
                    }
                    std::fprintf(stderr, "[rtm front] round %u: live %llu (max %u per sub-list of %u), items %llu (max %u)\n", r,
                                 (unsigned long long)live, live_max, nq, (unsigned long long)items, items_max);
                    if (stats) {  // trace diagnostics of this round (the counters are cumulative; the maximum is reset here)
                        static unsigned long long prev[32];
                        unsigned long long h2[32];
                        HIP_TRY(hipMemcpy(h2, d_cnt, sizeof h2, hipMemcpyDeviceToHost));
                        HIP_TRY(hipMemset(d_cnt + 21, 0, 8));
                        HIP_TRY(hipMemset(d_cnt + 26, 0, 16));
                        if (r == 0) std::memset(prev, 0, sizeof prev);
                        auto dl = [&](int i) { return (double)(h2[i] - prev[i]); };
                        std::fprintf(stderr, "[rtm front]   trace: slowest wave %llu ticks with %llu iterations (most walk iterations of a ray %llu, most test steps %llu), mean wave %.0f ticks; walk iters %.0f (%.1f lanes) test steps %.0f (%.1f lanes); of wave time: refill %.1f%% walk %.1f%% test %.1f%%\n",
                                     h2[21] >> 24, h2[21] & 0xFFFFFFull, h2[26], h2[27], dl(22) > 0 ? dl(20) / dl(22) : 0.0, dl(11), dl(11) > 0 ? dl(12) / dl(11) : 0.0, dl(9),
                                     dl(9) > 0 ? dl(10) / dl(9) : 0.0, 100.0 * dl(23) / dl(20), 100.0 * dl(24) / dl(20), 100.0 * dl(25) / dl(20));
                        std::memcpy(prev, h2, sizeof prev);
                        prev[21] = 0;
                        prev[26] = 0; prev[27] = 0;
                    }
                }
            }
        }
#ifdef RTM_WAVE_TIMES
        if (const char* path = std::getenv("RTM_WAVE_TIMES")) {  // written by the diagnostics build only
            if (chunk0 == 0) {
                std::vector<unsigned long long> h((size_t)64 * 8192 * 4);
                HIP_TRY(hipDeviceSynchronize());
                HIP_TRY(hipMemcpy(h.data(), wt_buf, h.size() * 8, hipMemcpyDeviceToHost));
                if (FILE* f = std::fopen(path, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
            }
        }
#endif
        for (uint32_t gi = 0; gi < n_group; ++gi) {
            const rtf::FrontArgs& A = args_of[gi];
            hipStream_t stream = lanes[(chunk0 + gi) % n_conc];
            mark(2, false, stream);
            hipLaunchKernelGGL(rtf::front_resolve_kernel, dim3((A.npix + 255) / 256), block, 0, stream, A, d_rgb);
            mark(2, true, stream);
            HIP_TRY(hipGetLastError());
        }
    }
    if (n_conc > 1) {  // join: the caller's stream continues after every internal stream
        for (uint32_t i = 0; i < n_conc; ++i) {
            HIP_TRY(hipEventRecord(g_front_join[i], lanes[i]));
            HIP_TRY(hipStreamWaitEvent(stream, g_front_join[i], 0));
        }
    }
    return mark_rc;
}

// Entry and exit of every call that uses the process-wide device state (see g_render_busy): one host thread at a time;
// the frame starts after the previous frame's last kernel, whatever stream that was queued on.
struct FrameGuard {
    bool owned = false;
    int enter(hipStream_t stream) {
        if (g_render_busy.exchange(true)) return rtm_internal_fail(RTM_ERR_STATE, "another thread is inside a render call (the library is not re-entrant)");
        owned = true;
        if (g_frame_done_valid) HIP_TRY(hipStreamWaitEvent(stream, g_frame_done, 0));
        return RTM_OK;
    }
    int leave(hipStream_t stream) {
        HIP_TRY(hipEventRecord(g_frame_done, stream));
        g_frame_done_valid = true;
        return RTM_OK;
    }
    ~FrameGuard() {
        if (owned) g_render_busy.store(false);
    }
};
}  // namespace

int rtm_render_device(const rtm_scene_dev* sd, const rtm_params* p, double* d_rgb, uint32_t* d_obj, uint32_t* d_tri,
                      rtm_stats* out_stats, void* stream_) {
    if (int rc = need_init()) return rc;
    if (!sd || !d_rgb) return rtm_internal_fail(RTM_ERR_INVALID, "scene or output is null");
    if (int rc = check_params(p)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const uint32_t n_rows = rtm_local_rows(p);
    const dim3 block(256);
    rtw::RenderArgsW AW{p->width, p->height, p->spp, p->max_bounces, p->seed, p->row_begin, p->row_stride, n_rows, p->flags,
                        g_s_threshold, g_t_test_frac, g_t_min_lanes, (p->width + 7) / 8, ((p->width + 7) / 8) * ((n_rows + 7) / 8), 0u, g_work_counter};
    {   // occupancy bitmaps that fit the LDS budget, models in order (same rule as the kernel's staging loop)
        uint32_t words = 0;
        for (uint32_t w : sd->occ_words) {
            if (words + w > (uint32_t)rtw::kOccLdsWordsMax) break;
            words += w;
        }
        AW.occ_words = words;
    }
    const size_t wave_lds = (size_t)rtw::kColdBytes + (size_t)AW.occ_words * 4;
    // persistent grid: enough workgroups to fill every CU at the kernel's occupancy; each wave claims 8x8 tiles
    static const uint32_t wgs_env = std::getenv("RTM_WGS_PER_CU") ? (uint32_t)std::max(1, std::atoi(std::getenv("RTM_WGS_PER_CU"))) : 0u;
    const uint32_t wgs_per_cu = wgs_env ? wgs_env : (uint32_t)RTM_WAVE_OCC;
    const uint32_t wave_wgs = std::min<uint32_t>((AW.n_tiles + 3) / 4, (uint32_t)g_num_cus * wgs_per_cu);
    const bool front = !(p->flags & RTM_FLAG_KERNEL_WAVE) && g_kernel == 2 && front_supported(sd, p);
    FrameGuard guard;
    if (int rc = guard.enter(stream)) return rc;
    int front_rc = RTM_OK;
    auto launch = [&](bool stats, unsigned long long* d_cnt) {
        if (front) {
            front_rc = render_front(sd, p, n_rows, front_occ_words(sd), d_rgb, d_obj, d_tri, stats ? d_cnt : nullptr, stream);
            return front_rc == RTM_OK ? hipSuccess : hipErrorUnknown;
        }
        hipError_t me = hipMemsetAsync(g_work_counter, 0, 256, stream);  // claimed tiles start at 0 every frame
        if (me != hipSuccess) return me;
        if (stats) hipLaunchKernelGGL(rtw::render_wave_kernel<true>, dim3(wave_wgs), block, wave_lds, stream, sd->view, AW, d_rgb, d_obj, d_tri, d_cnt);
        else hipLaunchKernelGGL(rtw::render_wave_kernel<false>, dim3(wave_wgs), block, wave_lds, stream, sd->view, AW, d_rgb, d_obj, d_tri, d_cnt);
        return hipGetLastError();
    };
    if (out_stats) {
        unsigned long long* d_cnt = nullptr;
        HIP_TRY(hipMalloc((void**)&d_cnt, 40 * sizeof(unsigned long long)));
        hipError_t e = hipMemsetAsync(d_cnt, 0, 40 * sizeof(unsigned long long), stream);
        if (e == hipSuccess) e = launch(true, d_cnt);
        unsigned long long h[32] = {0};
        if (e == hipSuccess) e = hipMemcpyAsync(h, d_cnt, sizeof h, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        (void)hipFree(d_cnt);
        if (front_rc != RTM_OK) return front_rc;
        if (e != hipSuccess) return rtm_internal_fail(RTM_ERR_DEVICE, std::string("render (stats): ") + hipGetErrorString(e));
        if (int rc = guard.leave(stream)) return rc;
        if (std::getenv("RTM_DIAG") && front && h[22])
            std::fprintf(stderr, "[rtm diag front] trace waves %llu: mean %.0f cycles, sum of per-launch max not tracked; max single wave %llu cycles; total wave-cycles %.3g: refill %.1f%% (%.0f/refill)  walk %.1f%% (%.0f/iter)  test %.1f%% (%.0f/step)\n",
                         h[22], (double)h[20] / (double)h[22], h[21] >> 24, (double)h[20], 100.0 * h[23] / (double)h[20], h[15] ? (double)h[23] / h[15] : 0.0,
                         100.0 * h[24] / (double)h[20], h[11] ? (double)h[24] / h[11] : 0.0, 100.0 * h[25] / (double)h[20], h[9] ? (double)h[25] / h[9] : 0.0);
        if (std::getenv("RTM_DIAG") && front)
            std::fprintf(stderr, "[rtm diag front] test steps %llu (avg %.1f lanes)  walk iters %llu (avg %.1f lanes)  test slots %llu (avg %.1f lanes)  refills %llu (avg %.1f lanes)\n",
                         h[9], h[9] ? (double)h[10] / (double)h[9] : 0.0, h[11], h[11] ? (double)h[12] / (double)h[11] : 0.0, h[13],
                         h[13] ? (double)h[14] / (double)h[13] : 0.0, h[15], h[15] ? (double)h[16] / (double)h[15] : 0.0);
        if (std::getenv("RTM_DIAG") && front && (h[9] + h[11]))
            std::fprintf(stderr, "[rtm diag front] per step (walk or test), lanes on average: idle %.1f, in another state or model %.1f; testers waiting in a walk step %.1f, walkers waiting in a test step %.1f\n",
                         (double)h[28] / (double)(h[9] + h[11]), (double)h[29] / (double)(h[9] + h[11]), h[11] ? (double)h[30] / (double)h[11] : 0.0, h[9] ? (double)h[31] / (double)h[9] : 0.0);
        if (std::getenv("RTM_DIAG") && !front && h[9])
            std::fprintf(stderr, "[rtm diag] S rounds %llu (avg %.1f lanes, %.0f cyc/round)  T steps %llu (avg %.1f lanes, %.0f cyc/step)  wave cycles: S %.1f%% T %.1f%% of %.3g\n",
                         h[9], (double)h[10] / (double)h[9], (double)h[13] / (double)h[9], h[11], h[11] ? (double)h[12] / (double)h[11] : 0.0,
                         h[11] ? (double)h[14] / (double)h[11] : 0.0, 100.0 * (double)h[13] / (double)h[15], 100.0 * (double)h[14] / (double)h[15], (double)h[15]);
        if (std::getenv("RTM_DIAG") && !front && h[9])
            std::fprintf(stderr, "[rtm diag] walks %llu wave-steps, %llu wave-iters (avg %.1f lanes); test slots %llu (avg %.1f lanes); bbox %llu cells %llu tri %llu mhit %llu\n",
                         h[20], h[16], h[16] ? (double)h[17] / (double)h[16] : 0.0, h[18], h[18] ? (double)h[19] / (double)h[18] : 0.0, h[5], h[6], h[7], h[8]);
        if (std::getenv("RTM_DIAG") && !front && h[9] && h[24])
            std::fprintf(stderr, "[rtm diag] S sub-blocks (cyc/round): decide+rng %.0f  normal %.0f  shadow+finish+sample %.0f  next-ray %.0f  analytic %.0f  enter %.0f\n",
                         (double)h[24] / (double)h[9], (double)h[25] / (double)h[9], (double)h[26] / (double)h[9], (double)h[27] / (double)h[9],
                         (double)h[28] / (double)h[9], (double)h[29] / (double)h[9]);
        out_stats->rays_camera = h[0]; out_stats->rays_bounce = h[1]; out_stats->rays_shadow = h[2];
        out_stats->sphere_tests = h[3]; out_stats->plane_tests = h[4]; out_stats->bbox_tests = h[5];
        out_stats->cells_visited = h[6]; out_stats->tri_tests = h[7]; out_stats->model_hits = h[8];
        return RTM_OK;
    }
    {
        const hipError_t e = launch(false, nullptr);
        if (front_rc != RTM_OK) return front_rc;
        HIP_TRY(e);
    }
    return guard.leave(stream);
}

// ---- the entry point that allocates nothing ----------------------------------------------------------------------
// The work space of the ray-front pipeline (path state, query results, ray items, sub-queue counters: rt_front.h) in a
// static device array instead of hipMalloc'ed memory.  Each GPU the library runs on holds its own copy, at its own
// address, until the process ends; rtm_init asks for that address once per device (the array's NAME is only a host-side
// stand-in for the symbol and is never handed to a kernel or a memset).
constexpr size_t kStaticScratchBytes = (size_t)RTM_NOALLOC_SCRATCH_MIB << 20;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
constexpr int kMaxDevices = 64;
static char* g_scratch_ptr[kMaxDevices] = {};  // per device: where that device's copy of `scratch` lives

static int scratch_bind(int device) {
    if (device < 0 || device >= kMaxDevices) return rtm_internal_fail(RTM_ERR_INVALID, "device index beyond the library's table of devices");
    if (g_scratch_ptr[device]) return RTM_OK;
    void* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));  // (the current device is `device`: rtm_init has just selected it)
    if (e != hipSuccess || !ptr) return rtm_internal_fail(RTM_ERR_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    g_scratch_ptr[device] = (char*)ptr;
    return RTM_OK;
}

uint64_t rtm_noalloc_scratch_bytes(void) { return (uint64_t)kStaticScratchBytes; }

int rtm_render_device_noalloc(const rtm_scene_dev* sd, const rtm_params* p, double* d_rgb, uint32_t* d_obj, uint32_t* d_tri, void* stream_) {
    if (int rc = need_init()) return rc;
    if (!sd || !d_rgb) return rtm_internal_fail(RTM_ERR_INVALID, "scene or output is null");
    if (int rc = check_params(p)) return rc;
    if (p->flags & RTM_FLAG_KERNEL_WAVE) return rtm_internal_fail(RTM_ERR_INVALID, "rtm_render_device_noalloc runs the ray-front pipeline only");
    if (!front_supported(sd, p)) return rtm_internal_fail(RTM_ERR_INVALID, "scene or frame beyond the ray-front pipeline's limits (grid > 512 cells per side, > 429 496 729 list entries or >= 2^27 cells over all models, > 65 535 bounces)");
    if (sd->device != g_device || !g_scratch_ptr[g_device]) return rtm_internal_fail(RTM_ERR_STATE, "scene was uploaded to another device, or rtm_init has not bound the scratch array");
    if (g_render_busy.exchange(true)) return rtm_internal_fail(RTM_ERR_STATE, "another thread is inside a render call (the library is not re-entrant)");
    const FrontFixedWs ws{g_scratch_ptr[g_device], kStaticScratchBytes};
    // Only kernel launches from here on: no allocation, no event, no memset (every counter word the kernels read is
    // written by the shade kernel of the same round; the scratch keeps the last call's contents and none of them is read
    // before it is rewritten).
    const int rc = render_front(sd, p, rtm_local_rows(p), front_occ_words(sd), d_rgb, d_obj, d_tri, nullptr, (hipStream_t)stream_, nullptr, &ws);
    g_render_busy.store(false);
    return rc;
}

int rtm_time_render_device(const rtm_scene_dev* sd, const rtm_params* p, double* d_rgb, int iters, void* stream_, double* out_ms) {
    if (!out_ms || iters <= 0) return rtm_internal_fail(RTM_ERR_INVALID, "bad iters/out");
    hipStream_t stream = (hipStream_t)stream_;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    int rc = RTM_OK;
    hipError_t e = hipEventRecord(e0, stream);
    for (int i = 0; i < iters && rc == RTM_OK; ++i) rc = rtm_render_device(sd, p, d_rgb, nullptr, nullptr, nullptr, stream_);
    if (e == hipSuccess) e = hipEventRecord(e1, stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != RTM_OK) return rc;
    if (e != hipSuccess) return rtm_internal_fail(RTM_ERR_DEVICE, std::string("timing: ") + hipGetErrorString(e));
    *out_ms = (double)ms / iters;
    return RTM_OK;
}

int rtm_profile_render_device(const rtm_scene_dev* sd, const rtm_params* p, double* d_rgb, void* stream_, rtm_profile* out) {
    if (int rc = need_init()) return rc;
    if (!sd || !d_rgb || !out) return rtm_internal_fail(RTM_ERR_INVALID, "scene, output or profile is null");
    if (int rc = check_params(p)) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    *out = rtm_profile{};
    const uint32_t n_rows = rtm_local_rows(p);
    const bool front = !(p->flags & RTM_FLAG_KERNEL_WAVE) && g_kernel == 2 && front_supported(sd, p);
    if (!front) {  // one kernel: its time and the reference-equivalent counters
        double ms = 0.0;
        if (int rc = rtm_time_render_device(sd, p, d_rgb, 1, stream_, &ms)) return rc;
        rtm_stats st{};
        if (int rc = rtm_render_device(sd, p, d_rgb, nullptr, nullptr, &st, stream_)) return rc;
        out->total_ms = out->trace_ms = ms;
        out->trace_launches = 1;
        out->cells_visited = st.cells_visited; out->tri_tests = st.tri_tests;
        out->rays_total = st.rays_camera + st.rays_bounce + st.rays_shadow;
        out->traced_rays = st.bbox_tests;
        out->sphere_tests = st.sphere_tests; out->plane_tests = st.plane_tests; out->bbox_tests = st.bbox_tests;
        return RTM_OK;
    }
    const uint32_t occ_words = front_occ_words(sd);
    FrameGuard guard;
    if (int rc = guard.enter(stream)) return rc;
    unsigned long long* d_buf = nullptr;  // [0..31] counters of the counting run, [32] traced rays
    HIP_TRY(hipMalloc((void**)&d_buf, 40 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(d_buf, 0, 40 * sizeof(unsigned long long), stream));
    FrontProfile prof;
    prof.d_items = d_buf + 32;
    int rc = render_front(sd, p, n_rows, occ_words, d_rgb, nullptr, nullptr, nullptr, stream, &prof);
    hipError_t e = hipStreamSynchronize(stream);
    if (std::getenv("RTM_PROFILE_ROUNDS") && rc == RTM_OK && e == hipSuccess) {  // per-launch durations, in launch order
        for (int k = 0; k < 2; ++k) {
            std::fprintf(stderr, "[rtm profile] %s launches (ms):", k ? "trace" : "shade");
            for (auto& pr : prof.ev[k]) {
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, pr.first, pr.second);
                std::fprintf(stderr, " %.3f", ms);
            }
            std::fprintf(stderr, "\n");
        }
    }
    double* const slot[3] = {&out->shade_ms, &out->trace_ms, &out->resolve_ms};
    for (int k = 0; k < 3; ++k)
        for (auto& pr : prof.ev[k]) {
            float ms = 0.f;
            if (rc == RTM_OK && e == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) *slot[k] += (double)ms;
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    out->shade_launches = (uint32_t)prof.ev[0].size();
    out->trace_launches = (uint32_t)prof.ev[1].size();
    out->total_ms = out->shade_ms + out->trace_ms + out->resolve_ms;
    if (rc == RTM_OK && e == hipSuccess) {
        FrontProfile cnt_run;
        cnt_run.keep_cull = true;
        rc = render_front(sd, p, n_rows, occ_words, d_rgb, nullptr, nullptr, d_buf, stream, &cnt_run);
        unsigned long long h[40] = {0};
        if (rc == RTM_OK) e = hipMemcpyAsync(h, d_buf, sizeof h, hipMemcpyDeviceToHost, stream);
        if (rc == RTM_OK && e == hipSuccess) e = hipStreamSynchronize(stream);
        out->rays_total = h[0] + h[1] + h[2] - h[33];  // (h[33]: shadow queries answered without a ray)
        out->cells_visited = h[6];
        out->tri_tests = h[7];
        out->traced_rays = h[32];
        out->sphere_tests = h[3]; out->plane_tests = h[4]; out->bbox_tests = h[5];
    }
    (void)hipFree(d_buf);
    if (rc != RTM_OK) return rc;
    if (e != hipSuccess) return rtm_internal_fail(RTM_ERR_DEVICE, std::string("profile: ") + hipGetErrorString(e));
    return guard.leave(stream);
}

int rtm_timed_events_collect(double* out_trace_ms, uint32_t* out_trace_launches) {
    if (int rc = need_init()) return rc;
    if (!out_trace_ms || !out_trace_launches) return rtm_internal_fail(RTM_ERR_INVALID, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    double total = 0.0;
    uint32_t n = 0;
    for (size_t i = 0; i < g_front_ev_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_front_ev_pool[i].first, g_front_ev_pool[i].second) == hipSuccess) {
            total += (double)ms;
            ++n;
        }
    }
    g_front_ev_used = 0;
    *out_trace_ms = total;
    *out_trace_launches = n;
    return RTM_OK;
}

int rtm_render(const rtm_scene* scene, const rtm_params* p, double* out_rgb, uint32_t* out_obj, uint32_t* out_tri, rtm_stats* out_stats) {
    if (int rc = need_init()) return rc;
    if (!out_rgb) return rtm_internal_fail(RTM_ERR_INVALID, "out_rgb is null");
    if (int rc = check_params(p)) return rc;
    rtm_scene_dev* sd = nullptr;
    if (int rc = rtm_scene_upload(scene, &sd)) return rc;
    const size_t npx = (size_t)rtm_local_rows(p) * p->width;
    double* d_rgb = nullptr;
    uint32_t *d_obj = nullptr, *d_tri = nullptr;
    int rc = RTM_OK;
    auto dev = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == RTM_OK) rc = rtm_internal_fail(RTM_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
        return e == hipSuccess;
    };
    dev(hipMalloc((void**)&d_rgb, npx * 24), "hipMalloc rgb");
    if (rc == RTM_OK && out_obj) dev(hipMalloc((void**)&d_obj, npx * 4), "hipMalloc obj");
    if (rc == RTM_OK && out_tri) dev(hipMalloc((void**)&d_tri, npx * 4), "hipMalloc tri");
    if (rc == RTM_OK) rc = rtm_render_device(sd, p, d_rgb, d_obj, d_tri, out_stats, nullptr);
    if (rc == RTM_OK) dev(hipDeviceSynchronize(), "render");
    if (rc == RTM_OK) dev(hipMemcpy(out_rgb, d_rgb, npx * 24, hipMemcpyDeviceToHost), "copy rgb");
    if (rc == RTM_OK && out_obj) dev(hipMemcpy(out_obj, d_obj, npx * 4, hipMemcpyDeviceToHost), "copy obj");
    if (rc == RTM_OK && out_tri) dev(hipMemcpy(out_tri, d_tri, npx * 4, hipMemcpyDeviceToHost), "copy tri");
    (void)hipFree(d_rgb);
    (void)hipFree(d_obj);
    (void)hipFree(d_tri);
    rtm_scene_free(sd);
    return rc;
}

int rtm_trace_rays(const rtm_scene_dev* sd, uint64_t n, const double* rays6, uint32_t* out_obj, uint32_t* out_tri, double* out_t,
                   double* out_nrm) {
    if (int rc = need_init()) return rc;
    if (!sd || !rays6 || !out_obj || !out_tri || !out_t || !out_nrm) return rtm_internal_fail(RTM_ERR_INVALID, "null argument");
    if (n == 0) return RTM_OK;
    if (n > (1ull << 31)) return rtm_internal_fail(RTM_ERR_INVALID, "too many rays");
    double *d_rays = nullptr, *d_t = nullptr, *d_n = nullptr;
    uint32_t *d_o = nullptr, *d_i = nullptr;
    int rc = RTM_OK;
    auto dev = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == RTM_OK) rc = rtm_internal_fail(RTM_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    };
    dev(hipMalloc((void**)&d_rays, n * 48), "hipMalloc");
    dev(hipMalloc((void**)&d_t, n * 8), "hipMalloc");
    dev(hipMalloc((void**)&d_n, n * 24), "hipMalloc");
    dev(hipMalloc((void**)&d_o, n * 4), "hipMalloc");
    dev(hipMalloc((void**)&d_i, n * 4), "hipMalloc");
    if (rc == RTM_OK) dev(hipMemcpy(d_rays, rays6, n * 48, hipMemcpyHostToDevice), "copy rays");
    if (rc == RTM_OK) {
        hipLaunchKernelGGL(trace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, sd->view, n, d_rays, d_o, d_i, d_t, d_n);
        dev(hipGetLastError(), "trace launch");
        dev(hipDeviceSynchronize(), "trace");
    }
    if (rc == RTM_OK) {
        dev(hipMemcpy(out_obj, d_o, n * 4, hipMemcpyDeviceToHost), "copy");
        dev(hipMemcpy(out_tri, d_i, n * 4, hipMemcpyDeviceToHost), "copy");
        dev(hipMemcpy(out_t, d_t, n * 8, hipMemcpyDeviceToHost), "copy");
        dev(hipMemcpy(out_nrm, d_n, n * 24, hipMemcpyDeviceToHost), "copy");
    }
    (void)hipFree(d_rays); (void)hipFree(d_t); (void)hipFree(d_n); (void)hipFree(d_o); (void)hipFree(d_i);
    return rc;
}

int rtm_f64_ops_selftest(uint64_t n, const double* a, const double* b, double* out6) {
    if (int rc = need_init()) return rc;
    if (!a || !b || !out6 || n == 0 || n > (1ull << 28)) return rtm_internal_fail(RTM_ERR_INVALID, "bad argument");
    double *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
    int rc = RTM_OK;
    auto dev = [&](hipError_t e, const char* what) {
        if (e != hipSuccess && rc == RTM_OK) rc = rtm_internal_fail(RTM_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    };
    dev(hipMalloc((void**)&d_a, n * 8), "hipMalloc");
    dev(hipMalloc((void**)&d_b, n * 8), "hipMalloc");
    dev(hipMalloc((void**)&d_o, n * 48), "hipMalloc");
    if (rc == RTM_OK) dev(hipMemcpy(d_a, a, n * 8, hipMemcpyHostToDevice), "copy");
    if (rc == RTM_OK) dev(hipMemcpy(d_b, b, n * 8, hipMemcpyHostToDevice), "copy");
    if (rc == RTM_OK) {
        hipLaunchKernelGGL(f64_ops_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, d_a, d_b, d_o);
        dev(hipGetLastError(), "launch");
        dev(hipDeviceSynchronize(), "f64 ops");
    }
    if (rc == RTM_OK) dev(hipMemcpy(out6, d_o, n * 48, hipMemcpyDeviceToHost), "copy");
    (void)hipFree(d_a); (void)hipFree(d_b); (void)hipFree(d_o);
    return rc;
}

int rtm_tonemap_device(const double* d_rgb, uint64_t n_pixels, int gamma, uint8_t* d_rgb8, void* stream) {
    if (int rc = need_init()) return rc;
    if (!d_rgb || !d_rgb8) return rtm_internal_fail(RTM_ERR_INVALID, "null argument");
    const uint64_t n = n_pixels * 3;
    if (n == 0) return RTM_OK;
    hipLaunchKernelGGL(tonemap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_rgb, n, gamma, d_rgb8);
    HIP_TRY(hipGetLastError());
    return RTM_OK;
}

int rtm_render_rgb8(const rtm_scene_dev* sd, const rtm_params* p, int gamma, uint8_t* out_rgb8, double* out_seconds) {
    if (int rc = need_init()) return rc;
    if (!sd || !out_rgb8) return rtm_internal_fail(RTM_ERR_INVALID, "scene or output is null");
    if (int rc = check_params(p)) return rc;
    const size_t n_px = (size_t)rtm_local_rows(p) * p->width;
    if (n_px == 0) {
        if (out_seconds) *out_seconds = 0.0;
        return RTM_OK;
    }
    double* d_rgb = nullptr;
    uint8_t* d_rgb8 = nullptr;
    HIP_TRY(hipMalloc((void**)&d_rgb, n_px * 3 * sizeof(double)));
    if (hipMalloc((void**)&d_rgb8, n_px * 3) != hipSuccess) {
        (void)hipFree(d_rgb);
        return rtm_internal_fail(RTM_ERR_DEVICE, "rtm_render_rgb8: out of device memory");
    }
    // the bracket of main.rs:156-187: render, then the conversion to 8 bits, image in host memory.  Without gamma the
    // conversion runs on the device (3 bytes per pixel cross PCIe).  With gamma the f64 framebuffer comes to the host and
    // value.powf(1.0 / 2.2) (main.rs:42-48) is the host libm's pow -- what Rust's f64::powf calls -- so the bytes are the
    // reference's; the device's pow() may differ from it in the last place, which flips a code value now and then.
    const auto t0 = std::chrono::steady_clock::now();
    int rc = rtm_render_device(sd, p, d_rgb, nullptr, nullptr, nullptr, nullptr);
    hipError_t e = hipSuccess;
    if (rc == RTM_OK && gamma) {
        std::vector<double> host(n_px * 3);
        e = hipMemcpy(host.data(), d_rgb, n_px * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess)
            for (size_t i = 0; i < n_px * 3; ++i) {
                double v = host[i];
                v = v > 0.0 ? std::pow(v, 1.0 / 2.2) : 0.0;
                v = v * 255.0;
                out_rgb8[i] = !(v > 0.0) ? 0 : (v >= 255.0 ? 255 : (uint8_t)v);  // (c*255.0) as u8: truncating, saturating, NaN -> 0
            }
    } else if (rc == RTM_OK) {
        rc = rtm_tonemap_device(d_rgb, n_px, 0, d_rgb8, nullptr);
        if (rc == RTM_OK) e = hipMemcpy(out_rgb8, d_rgb8, n_px * 3, hipMemcpyDeviceToHost);
    }
    if (out_seconds) *out_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)hipFree(d_rgb);
    (void)hipFree(d_rgb8);
    if (rc != RTM_OK) return rc;
    if (e != hipSuccess) return rtm_internal_fail(RTM_ERR_DEVICE, std::string("rtm_render_rgb8: ") + hipGetErrorString(e));
    return RTM_OK;
}

int rtm_deinterleave_device(const double* d_blocks, uint32_t width, uint32_t height, uint32_t n_ranks, double* d_rgb, void* stream) {
    if (int rc = need_init()) return rc;
    if (!d_blocks || !d_rgb || n_ranks == 0 || n_ranks > 64) return rtm_internal_fail(RTM_ERR_INVALID, "bad argument");
    const uint64_t n = (uint64_t)width * height * 3;
    if (n == 0) return RTM_OK;
    hipLaunchKernelGGL(deinterleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_blocks, width, height,
                       n_ranks, d_rgb);
    HIP_TRY(hipGetLastError());
    return RTM_OK;
}

}  // extern "C"
