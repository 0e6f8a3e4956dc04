// rt_front.h -- the render path as ray "fronts": every sample path of a band of pixel rows advances one path
// vertex per round, and each round is two kernels over compacted lists in HBM
//     shade  (one lane per live path): consume the finished queries in the reference's order, create the next
//            vertex, produce its rays -- bounce ray and up to two shadow rays at once -- including the analytic
//            primitives, the slab test and the DDA initialisation (main.rs:57-132, core.rs:68-125,183-210);
//     trace  (persistent lanes fed from a queue): the 3D-DDA walk and the triangle tests (core.rs:129-167,
//            model.rs:43-74) for the rays that entered a model's box.
// Why (DESIGN.md "Kernel", measurements in profiles/): the single-kernel state machine (rt_wave.h; a multi-pixel
// variant of it was measured in round 1 and dropped) is bound by VALU issue at 25-45 % lane utilisation, because a wave always holds a mix of lanes that shade,
// walk and test.  Sorting by activity needs a pool much larger than a wave; 288 GB of HBM3E hold the state of
// tens of millions of paths, so the sort is done through memory: the shade kernel runs with every lane doing
// the same thing, and the trace kernel's lanes only ever walk or test -- a finished lane pulls the next ray from
// the queue (a dozen coalesced loads) instead of waiting for a set-up round.
// The order in which results are COMBINED is the reference's (lights in order, then the surface term, then the
// attenuation, samples in order); only the order in which rays are traced changes, and every query is a pure
// function of its ray.  The arithmetic of each block is the op sequence of rt_device.h: bit-identical results.
#pragma once
#include <type_traits>

#include "rt_device.h"

#pragma clang fp contract(off)

namespace rtf {
using namespace rtd;

#ifndef RTM_FRONT_OCC
#define RTM_FRONT_OCC 4  // waves per SIMD the trace kernel's register allocation must leave room for
#endif
#ifndef RTM_SHADE_OCC
#define RTM_SHADE_OCC 4  // waves per SIMD the shade kernel's register allocation must leave room for (128 VGPRs: shade and trace waves of concurrent chunks share a SIMD)
#endif
#ifndef RTM_MANY_OCC
#define RTM_MANY_OCC 3  // ... and for its instantiation for scenes with many spheres (168 VGPRs: with the packed pass and the f32 screen, 128 spill 116 registers -- multisphere 26.1 ms at four waves, 17.8 at three, 19.4 at two)
#endif
#ifndef RTM_FWALK_UNROLL
#define RTM_FWALK_UNROLL 4
#endif
#ifndef RTM_FTRI_WIDE
#define RTM_FTRI_WIDE 4  // list entries a lane tests per step
#endif

constexpr int kBlock = 256;
constexpr uint32_t kMaxListEntries = 0xFFFFFFFFu / 10u;  // list positions over all models: a position times 10 (doubles per record) is formed in 32 bits
constexpr int kShadeLights = 32, kShadeMats = 256, kShadeModels = 256, kShadeSpheres = 256, kShadePlanes = 64;  // table sizes the shade kernel keeps in LDS
constexpr size_t kShadeLdsBytes = (size_t)(kShadeLights * 8 + kShadeMats * 4) * 8 + (size_t)(kShadeModels + kShadeSpheres + kShadePlanes) * 2 * 4;
constexpr size_t kShadeSphBytes = (size_t)kShadeSpheres * 4 * 8;  // (MANY with TABS) the f64 spheres, behind the pack tables: what a lane reads for its screen survivors
constexpr size_t kShadePackBytes = (size_t)(kBlock / 64) * 192 * 4;  // (MANY) one table of 192 (lane, slot) words per wave, behind the tables
constexpr int kModelTab = 256;               // models whose grid word the trace kernel keeps in LDS
constexpr uint32_t kMaxCellsPerSide = 512;   // cell coordinates travel packed, 10 bits each, biased by one; the walk's countdown has 9-bit fields
constexpr uint32_t kRemGuards = (1u << 9) | (1u << 19) | (1u << 29);  // see `rem` in the trace kernel

// Per-path state, array of structures of arrays: paths in blocks of 64, a block's fields back to back, so double
// field f of path p is D[at64(pf_n, f, p)] (likewise U, the query results and the trace items).  A wave's
// accesses of one iteration are 512-byte runs inside ONE ~10 KB region; with whole-frame SoA arrays (hundreds of MB
// apart) every field sat on its own page and address translation, not HBM, set the pace of the shade kernel.
// Only what the next round cannot recompute travels: 13 doubles and 4 words per live path (11 and 3 with A.rehit and
// A.att_replay: PF_BA0, PF_ATT and PU_HKP are then neither written nor read) in scenes with at most two
// lights (the fields behind PF_EFF / PU_HOME exist for scenes with more, where a vertex waits through several rounds).
enum : int {
    PF_COL = 0,    // colour accumulated along the path (main.rs:59)
    PF_ATT = 3,    // attenuation so far
    PF_O = 4,      // origin shared by the path's queries: the camera or the vertex being shaded
    PF_D0 = 7,     // direction of the camera / bounce query
    PF_BA0 = 10,   // its closest analytic hit (spheres, planes)
    PF_EFF = 11,   // dot(direction to the light, shading normal) of the two shadow queries (main.rs:95): all that the
                   // consuming round needs of the normal
    PF_NRM = 13,   // (more than two lights) shading normal of the vertex being shaded: further batches of lights need it
    PF_DIFF = 16,  // (more than two lights) diffuse sum so far
    PF_HELD = 19,  // (more than two lights) t of slot 0's model hit, kept while further batches of shadow rays are traced
    PF_COUNT = 20
};
enum : int { PU_FLB = 0 /* FB_* flags | bounce << FB_BOUNCE_SHIFT */, PU_HKP = 1 /* slot 0's analytic hit: kind << 30 | primitive */, PU_MAT = 2,
             PU_HOME = 3 /* slot in round 0: names pixel and sample */, PU_LI = 4, PU_HTRI = 5, PU_HM = 6 /* (more than two lights) */, PU_COUNT = 7 };
enum : uint32_t {
    FB_HASV = 1u,                              // a shaded vertex waits for its shadow results
    FB_FOUND0 = 2u,                            // the path query has an analytic hit (PF_BA0, PU_HKP)
    FB_TQ0 = 16u, FB_TQ1 = 32u, FB_TQ2 = 64u,  // query j went to the trace kernel: its answer is the next one of region j of the wave's result list
    FB_C1 = 128u, FB_C2 = 256u,                // shadow query j can contribute: it faces its light (eff > 0.001, main.rs:96) and no sphere or
                                               // plane stands between (that occlusion is decided when the ray is made, see issue())
    FB_NSH_SHIFT = 9u,                         // two bits: shadow queries in flight (0..2)
    FB_HELD = 2048u,                           // slot 0's answer is in PF_HELD / PU_HTRI / PU_HM
    FB_BOUNCE_SHIFT = 12u                      // the path's bounce count (16 bits) rides in the same word
};
constexpr uint32_t kMaxFrontBounces = 0xFFFFu;
// trace items, structure of arrays over the item index.  The origin is not in the item: all queries of a path start at
// PF_O of the state the shade kernel has just written, and the item names that slot; the direction of a path query
// (region 0) is that state's PF_D0, only shadow queries bring theirs; tDelta (core.rs:119-125) is three divisions of
// numbers the trace kernel has, made where the item is read rather than carried through memory.
enum : int { IT_TM = 0, IT_D = 3, IT_DCOUNT = 6 };
enum : int { IU_CELL = 0, IU_P = 1, IU_M = 2, IU_COUNT = 3 };
enum : uint32_t { IM_SX = 1u << 16, IM_SY = 1u << 17, IM_SZ = 1u << 18 };
// device counters, one set per sub-queue (each written by the one wave that owns the sub-queue)
// (CT_CLAIM is the one word several waves touch: the trace kernel's waves claim runs of a sub-queue's items from it)
// A sub-queue has three REGIONS of cap_l items, one per query slot j: item n of region j sits at sq * cap_i + j * cap_l + n and
// its answer goes to res[] at the same index.  Within a region items are in path order, so the shade wave of the next
// round finds the answers of its paths by counting FB_TQj flags as it goes -- no result slot per path, no query id.
enum : int { CT_LIVE0 = 0, CT_LIVE1 = 1, CT_CLAIM = 2, CT_ITEMS0 = 3 /* .. +2 */, CT_STRIDE = 8 };

struct __attribute__((aligned(16))) FrontResult {  // one 16-byte store per traced ray
    double t;
    uint32_t tri, model;  // model = kNoId: no model was hit
};

struct FrontArgs {
    uint32_t W, H, spp, max_bounces;
    uint64_t seed;
    uint32_t row_begin, row_stride, flags;
    uint32_t row0, rows;        // this band: local rows [row0, row0 + rows)
    uint32_t tiles_x, n_tiles;  // 8x8 pixel tiles over W x rows; pixel ids are tile-major
    uint32_t npix, np;          // npix = n_tiles * 64 (ragged tiles padded), np = npix * spp paths.  Logical path = 64-path block
                                // (sample, tile) x lane; block b is stored at slot (b % nq) * cap_l + (b / nq) * 64: wave b % nq owns it
    uint32_t pf_n, pu_n;        // doubles / words of path state per path (the strides of D and U): 13 / 4 in scenes with at most two lights, else PF_COUNT / PU_COUNT
    uint32_t cull;              // 1: shadow queries that cannot contribute (eff <= 0.001) are not traced
    uint32_t refill;            // trace kernel: fetch new items when at least this many lanes of a wave are idle
    uint32_t claim, steal;      // trace kernel: items claimed at a time; sub-queues of other waves a wave may try when its own are empty
    uint32_t coop;              // trace kernel: 1 = when at most four rays of a wave are testing, the other lanes test their entries
    uint32_t t_test_frac, t_min_lanes;
    uint32_t att_replay;        // shade kernel: 1 = the attenuation is multiplied up again from the vertices' metal flags (16 bits beside the material) instead of travelling as a double
    uint32_t rehit;             // shade kernel: 1 = the analytic hit of a path query is computed again when its answer is taken (few primitives)
    uint32_t sph_screen;        // shade kernel (MANY): 1 = the packed-f32 screen in front of the f64 sphere tests
    uint32_t inlane_lights;     // shade kernel: 1 = the further light batches of a vertex (more than two lights) are taken in the lane
    uint32_t inlane;            // shade kernel: paths whose queries were all settled without the trace kernel advance in the lane when at least this many lanes of the wave do (65 = never)
    uint32_t wide;              // trace kernel: list entries a lane tests per step (at most RTM_FTRI_WIDE)
    uint32_t pool_refill;       // pool trace kernel (rt_pool.h): idle slots of a wave's pool that trigger a refill
    uint32_t pool_walk_more;    // pool trace kernel: a walk batch goes on while at least this many of its lanes are still walking
    uint32_t occ_words;         // bitmap words the trace kernel stages in LDS: all of g_occ7 (occ_plain = 0) or its plain occupancy alone
    uint32_t occ_plain;         // 1: only the plain occupancy (slot 6 of every group of eight words) is staged, one word per 32 cells
    uint32_t occ_shift, occ_m1, occ_m2;  // bitmap word of cell c arriving from direction a: ((c >> 5) << occ_shift) + ((a & occ_m1) | occ_m2)
    double* D[2];               // path state, streamed from one buffer to the other every round
    uint32_t* U[2];
    double* fin;                // [at64(3, c, (sample * n_tiles + tile) * 64 + pixel of the tile)]: colour of a finished sample
    uint32_t home_sy, home_ss, home_mx, home_my;  // PU_HOME = pixel of the tile | tile column << 6 | tile row << home_sy | sample << home_ss
    FrontResult* res;           // [item index]: what the trace kernel found for the query of that item
    double* itD;
    uint32_t* itU;
    uint32_t cap_items;         // nq * cap_i: stride of the item arrays
    uint32_t nq;                // sub-queues: wave w of the shade kernel reads live sub-list w % nq and appends to the same one
    uint32_t cap_l, cap_i;      // capacity of one live sub-list / one item sub-queue (cap_i = 3 * cap_l: a region of cap_l per query slot)
    uint32_t* ctr;              // [nq][CT_STRIDE]
#ifdef RTM_WAVE_TIMES
    unsigned long long* dbg;    // diagnostics build (tools/wave_times.py): [trace wave][4] = begin, queue exhausted, end (100 MHz ticks), rays taken
#endif
};

RTM_DEV size_t at64(int n_fields, int f, size_t i) { return ((i >> 6) * (size_t)n_fields + (size_t)f) * 64u + (i & 63u); }

RTM_DEV uint64_t wave_sum(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, off, 64);
        const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
template <bool STATS>
RTM_DEV void flush_counters(const Counters& cnt, unsigned long long* out_cnt) {
    if (STATS) {
        for (int i = 0; i < 9; ++i) {
            const uint64_t v = wave_sum(cnt.v[i]);
            if ((threadIdx.x & 63u) == 0u && v) atomicAdd(out_cnt + i, (unsigned long long)v);
        }
    }
}

// ================================================================================================
// shade: round 0 creates the camera rays (main.rs:160-168); later rounds consume and re-issue
// TABS: the light, material and per-model (object, material) tables are staged in LDS (scenes within kShade* entries):
// the paths of a wave are at different lights and materials, so these are per-lane lookups -- of tables that every wave
// reads all the time.
// MANY: the scene has more than four spheres, so its queries are bound by the sphere loop (multisphere: 125 discriminants per
// query): the queries of a wave's three slots are packed into passes of 64 over the spheres (see the end of the pass loop).
template <bool STATS, bool ROUND0, bool TABS, bool MANY = false>
__global__ __launch_bounds__(kBlock, MANY ? RTM_MANY_OCC : RTM_SHADE_OCC) void front_shade_kernel(RtmDevScene S, FrontArgs A, uint32_t round, uint32_t* __restrict__ out_obj,
                                                              uint32_t* __restrict__ out_tri, unsigned long long* __restrict__ out_cnt) {
    const uint32_t lane = threadIdx.x & 63u;
    // [kShadeLights * 8] lights, [kShadeMats * 4] materials, then u32 pairs: [kShadeModels] (object, material) of the models and
    // [kShadeSpheres + kShadePlanes] (material, object) of the spheres and the planes
    extern __shared__ double sh_tab[];
    if (TABS) {
        uint32_t* const sh_mod = (uint32_t*)(sh_tab + kShadeLights * 8 + kShadeMats * 4);
        uint32_t* const sh_ids = sh_mod + kShadeModels * 2;
        for (uint32_t i = threadIdx.x; i < S.n_lights * 8u; i += kBlock) sh_tab[i] = S.lights[i];
        for (uint32_t i = threadIdx.x; i < S.n_materials * 4u; i += kBlock) sh_tab[kShadeLights * 8 + i] = S.materials[i];
        for (uint32_t i = threadIdx.x; i < S.n_models; i += kBlock) { sh_mod[i * 2] = S.models[i].obj; sh_mod[i * 2 + 1] = S.models[i].mat; }
        for (uint32_t i = threadIdx.x; i < S.n_spheres * 2u; i += kBlock) sh_ids[i] = S.sph_ids[i];
        for (uint32_t i = threadIdx.x; i < S.n_planes * 2u; i += kBlock) sh_ids[kShadeSpheres * 2 + i] = S.pl_ids[i];
        if (MANY) {
            double* const sh_sph_w = (double*)((char*)sh_tab + kShadeLdsBytes + kShadePackBytes);
            for (uint32_t i = threadIdx.x; i < S.n_spheres * 4u; i += kBlock) sh_sph_w[i] = S.spheres[i];
        }
        __syncthreads();
    }
    const uint32_t* const sh_mod = (const uint32_t*)(sh_tab + kShadeLights * 8 + kShadeMats * 4);
    const uint32_t* const sh_ids = sh_mod + kShadeModels * 2;
    // (material, object) of an analytic primitive named by kind << 30 | index: not carried with the path, looked up when its hit is taken
    auto prim_ids = [&](uint32_t kp, uint32_t& mt, uint32_t& obj) __attribute__((always_inline)) {
        const uint32_t prim = kp & 0x3FFFFFFFu, plane = kp >> 30;  // (kind 0 or 1 here)
        if (TABS) { const uint32_t at = (plane ? (uint32_t)kShadeSpheres : 0u) * 2u + prim * 2u; mt = sh_ids[at]; obj = sh_ids[at + 1u]; }
        else { const uint32_t* const t = (plane ? S.pl_ids : S.sph_ids) + (size_t)prim * 2; mt = t[0]; obj = t[1]; }
    };
    auto light_at = [&](uint32_t i, int c) -> double { return TABS ? sh_tab[i * 8u + c] : S.lights[(size_t)i * 8 + c]; };
    auto light3 = [&](uint32_t i, int c) -> V3 { return mk(light_at(i, c), light_at(i, c + 1), light_at(i, c + 2)); };
    auto mat_at = [&](uint32_t i, int c) -> double { return TABS ? sh_tab[kShadeLights * 8 + i * 4u + c] : S.materials[(size_t)i * 4 + c]; };
    auto model_obj = [&](uint32_t m) -> uint32_t { return TABS ? sh_mod[m * 2u] : S.models[m].obj; };
    auto model_mat = [&](uint32_t m) -> uint32_t { return TABS ? sh_mod[m * 2u + 1u] : S.models[m].mat; };
    // The analytic primitives and the model headers are read by every query of every path, at wave-uniform addresses,
    // and nothing writes them during a frame: through the constant address space they come as scalar loads (one
    // s_load_dwordx8 per sphere) instead of one vector load per lane with its latency in every loop iteration.
    const cptr<double> sph_c = as_const(S.spheres), pl_c = as_const(S.planes);
    const cptr<uint32_t> sph_ids_c = as_const(S.sph_ids), pl_ids_c = as_const(S.pl_ids);
    const cptr<RtmDevModel> models_c = as_const(S.models);
    // One sub-list per wave (the grid has exactly nq waves): nothing is appended with atomics.  (Device-scope atomics
    // resolve outside the per-XCD L2 on this part; ~10 ns each when they share an address, which made a single queue
    // counter the bottleneck of both kernels.)
    // The state of the wave's live paths is STREAMED: round r reads slots [0, n) of its region in buffer r & 1 and
    // writes the paths that stay alive to slots [0, n') of the other buffer.  Live paths are therefore always dense in
    // memory -- late rounds, when a few per cent of the paths are left, read full cache lines instead of one 8-byte
    // field per sector -- and no list of live path ids is needed.
    const uint32_t sq = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    uint32_t* const my_ctr = A.ctr + (size_t)sq * CT_STRIDE;
    uint32_t n_live_local = 0;                       // wave-uniform
    uint32_t n_items0 = 0, n_items1 = 0, n_items2 = 0;  // items appended to the three regions of the wave's sub-queue
    uint32_t n_res0 = 0, n_res1 = 0, n_res2 = 0;        // answers of the previous round consumed so far, per region
    // round 0: the wave's paths are the 64-path blocks k * nq + sq (sample, tile), dealt round-robin for balance
    const uint32_t blocks_total = (A.np + 63u) / 64u;
    const uint32_t n_in_list = ROUND0 ? ((blocks_total + A.nq - 1u - sq) / A.nq) * 64u : my_ctr[round & 1u];
    const double* __restrict__ const Din = A.D[round & 1u];
    const uint32_t* __restrict__ const Uin = A.U[round & 1u];
    double* __restrict__ const Dout = A.D[(round + 1u) & 1u];
    uint32_t* __restrict__ const Uout = A.U[(round + 1u) & 1u];
    const int K = (int)A.max_bounces;
    const uint32_t nL = S.n_lights;
    Counters cnt;
    if (STATS)
        for (int i = 0; i < 9; ++i) cnt.v[i] = 0;
    uint64_t n_culled = 0;  // STATS: shadow queries answered without a ray (their light is behind the surface)

    // The flag word of a path is read one iteration ahead: which of the previous round's answers exist (and so where
    // they are) is known when the iteration's loads are issued, instead of one memory latency later.
    uint32_t flb_next = (!ROUND0 && lane < n_in_list) ? Uin[at64((int)A.pu_n, PU_FLB, (size_t)sq * A.cap_l + lane)] : 0u;
#ifdef RTM_SHADE_TIMES
    unsigned long long st_t[5] = {0, 0, 0, 0, 0};  // diagnostics build: cycles waiting for the loads / answers, computing, storing; iterations
#endif
    for (uint32_t k = 0; k * 64u < n_in_list; ++k) {  // wave-uniform trip count
#ifdef RTM_SHADE_TIMES
        const unsigned long long st_a = __builtin_amdgcn_s_memtime();
#endif
        const uint32_t idx = k * 64u + lane;
        bool valid = idx < n_in_list;
        const uint32_t p = sq * A.cap_l + (valid ? idx : 0u);  // slot read this round
        if (ROUND0) valid = (k * A.nq + sq) < blocks_total;
        const uint32_t flb = flb_next;
        flb_next = (!ROUND0 && idx + 64u < n_in_list) ? Uin[at64((int)A.pu_n, PU_FLB, (size_t)p + 64u)] : 0u;
        auto ldD = [&](int f) -> double { return Din[at64((int)A.pf_n, f, p)]; };
        auto ldD3 = [&](int f) -> V3 { return mk(ldD(f), ldD(f + 1), ldD(f + 2)); };
        auto ldU = [&](int f) -> uint32_t { return Uin[at64((int)A.pu_n, f, p)]; };

        // which sample of which pixel the path is: packed into one word in round 0 (the wave's k-th block is block
        // k * nq + sq = (sample, tile): divisions of wave-uniform numbers) and carried along -- later rounds unpack it
        // with shifts instead of dividing by the tile counts for every path
        uint32_t home, s, tile_x, tile_y;
        if (ROUND0) {
            const uint32_t blk = k * A.nq + sq;
            s = blk / A.n_tiles;
            const uint32_t tile = blk - s * A.n_tiles;
            tile_y = tile / A.tiles_x;
            tile_x = tile - tile_y * A.tiles_x;
            home = lane | (tile_x << 6) | (tile_y << A.home_sy) | (s << A.home_ss);
        } else {
            home = valid ? ldU(PU_HOME) : 0u;
            tile_x = (home >> 6) & A.home_mx; tile_y = (home >> A.home_sy) & A.home_my; s = home >> A.home_ss;
        }
        const uint32_t in = home & 63u;
        const uint32_t x = tile_x * 8u + (in & 7u);
        const uint32_t lyc = tile_y * 8u + (in >> 3);
        if (ROUND0 && !(x < A.W && lyc < A.rows)) valid = false;  // padding of a ragged tile: never alive
        const uint32_t ly = A.row0 + lyc;
        const uint32_t yy = A.row_begin + ly * A.row_stride;
        const uint32_t pixel = x + A.W * yy;  // main.rs:160-161,177

        // The analytic primitives of a query (core.rs:183-210): closest sphere / plane hit of the ray (org, dir), ties to the
        // lower object index.  Used when a query is made and -- in scenes with a handful of primitives (A.rehit) -- again when
        // its answer is taken: the same instructions on the same operands give the same bits, and 12 bytes per path and round
        // stay out of the streamed state (`count`: the work counters see the query once).
        auto analytic = [&](const V3 org, const V3 dir, const bool count, double& best, bool& found, uint32_t& b_obj, uint32_t& b_mat,
                            uint32_t& b_kp) __attribute__((always_inline)) {
            auto consider = [&](double t, uint32_t obj, uint32_t mt, uint32_t kp) __attribute__((always_inline)) {
                if (t <= kEps) return;  // core.rs:75-77
                if (t < best || (found && t == best && obj < b_obj)) {
                    best = t; found = true;
                    b_obj = obj; b_mat = mt; b_kp = kp;
                }
            };
            const double a = dot(dir, dir);
            if (MANY && A.sph_screen) {
                // Scenes with many spheres: a SCREEN in packed f32 (two spheres per instruction, operands from scalar loads) finds
                // the few spheres the ray can meet; only those run core.rs:183-195 in f64, each lane on its own survivors.  The
                // screen evaluates the same two quantities -- the half discriminant (d.oc)^2 - (d.d)(oc.oc - r^2) and d.oc -- and keeps
                // a sphere unless one of them is on the missing side by more than a bound on everything f32 lost:
                //   with L = max|o_i| + max|c_i|, D = max|d_i|, u = 2^-24:  |d.oc error| <= 24 u L D  (kept unless d.oc > 2^-17 L D: the
                //   near root is positive only for d.oc < 0), |half-discriminant error| <= u D^2 (300 L^2 + 24 r^2)  (kept unless it is
                //   below -2^-14 D^2 (L^2 + r_max^2)); DESIGN.md section 4.2 has the derivation.  Rays and scenes outside the range
                //   where f32 neither overflows nor underflows (`tame`) keep every sphere.  Survivors are considered in any order:
                //   `consider` breaks ties by object index itself.
                typedef float f2 __attribute__((ext_vector_type(2)));
                if (STATS && count) cnt.v[C_SPHERE] += S.n_spheres;
                const float ox = (float)org.x, oy = (float)org.y, oz = (float)org.z, dx = (float)dir.x, dy = (float)dir.y, dz = (float)dir.z;
                const float Lo = fmaxf(fabsf(ox), fmaxf(fabsf(oy), fabsf(oz))), Dm = fmaxf(fabsf(dx), fmaxf(fabsf(dy), fabsf(dz)));
                const float sc = Lo + S.sph_lc, sc2 = sc * sc + S.sph_r2max;
                const float md = 0x1p-14f * (Dm * Dm) * sc2, mb = 0x1p-17f * sc * Dm, af = (float)a;
                const bool tame = Lo < 1e8f && Dm < 1e8f && Dm > 1e-8f && sc2 > 1e-12f && sc2 < 1e17f && S.sph_screen_ok != 0u;
                const cptr<float> scr = as_const(S.sph_screen);
                for (uint32_t i0 = 0; i0 < S.n_spheres; i0 += 128u) {  // 128 spheres: four words of survivors, then ONE loop over them
                    uint32_t sv[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                    for (uint32_t w = 0; w < 4u; ++w) {
                        const uint32_t j0 = i0 + 32u * w;
                        if (j0 < S.n_spheres) {  // (wave-uniform)
                            uint32_t culled = 0;
                            const uint32_t n_in = min(32u, S.n_spheres - j0), pairs = (n_in + 1u) >> 1;
                            auto pair = [&](const uint32_t u) __attribute__((always_inline)) {
                                const cptr<float> t = scr + (size_t)(j0 + 2u * u) * 4u;  // (the table is padded to a multiple of 32 spheres, with zeros)
                                const f2 cx = {t[0], t[1]}, cy = {t[2], t[3]}, cz = {t[4], t[5]}, r2 = {t[6], t[7]};
                                const f2 ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
                                const f2 bh = __builtin_elementwise_fma(ocz, (f2)dz, __builtin_elementwise_fma(ocy, (f2)dy, ocx * dx));
                                const f2 cc = __builtin_elementwise_fma(ocz, ocz, __builtin_elementwise_fma(ocy, ocy, __builtin_elementwise_fma(ocx, ocx, -r2)));
                                const f2 tt = __builtin_elementwise_fma((f2)af, cc, (f2)(-md));
                                const f2 dd = __builtin_elementwise_fma(bh, bh, -tt);
                                const f2 v = __builtin_elementwise_min(dd, mb - bh);  // sign bit = out
                                culled = __builtin_amdgcn_alignbit(culled, __float_as_uint(v.x), 31);
                                culled = __builtin_amdgcn_alignbit(culled, __float_as_uint(v.y), 31);
                            };
                            if (pairs == 16u) {
#pragma unroll
                                for (uint32_t u = 0; u < 16u; ++u) pair(u);
                            } else {
#pragma unroll 2
                                for (uint32_t u = 0; u < pairs; ++u) pair(u);
                            }
                            culled <<= 32u - 2u * pairs;  // sphere j0 + k is bit 31 - k
                            sv[w] = (tame ? ~culled : ~0u) & (~0u << (32u - n_in));
                        }
                    }
                    while (__ballot((sv[0] | sv[1] | sv[2] | sv[3]) != 0u)) {
                        if ((sv[0] | sv[1] | sv[2] | sv[3]) != 0u) {
                            const uint32_t w = sv[0] ? 0u : sv[1] ? 1u : sv[2] ? 2u : 3u;
                            const uint32_t cur = sv[0] ? sv[0] : sv[1] ? sv[1] : sv[2] ? sv[2] : sv[3];
                            const uint32_t k = (uint32_t)__clz((int)cur), rest = cur & ~(0x80000000u >> k);
                            sv[0] = w == 0u ? rest : sv[0]; sv[1] = w == 1u ? rest : sv[1]; sv[2] = w == 2u ? rest : sv[2]; sv[3] = w == 3u ? rest : sv[3];
                            const uint32_t i = i0 + 32u * w + k;
                            double cx, cy, cz, r;
                            if (TABS) { const double* const sp = (const double*)((const char*)sh_tab + kShadeLdsBytes + kShadePackBytes) + i * 4u; cx = sp[0]; cy = sp[1]; cz = sp[2]; r = sp[3]; }
                            else { const double* const sp = S.spheres + (size_t)i * 4; cx = sp[0]; cy = sp[1]; cz = sp[2]; r = sp[3]; }
                            const V3 oc = org - mk(cx, cy, cz);
                            const double bb = 2.0 * dot(dir, oc);
                            const double c = dot(oc, oc) - r * r;
                            const double disc = bb * bb - 4.0 * a * c;
                            if (!(disc < 0.0)) {
                                const double t = (-bb - __builtin_sqrt(disc)) / (2.0 * a);
                                if (t > kEps && (t < best || (found && t == best))) {
                                    uint32_t mt, obj;
                                    prim_ids(i, mt, obj);
                                    if (t < best || obj < b_obj) { best = t; found = true; b_obj = obj; b_mat = mt; b_kp = i; }
                                }
                            }
                        }
                    }
                }
            } else
            // Solid::Sphere (core.rs:183-195), four at a time: the discriminants are evaluated together (their
            // operands are loaded together), then the few spheres the ray meets take the root, in scene order
            for (uint32_t i0 = 0; i0 < S.n_spheres; i0 += 4u) {
                double disc[4], bb[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t i = min(i0 + u, S.n_spheres - 1u);  // (a batch past the end repeats the last sphere, unused)
                    const double cx = sph_c[i * 4u], cy = sph_c[i * 4u + 1u], cz = sph_c[i * 4u + 2u], r = sph_c[i * 4u + 3u];
                    const V3 oc = org - mk(cx, cy, cz);
                    bb[u] = 2.0 * dot(dir, oc);
                    const double c = dot(oc, oc) - r * r;
                    disc[u] = bb[u] * bb[u] - 4.0 * a * c;
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; ++u) {
                    const uint32_t i = i0 + u;
                    if (i < S.n_spheres) {
                        if (STATS && count) cnt.v[C_SPHERE]++;
                        if (!(disc[u] < 0.0)) {
                            const double t = (-bb[u] - __builtin_sqrt(disc[u])) / (2.0 * a);
                            consider(t, sph_ids_c[i * 2 + 1], sph_ids_c[i * 2], i);
                        }
                    }
                }
            }
            for (uint32_t i = 0; i < S.n_planes; ++i) {
                if (STATS && count) cnt.v[C_PLANE]++;
                const cptr<double> pp = pl_c + (size_t)i * 4;
                const V3 pn = ld3(pp);
                const double dv = dot(pn, dir);
                if (fabs(dv) < kEps) continue;
                const V3 center = pn * pp[3];
                const double t = dot(center - org, pn) / dv;
                if (t < kEps) continue;
                consider(t, pl_ids_c[i * 2 + 1], pl_ids_c[i * 2], (1u << 30) | i);
            }
        };
        // what this round produces, and the state that travels on
        bool cont = false;  // the path stays alive
        bool go0 = false, go1 = false, go2 = false;
        V3 o = mk(0, 0, 0), dir0 = mk(0, 0, 1), nn = mk(0, 0, 1), col = mk(0, 0, 0), diff = mk(0, 0, 0);
        uint32_t hist = 0;  // (A.att_replay) bit k: vertex k was metal -- the attenuation is the product of the vertices' factors, in order
        double att = 1.0, ba0 = 0.0, eff0 = 0.0, eff1 = 0.0, held_t = 0.0;
        uint32_t fl = 0, bounce = 0, li = 0, n_sh = 0, mat = 0, held_tri = kNoId, held_m = kNoId;
        uint32_t h_kp = 0;  // slot 0's analytic hit (travels on while the model answer is pending)
        bool fresh0 = false;  // ba0 / h_kp / FB_FOUND0 were produced in this kernel (they are not read back in A.rehit mode)

        if (valid) {
            if (ROUND0) {
                double xo = 0.0, yo = 0.0;
                if (!(A.flags & RTM_FLAG_ZERO_JITTER)) {  // main.rs:162-163
                    uint32_t ra[4];
                    philox(A.seed, pixel, s, 0u, ra);
                    xo = u01(ra[0], ra[1]) * (0.5 - -0.5) + -0.5;
                    yo = u01(ra[2], ra[3]) * (0.5 - -0.5) + -0.5;
                }
                const double u = ((double)x + xo - (double)A.W * 0.5) / (double)A.W;   // main.rs:165
                const double v = ((double)yy + yo - (double)A.H * 0.5) / (double)A.H;  // main.rs:166
                dir0 = normalize(ld3(S.cam_fwd) * S.cam_zoom + ld3(S.cam_up) * v + ld3(S.cam_right) * u);  // camera.rs:47
                o = ld3(S.cam_pos);
                go0 = true;
                cont = true;
                if (STATS) cnt.v[C_CAM]++;
            } else {
                // ---- every load this round can need, issued before anything is decided: the kernel moves ~0.5 KB per
                //      path and is bound by memory latency, not by the few bytes a path in another state would skip
                fl = flb & ((1u << FB_BOUNCE_SHIFT) - 1u);
                bounce = flb >> FB_BOUNCE_SHIFT;
                o = ldD3(PF_O);
                if (!A.rehit) { ba0 = ldD(PF_BA0); h_kp = ldU(PU_HKP); }
                mat = ldU(PU_MAT);
                if (A.att_replay) { hist = mat >> 16; mat &= 0xFFFFu; }
                if (nL > 2u) { held_t = ldD(PF_HELD); held_tri = ldU(PU_HTRI); held_m = ldU(PU_HM); li = ldU(PU_LI); }  // (only then can a result be held)
                eff0 = ldD(PF_EFF); eff1 = ldD(PF_EFF + 1);
                if (nL > 2u) nn = ldD3(PF_NRM);
                dir0 = ldD3(PF_D0);
                col = ldD3(PF_COL);
                if (!A.att_replay) att = ldD(PF_ATT);
                if (nL > 2u) diff = ldD3(PF_DIFF);
            }
        }
#ifdef RTM_SHADE_TIMES
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st_b = __builtin_amdgcn_s_memtime();
#endif
        // ---- the answers of the previous round: the next entry of region j for every path whose flag says it asked
        //      (a held answer of slot 0 was read in the round after its ray was traced; its flag stays set)
        FrontResult r0, l_r1, l_r2;
        r0.t = 0.0; r0.tri = kNoId; r0.model = kNoId;
        l_r1 = r0; l_r2 = r0;
        if (!ROUND0) {
            const bool q0 = valid && (flb & FB_TQ0) && !(flb & FB_HELD), q1 = valid && (flb & FB_TQ1), q2 = valid && (flb & FB_TQ2);
            const unsigned long long m0 = __ballot(q0), m1 = __ballot(q1), m2 = __ballot(q2), below = (1ull << lane) - 1ull;
            const size_t rbase = (size_t)sq * A.cap_i;
            if (q0) r0 = A.res[rbase + n_res0 + (uint32_t)__popcll(m0 & below)];
            if (q1) l_r1 = A.res[rbase + A.cap_l + n_res1 + (uint32_t)__popcll(m1 & below)];
            if (q2) l_r2 = A.res[rbase + 2u * A.cap_l + n_res2 + (uint32_t)__popcll(m2 & below)];
            n_res0 += (uint32_t)__popcll(m0); n_res1 += (uint32_t)__popcll(m1); n_res2 += (uint32_t)__popcll(m2);
        }
#ifdef RTM_SHADE_TIMES
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st_c = __builtin_amdgcn_s_memtime();
#endif
        // A path advances IN THE LANE for as long as every answer it waits for is known without the trace kernel: a vertex whose
        // bounce ray and shadow rays all miss the models' boxes is consumed in the next pass of this loop (its queries were
        // settled by the analytic primitives when they were made) instead of travelling through memory to the next round.
        // A lane leaves the loop in the pass that appends a trace item for it, or when its sample ends; scenes without models
        // finish whole paths in round 0.  (What is combined, and in which order, is unchanged: only where the state waits.)
        bool active = valid;
#ifdef RTM_SHADE_TIMES
        unsigned long long st_d = 0;
#endif
        for (uint32_t pass = 0;; ++pass) {  // (wave-uniform)
        uint32_t h_obj = kNoId, h_mat = 0, h_tri = kNoId;  // slot 0's hit
        double best0 = 0.0;
        bool found0 = false;
        if (pass > 0u) { cont = false; go0 = false; }
        go1 = false; go2 = false;
        if (active && (!ROUND0 || pass > 0u)) {
            {
                if (fl & FB_HELD) { r0.t = held_t; r0.tri = held_tri; r0.model = held_m; }  // slot 0's answer came in an earlier round
                if (A.att_replay) {  // main.rs:118,124: the same multiplications in the same order as when the vertices were finished
                    att = 1.0;
                    for (uint32_t kb = 0; kb < bounce; ++kb) att = att * (((hist >> kb) & 1u) ? 0.8 : 0.1);
                }
                // ---- 0. decisions
                const bool a_v = (fl & FB_HASV) != 0u;
                const uint32_t n_in = a_v ? ((fl >> FB_NSH_SHIFT) & 3u) : 0u;  // shadow results to consume
                const bool a_more = a_v && li + n_in < nL;                      // further lights: another batch
                const bool a_fin = a_v && !a_more;                              // the vertex is complete (main.rs:107-125)
                const bool a_end = a_fin && (int)bounce + 1 >= K;               // ... and it was the last one allowed
                const bool b_take = !a_more && !a_end;                          // the path query's result is due
                if (a_more && (fl & FB_TQ0)) {  // keep slot 0's answer: the trace kernel reuses the result slots next round
                    held_t = r0.t; held_tri = r0.tri; held_m = r0.model;
                    fl |= FB_HELD;
                }
                // ---- 1. the path query: analytic hit, then the model hit the trace kernel found (core.rs:75-83)
                if (b_take) {
                    if (!MANY && A.rehit && !fresh0) {  // the analytic part of the query again, from the ray the state holds (A.rehit: at most four primitives)
                        double rb = __builtin_inf();
                        bool rf = false;
                        uint32_t ro = kNoId, rm = 0u, rk = 0u;
                        analytic(o, dir0, false, rb, rf, ro, rm, rk);
                        best0 = rb; found0 = rf; h_kp = rk;
                        if (rf) { h_mat = rm; h_obj = ro; }
                    } else {
                        best0 = ba0;
                        found0 = (fl & FB_FOUND0) != 0u;
                        if (found0) prim_ids(h_kp, h_mat, h_obj);
                    }
                    if ((fl & FB_TQ0) && r0.model != kNoId) {
                        const uint32_t m = r0.model;
                        const double t = r0.t;
                        const uint32_t mobj = model_obj(m);
                        if (!(t <= kEps) && (t < best0 || (found0 && t == best0 && mobj < h_obj))) {
                            best0 = t; found0 = true;
                            h_obj = mobj; h_mat = model_mat(m); h_kp = (2u << 30) | m; h_tri = r0.tri;
                        }
                    }
                }
                const bool b_hit = b_take && found0, b_miss = b_take && !found0;
                const bool end_sample = a_end || b_miss;
                const uint32_t mat_old = a_fin ? mat : 0u;
                const bool metal_old = mat_at(mat_old, 3) != 0.0;
                const uint32_t mat_new = b_hit ? h_mat : mat_old;
                const bool metal_new = mat_at(mat_new, 3) != 0.0;
                const uint32_t bounce_new = bounce + (a_fin ? 1u : 0u);  // index of the vertex a hit creates
                const bool plastic_hit = b_hit && !metal_new;
                uint32_t ra[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
                if (plastic_hit) {  // Vec3::random() of this bounce (vec3.rs:160-162)
                    philox(A.seed, pixel, s, 1u + 2u * bounce_new, ra);
                    philox(A.seed, pixel, s, 2u + 2u * bounce_new, rb);
                }
                // ---- 2. shadow results of the pending vertex, in light order (main.rs:80-104); o == P.  The ray towards
                //      the light and what it would add are recomputed from (P, n, light) -- the same expressions that
                //      produced the ray -- rather than carried through memory.
                if (n_in > 0u) {
                    if (li == 0u) diff = mk(0, 0, 0);
                    // main.rs:86-93 takes the closest hit of the shadow ray and asks whether it lies before the light:
                    // sqdist(ray_at(o, dir, t), o) <= dl2.  That predicate can only go from true to false as t grows (every
                    // operation in it is monotonic in t for t > 0, rounding included), so "the closest hit occludes" equals
                    // "some hit occludes": the spheres and planes were asked when the ray was made (issue(): FB_C is clear if
                    // one of them occludes), the models' closest hit is asked here.
                    auto shade = [&](const uint32_t j, const double eff, const uint32_t rm, const double rt) __attribute__((always_inline)) {
                        if (j < n_in && (fl & (FB_C1 << j))) {  // a query that cannot contribute is not looked at
                            const V3 lp = light3(li + j, 0);
                            const V3 dir = normalize(lp - o);
                            const double dl2 = sqdist(o, lp);
                            bool occluded = false;
                            if ((fl & (FB_TQ1 << j)) && rm != kNoId && !(rt <= kEps)) occluded = sqdist(ray_at(o, dir, rt), o) <= dl2;
                            if (!occluded) {  // eff > 0.001, or FB_C would not be set
                                const double lr = light_at(li + j, 6);
                                const double dd = fmax(dl2 / (lr * lr), 1.0);
                                const double decay = 1.0 / dd;
                                diff = diff + light3(li + j, 3) * decay * eff;
                            }
                        }
                    };
                    shade(0u, eff0, l_r1.model, l_r1.t);
                    shade(1u, eff1, l_r2.model, l_r2.t);
                    li += n_in;
                } else if (a_fin) {
                    diff = mk(0, 0, 0);  // a vertex without lights
                }
                // ---- 3. end of the light loop: the surface's colour, attenuation (main.rs:107-125)
                if (a_fin) {
                    col = col + mk(mat_at(mat_old, 0), mat_at(mat_old, 1), mat_at(mat_old, 2)) * att * (diff * (metal_old ? 0.2 : 0.9));
                    att = att * (metal_old ? 0.8 : 0.1);
                    hist |= (metal_old ? 1u : 0u) << (bounce & 15u);
                    ++bounce;
                    fl &= ~FB_HASV;
                }
                // ---- 4. the path query's result
                if (b_take && bounce == 0u && s == 0u) {
                    const size_t oi = (size_t)ly * A.W + x;
                    if (out_obj) out_obj[oi] = found0 ? h_obj : kNoId;
                    if (out_tri) out_tri[oi] = found0 ? h_tri : kNoId;
                }
                if (b_miss) col = col + mk(1, 1, 1) * att;  // main.rs:127
                if (end_sample) {  // the sample's colour, where the resolve kernel looks for it
                    const size_t fi = (size_t)((s * A.n_tiles + tile_y * A.tiles_x + tile_x) * 64u + in);
                    A.fin[at64(3, 0, fi)] = col.x; A.fin[at64(3, 1, fi)] = col.y; A.fin[at64(3, 2, fi)] = col.z;
                }
                bool norm0 = false;
                if (b_hit) {
                    // surface point and shading normal (core.rs:89,169-177,195,209; main.rs:68-73)
                    const V3 d = dir0;
                    const V3 P = ray_at(o, d, best0);
                    const uint32_t kind = h_kp >> 30, prim = h_kp & 0x3FFFFFFFu;
                    V3 raw;
                    if (kind == 0u) raw = ray_at(o, d, best0) - ld3(S.spheres + (size_t)prim * 4);
                    else if (kind == 1u) raw = ld3(S.planes + (size_t)prim * 4);  // used as written, not normalised
                    else raw = model_normal_raw(S.models[prim], h_tri, P);
                    nn = normalize(raw);
                    if (kind == 1u) nn = raw;
                    if (dot(d, nn) > 0.001) nn = nn * -1.0;
                    // bounce direction (main.rs:50-55): metal reflects (not renormalised, vec3.rs:172-175); plastic is
                    // normalize(n + random())
                    const V3 rnd = mk(u01(ra[0], ra[1]) * (1.0 - 0.0) + 0.0, u01(ra[2], ra[3]) * (1.0 - 0.0) + 0.0,
                                      u01(rb[0], rb[1]) * (1.0 - 0.0) + 0.0);
                    const V3 refl = d - nn * 2.0 * dot(d, nn);
                    const V3 dif = nn + rnd;
                    dir0 = sel3(metal_new, refl, dif);
                    norm0 = !metal_new;
                    mat = mat_new;
                    o = P;  // the shadow rays start here and so does the bounce (main.rs:82,121)
                    li = 0;
                    fl |= FB_HASV;
                }
                // ---- 5. the new queries
                go0 = b_hit && (int)bounce + 1 < K;
                const bool sh = b_hit || a_more;
                n_sh = sh ? (nL - li < 2u ? nL - li : 2u) : 0u;
                go1 = n_sh > 0u;
                go2 = n_sh > 1u;
                {
                    const V3 nd = normalize(dir0);  // vec3.rs:213-215: three divisions by the length
                    dir0 = sel3(norm0, nd, dir0);
                }
                if (STATS && go0) cnt.v[C_BOUNCE]++;
                if (STATS) cnt.v[C_SHADOW] += n_sh;
                cont = !end_sample;
                // flags that describe the queries are rebuilt below; slot 0's stay while its result is still pending
                fl &= a_more ? (FB_HASV | FB_FOUND0 | FB_TQ0 | FB_HELD) : FB_HASV;
            }
        }
        // ---- a query: analytic primitives (core.rs:183-210), slab test and DDA set-up of the first model whose box
        //      the ray meets (box3D.rs:64-93, core.rs:95-125); that model and the following ones are the trace kernel's
        // Shadow queries (j > 0) bring dl2 = |light - o|^2 and `lit`: a sphere or plane that occludes the light settles the
        // query here (its flag FB_C stays clear and no ray goes to the models, unless exact work counters are wanted).
        // The item's path slot (IU_P) is written once the lane knows its next slot, below.
        uint32_t it_of0 = 0, it_of1 = 0, it_of2 = 0;  // (item indices fit 32 bits: the trace kernel keeps them in a word, too)
        bool item0 = false, item1 = false, item2 = false;
        struct AHit { double best; uint32_t found, obj, kp; };  // closest sphere / plane of a query
        auto issue = [&](const uint32_t j, const bool go, const V3 dir, const double dl2, const bool lit, const AHit* pre) __attribute__((always_inline)) {
            bool item = false;
            int ix = 0, iy = 0, iz = 0;
            double tmx = 0, tmy = 0, tmz = 0;
            uint32_t mword = 0;
            if (go) {
                double best = __builtin_inf();
                bool found = false;
                uint32_t b_obj = kNoId, b_mat = 0, b_kp = 0;
                if (pre) { best = pre->best; found = pre->found != 0u; b_obj = pre->obj; b_kp = pre->kp; }  // (MANY: answered in the packed pass)
                else analytic(o, dir, true, best, found, b_obj, b_mat, b_kp);
                bool models = true;
                if (j == 0u) {
                    ba0 = best;
                    h_kp = b_kp;
                    fresh0 = true;
                    if (found) fl |= FB_FOUND0;
                } else {
                    const bool occ = found && sqdist(ray_at(o, dir, best), o) <= dl2;
                    if (lit && !occ) fl |= FB_C1 << (j - 1u);
                    models = !occ || !A.cull;
                }
                (void)b_mat;
                if (models) {
                    for (uint32_t m = 0; m < S.n_models && !item; ++m) {
                        const __attribute__((address_space(4))) RtmDevModel& M = models_c[m];
                        if (STATS) cnt.v[C_BBOX]++;
                        const double fx = 1.0 / dir.x, fy = 1.0 / dir.y, fz = 1.0 / dir.z;
                        const double t1 = (M.bmin[0] - o.x) * fx, t2 = (M.bmax[0] - o.x) * fx;
                        const double t3 = (M.bmin[1] - o.y) * fy, t4 = (M.bmax[1] - o.y) * fy;
                        const double t5 = (M.bmin[2] - o.z) * fz, t6 = (M.bmax[2] - o.z) * fz;
                        const double tmin = fmax(fmax(fmin(t1, t2), fmin(t3, t4)), fmin(t5, t6));
                        const double tmax = fmin(fmin(fmax(t1, t2), fmax(t3, t4)), fmax(t5, t6));
                        if (!(tmax < 0.0 || tmin > tmax)) {
                            const V3 P0 = ray_at(o, dir, tmin);
                            const int n = (int)M.n;
                            ix = cell_coord((P0.x - M.bmin[0]) / M.cell[0], n);
                            iy = cell_coord((P0.y - M.bmin[1]) / M.cell[1], n);
                            iz = cell_coord((P0.z - M.bmin[2]) / M.cell[2], n);
                            const int sx = dir.x > 0.0 ? 1 : -1, sy = dir.y > 0.0 ? 1 : -1, sz = dir.z > 0.0 ? 1 : -1;
                            const double nbx = (M.bmin[0] + M.hc2[0] * ((double)(ix + (sx > 0 ? 1 : 0)) + 0.5)) - M.hc[0];
                            const double nby = (M.bmin[1] + M.hc2[1] * ((double)(iy + (sy > 0 ? 1 : 0)) + 0.5)) - M.hc[1];
                            const double nbz = (M.bmin[2] + M.hc2[2] * ((double)(iz + (sz > 0 ? 1 : 0)) + 0.5)) - M.hc[2];
                            tmx = (nbx - P0.x) / dir.x; tmy = (nby - P0.y) / dir.y; tmz = (nbz - P0.z) / dir.z;
                            mword = m | (sx > 0 ? IM_SX : 0u) | (sy > 0 ? IM_SY : 0u) | (sz > 0 ? IM_SZ : 0u);
                            item = true;
                        }
                    }
                    if (item) fl |= FB_TQ0 << j;
                }
            }
            // append to the wave's trace sub-queue
            const unsigned long long want = __ballot(item);
            if (want) {
                uint32_t& n_items_j = j == 0u ? n_items0 : (j == 1u ? n_items1 : n_items2);
                const uint32_t at = j * A.cap_l + n_items_j;  // region j of the wave's sub-queue
                n_items_j += (uint32_t)__popcll(want);
                if (item) {
                    const size_t it = (size_t)sq * A.cap_i + at + (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                    double* const I = A.itD;
                    if (j != 0u) { I[at64(IT_DCOUNT, IT_D + 0, it)] = dir.x; I[at64(IT_DCOUNT, IT_D + 1, it)] = dir.y; I[at64(IT_DCOUNT, IT_D + 2, it)] = dir.z; }
                    I[at64(IT_DCOUNT, IT_TM + 0, it)] = tmx; I[at64(IT_DCOUNT, IT_TM + 1, it)] = tmy; I[at64(IT_DCOUNT, IT_TM + 2, it)] = tmz;
                    A.itU[at64(IU_COUNT, IU_CELL, it)] = (uint32_t)(ix + 1) | ((uint32_t)(iy + 1) << 10) | ((uint32_t)(iz + 1) << 20);
                    A.itU[at64(IU_COUNT, IU_M, it)] = mword;
                    if (j == 0u) { it_of0 = (uint32_t)it; item0 = true; } else if (j == 1u) { it_of1 = (uint32_t)it; item1 = true; } else { it_of2 = (uint32_t)it; item2 = true; }
                }
            }
        };
        {
            // shadow rays to lights li, li+1 (main.rs:80-85) and what each would add to the diffuse sum (main.rs:96-102)
            V3 sdir[2] = {mk(0, 0, 1), mk(0, 0, 1)};
            bool strace[2] = {false, false}, slit[2] = {false, false};
            double sdl2[2] = {0.0, 0.0};
            auto shadow_ray = [&](const uint32_t j, const bool go) __attribute__((always_inline)) {
                if (go) {
                    const V3 lp = light3(li + j, 0);
                    sdir[j] = normalize(lp - o);
                    sdl2[j] = sqdist(o, lp);
                    const double eff = dot(sdir[j], nn);
                    if (j == 0u) eff0 = eff; else eff1 = eff;
                    slit[j] = eff > 0.001;  // main.rs:96: otherwise the query's answer is never used
                    strace[j] = slit[j] || !A.cull;
                    if (STATS && !strace[j]) n_culled++;
                }
            };
            shadow_ray(0u, active && go1);
            shadow_ray(1u, active && go2);
            const bool g0 = active && go0, g1 = active && go1 && strace[0], g2 = active && go2 && strace[1];  // (the host turns culling off when exact work counters are wanted)
            if (MANY) {
                // The queries of the wave as ONE list, 64 per pass over the spheres: those of slot 0 in lane order, then slot 1,
                // then slot 2.  (Slot by slot, each pass runs with the lanes that have a query in THAT slot: shadow queries that
                // face away from their light are culled, paths end, and the passes are half empty.)  The lane that holds query q
                // of a pass pulls the ray from the lane that made it (register shuffles), runs the analytic primitives on it, and
                // the makers pull the answers back: the same query with the same operands, evaluated in another lane.
                const unsigned long long m0 = __ballot(g0), m1 = __ballot(g1), m2 = __ballot(g2), below = (1ull << lane) - 1ull;
                const uint32_t n0 = (uint32_t)__popcll(m0), n01 = n0 + (uint32_t)__popcll(m1), n012 = n01 + (uint32_t)__popcll(m2);
                uint32_t* const tab = (uint32_t*)((char*)sh_tab + (TABS ? kShadeLdsBytes : 0)) + (threadIdx.x >> 6) * 192u;  // [192] per wave
                const uint32_t q0 = (uint32_t)__popcll(m0 & below), q1 = n0 + (uint32_t)__popcll(m1 & below), q2 = n01 + (uint32_t)__popcll(m2 & below);
                if (g0) tab[q0] = lane;
                if (g1) tab[q1] = lane | 64u;
                if (g2) tab[q2] = lane | 128u;
                __builtin_amdgcn_wave_barrier();
                AHit h0{__builtin_inf(), 0u, kNoId, 0u}, h1 = h0, h2 = h0;
                for (uint32_t base = 0; base < n012; base += 64u) {  // (wave-uniform)
                    const bool have = base + lane < n012;
                    const uint32_t e = have ? tab[base + lane] : 0u;
                    const int src = (int)(e & 63u);
                    const uint32_t slot = e >> 6;
                    const V3 po = mk(__shfl(o.x, src, 64), __shfl(o.y, src, 64), __shfl(o.z, src, 64));
                    const V3 p0 = mk(__shfl(dir0.x, src, 64), __shfl(dir0.y, src, 64), __shfl(dir0.z, src, 64));
                    const V3 p1 = mk(__shfl(sdir[0].x, src, 64), __shfl(sdir[0].y, src, 64), __shfl(sdir[0].z, src, 64));
                    const V3 p2 = mk(__shfl(sdir[1].x, src, 64), __shfl(sdir[1].y, src, 64), __shfl(sdir[1].z, src, 64));
                    const V3 pd = sel3(slot == 0u, p0, sel3(slot == 1u, p1, p2));
                    double best = __builtin_inf();
                    bool found = false;
                    uint32_t b_obj = kNoId, b_mat = 0, b_kp = 0;
                    if (have) analytic(po, pd, true, best, found, b_obj, b_mat, b_kp);
                    (void)b_mat;
                    auto fetch = [&](const bool g, const uint32_t q, AHit& h) __attribute__((always_inline)) {
                        const bool here = g && q >= base && q < base + 64u;  // this lane's query of the slot was answered in this pass
                        const int from = here ? (int)(q - base) : 0;
                        const double fb = __shfl(best, from, 64);
                        const uint32_t ff = (uint32_t)__shfl((int)(found ? 1u : 0u), from, 64);
                        const uint32_t fo = (uint32_t)__shfl((int)b_obj, from, 64), fk = (uint32_t)__shfl((int)b_kp, from, 64);
                        if (here) { h.best = fb; h.found = ff; h.obj = fo; h.kp = fk; }
                    };
                    fetch(g0, q0, h0); fetch(g1, q1, h1); fetch(g2, q2, h2);
                }
                __builtin_amdgcn_wave_barrier();
                issue(0u, g0, dir0, 0.0, false, &h0);
                issue(1u, g1, sdir[0], sdl2[0], slit[0], &h1);
                issue(2u, g2, sdir[1], sdl2[1], slit[1], &h2);
            } else {
                issue(0u, g0, dir0, 0.0, false, nullptr);
                issue(1u, g1, sdir[0], sdl2[0], slit[0], nullptr);
                issue(2u, g2, sdir[1], sdl2[1], slit[1], nullptr);
            }
        }
        fl = (fl & ~(3u << FB_NSH_SHIFT)) | (n_sh << FB_NSH_SHIFT);
        // ---- who waits for the trace kernel, who goes round again in the lane
        const bool pending = (fl & (FB_TQ1 | FB_TQ2)) != 0u || ((fl & FB_TQ0) != 0u && !(fl & FB_HELD));
        // (... provided enough lanes of the wave do: a pass with a few lanes costs what a full one costs, and the paths that go
        // through memory instead come back compacted)
        const bool could = active && cont && !pending;
        // (a vertex that still has lights to go -- scenes with more than two -- always goes on in the lane when it can: every path of
        // the wave that shaded a vertex this pass is in that state, so the next pass is as full as this one)
        const bool more_lights = (fl & FB_HASV) != 0u && li + n_sh < nL;
        const bool again = could && ((more_lights && A.inlane_lights != 0u) || (uint32_t)__popcll(__ballot(could)) >= A.inlane);
        const bool keep = active && cont && !again;
        // ---- the slot the path moves to (dense, in the other buffer)
        const unsigned long long alive = __ballot(keep);
        const uint32_t p_out = sq * A.cap_l + n_live_local + (uint32_t)__popcll(alive & ((1ull << lane) - 1ull));
        n_live_local += (uint32_t)__popcll(alive);
        auto stD = [&](int f, double v) { Dout[at64((int)A.pf_n, f, p_out)] = v; };
        auto stD3 = [&](int f, V3 v) { stD(f, v.x); stD(f + 1, v.y); stD(f + 2, v.z); };
        auto stU = [&](int f, uint32_t v) { Uout[at64((int)A.pu_n, f, p_out)] = v; };
#ifdef RTM_SHADE_TIMES
        st_d = __builtin_amdgcn_s_memtime();
#endif
        if (keep) {  // the state moves on
            // the path's next slot: where the trace kernel finds the origin of its rays
            if (item0) A.itU[at64(IU_COUNT, IU_P, it_of0)] = p_out;
            if (item1) A.itU[at64(IU_COUNT, IU_P, it_of1)] = p_out;
            if (item2) A.itU[at64(IU_COUNT, IU_P, it_of2)] = p_out;
            stD3(PF_COL, col);
            if (!A.att_replay) stD(PF_ATT, att);
            stD3(PF_O, o); stD3(PF_D0, dir0);
            if (!A.rehit) stD(PF_BA0, ba0);
            stD(PF_EFF, eff0); stD(PF_EFF + 1, eff1);
            if (nL > 2u) { stD3(PF_NRM, nn); stD3(PF_DIFF, diff); stD(PF_HELD, held_t); stU(PU_HTRI, held_tri); stU(PU_HM, held_m); stU(PU_LI, li); }
            stU(PU_FLB, fl | (bounce << FB_BOUNCE_SHIFT));
            if (!A.rehit) stU(PU_HKP, h_kp);
            stU(PU_MAT, A.att_replay ? (mat | (hist << 16)) : mat);
            stU(PU_HOME, home);
        }
        active = again;
        if (!__ballot(active)) break;
        r0.t = 0.0; r0.tri = kNoId; r0.model = kNoId;  // later passes: nothing was traced for this lane
        l_r1 = r0; l_r2 = r0;
        }
#ifdef RTM_SHADE_TIMES
        {
            const unsigned long long st_e = __builtin_amdgcn_s_memtime();
            st_t[0] += st_b - st_a; st_t[1] += st_c - st_b; st_t[2] += st_d - st_c; st_t[3] += st_e - st_d; st_t[4] += 1;
        }
#endif
    }
#ifdef RTM_SHADE_TIMES
    if (lane == 0 && out_cnt)
        for (int i = 0; i < 5; ++i) atomicAdd(out_cnt + 34 + i, st_t[i]);
#endif
    if (lane == 0) {
        my_ctr[(round + 1u) & 1u] = n_live_local;
        my_ctr[CT_ITEMS0] = n_items0; my_ctr[CT_ITEMS0 + 1] = n_items1; my_ctr[CT_ITEMS0 + 2] = n_items2;
        my_ctr[CT_CLAIM] = 0u;
    }
    flush_counters<STATS>(cnt, out_cnt);
    if (STATS) {
        const uint64_t v = wave_sum(n_culled);
        if (lane == 0 && v) atomicAdd(out_cnt + 33, (unsigned long long)v);
    }
}

// ================================================================================================
// trace: persistent lanes; a lane holds one ray and only ever walks cells or tests triangles
enum : uint32_t { TS_IDLE = 0, TS_ENTER = 1, TS_TRAV = 2 };

// ALL_LDS: every model's occupancy bitmap fits the LDS budget (the host checks)
// HAS_LONG: some cell of some model has more than 32 entries (the box screen exists only in that instantiation)
template <bool STATS, bool ALL_LDS, bool HAS_LONG = false>
__global__ __launch_bounds__(kBlock, RTM_FRONT_OCC) void front_trace_kernel(RtmDevScene S, FrontArgs A, uint32_t round, unsigned long long* __restrict__ out_cnt) {
    extern __shared__ uint32_t occ_lds[];  // [A.occ_words]
    __shared__ uint32_t mtab[kModelTab];     // per model: cells per side | first scene cell / 32 << 10
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t m = tid; m < min(S.n_models, (uint32_t)kModelTab); m += kBlock)
        mtab[m] = S.models[m].n | ((S.models[m].cell_base >> 5) << 10);
    {  // stage the scene-wide bitmaps (the host passes occ_words = all of them, the plain occupancy alone, or 0 when neither fits)
        const gptr<uint32_t> src = as_global(S.g_occ7);
        for (uint32_t i = tid; i < A.occ_words; i += kBlock) occ_lds[i] = src[A.occ_plain ? i * 8u + 6u : i];
    }
    __syncthreads();
    Counters cnt;
    if (STATS)
        for (int i = 0; i < 9; ++i) cnt.v[i] = 0;
    const uint64_t dg_begin = STATS ? __builtin_amdgcn_s_memtime() : 0;
    uint64_t dg_tr = 0, dg_tw = 0, dg_tt = 0, dg_t0 = 0;  // cycles in refill / walk / test (STATS)
    // The wave's position in the queue.  Sub-queues sq, sq + waves, sq + 2 waves, ... are its own; their items are
    // claimed A.claim at a time from the sub-queue's cursor (one atomic per claim, by lane 0), so that a wave whose own
    // sub-queues are empty can take runs from up to A.steal sub-queues of other waves: rays differ in cost by an order
    // of magnitude and the launch lasts as long as its slowest wave.  Every wave stops after a bounded number of
    // failed claims; nothing waits on another wave.
    const uint32_t trace_waves = gridDim.x * (kBlock / 64);
    const uint32_t wave_id = blockIdx.x * (kBlock / 64) + (tid >> 6);
    uint32_t sq = wave_id, sq_items = 0, sq_n0 = 0, sq_n01 = 0, stolen = 0;  // items of sub-queue sq: all regions, region 0, regions 0 and 1
    bool stealing = false;
    uint32_t pool_next = 0, pool_end = 0;  // unread items of the current claim, counted through the sub-queue's three regions
    auto open_queue = [&]() __attribute__((always_inline)) {  // (wave-uniform)
        const uint32_t* const c = A.ctr + (size_t)sq * CT_STRIDE + CT_ITEMS0;
        sq_n0 = c[0]; sq_n01 = sq_n0 + c[1]; sq_items = sq_n01 + c[2];
    };
    if (sq < A.nq) open_queue();
    uint32_t steal_max = sq_items >= A.claim ? A.steal : 0u;  // (short queues, late rounds: nothing worth taking)
    const gptr<double> I = as_global(A.itD);
    const gptr<uint32_t> IU = as_global(A.itU);
    const gptr<double> Pst = as_global(A.D[(round + 1u) & 1u]);  // the path state the shade kernel has just written: the rays' origins

    uint32_t st = TS_IDLE, model_i = 0, q = 0;
    const gptr<RtmDevModel> models_g = (gptr<RtmDevModel>)S.models;
    const gptr<uint32_t> cell_info = as_global(S.g_cell_info);
    const gptr<double> cell_rec = as_global(S.g_cell_rec);
    const gptr<double> cell_frame = as_global(S.g_cell_frame);
    const gptr<uint32_t> cell_box8 = as_global(S.g_cell_box8);
    const gptr<uint32_t> occ_glb = as_global(S.g_occ7);
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    double tmx = 0, tmy = 0, tmz = 0, tdx = 0, tdy = 0, tdz = 0;
    // Where the ray is, in the form the walk needs (nothing is multiplied per step):
    //   cv  = scene cell << 3 | arrival code (0..5 = the step came along -x,+x,-y,+y,-z,+z; 6 = entry cell, no step yet);
    //   rem = steps left before the ray leaves the grid, per axis (x | y << 10 | z << 20), with a guard bit above each 9-bit
    //         field: a step along an axis that has none left borrows its guard, which is the bounds test of core.rs:129-134;
    //   wx, wy, wz = what a step along each axis adds to cv: (cell index stride, signed) << 3 | arrival code.
    uint32_t cv = 6u, rem = kRemGuards, wx = 0, wy = 0, wz = 0;
    uint32_t k = 0, k_end = 0, m_tri = 0;  // [k, k_end): the rest of the current cell's list, k = first entry of the 32-entry window
    uint32_t wmask = 0;                    // untested entries of the window
    uint32_t wmask_hi = 0;                 // long cells: the 32 entries behind the window, already screened (F_HI)
    double m_t = 0.0;
    // per-ray flags, kept as bits of one VGPR: as separate bools they live in SGPR lane masks and every join of the
    // divergent control flow below costs a handful of scalar mask merges per flag (half of the kernel's instructions)
    enum : uint32_t {
        F_HAVE = 1u,   // a triangle of the current cell was accepted (core.rs:140-149)
        F_OPEN = 4u,   // the current cell's list is not exhausted
        F_WINIT = 8u,  // the window of a newly opened cell has not been set up yet
        F_LONG = 16u,  // the window belongs to a long cell and has not been screened against the entries' boxes yet
        F_HI = 32u     // the box screen also covered the 32 entries after the window: their mask waits in wmask_hi
    };
    uint32_t fl = 0;
    double bt = 0.0;  // closest model hit of this ray so far, over the models visited (core.rs:79-83)
    uint32_t bm = kNoId, btri = kNoId;
    bool exhausted = sq >= A.nq;
#ifdef RTM_WAVE_TIMES
    const unsigned long long wt_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long wt_exh = 0, wt_rays = 0;
#endif
    uint64_t dg_steps = 0, dg_step_lanes = 0, dg_walks = 0, dg_walk_lanes = 0, dg_slots = 0, dg_slot_lanes = 0, dg_refills = 0, dg_refill_lanes = 0;  // STATS only, wave-uniform

    uint64_t dg_idle = 0, dg_other = 0, dg_wait_t = 0, dg_wait_w = 0;  // STATS: lanes idle / in another state or model at a step; testers waiting in walk steps, walkers in test steps
    uint32_t dg_ray_its = 0, dg_max_walks = 0, dg_max_tests = 0;  // STATS: walk iterations (low half) and test steps of this lane's ray; the lane's maxima
    auto finish_ray = [&]() __attribute__((always_inline)) {
        if (STATS) { dg_max_walks = max(dg_max_walks, dg_ray_its & 0xFFFFu); dg_max_tests = max(dg_max_tests, dg_ray_its >> 16); }
        FrontResult r;
        r.t = bt; r.tri = btri; r.model = bm;
        A.res[q] = r;  // q = the item's index
        st = TS_IDLE;
    };
    auto model_result = [&]() __attribute__((always_inline)) {  // the lane's model reports a hit (core.rs:75-83 over the models)
        if (!(m_t <= kEps)) {
            bool take = bm == kNoId || m_t < bt;
            // equal t in two models: the object earlier in the scene file keeps it (core.rs:79) -- looked up only then
            if (!take && m_t == bt) take = models_g[model_i].obj < models_g[bm].obj;
            if (take) { bt = m_t; bm = model_i; btri = m_tri; }
        }
    };
    auto leave_model = [&]() __attribute__((always_inline)) {  // on to the next model, or the ray is done
        model_i += 1u;
        if (model_i >= S.n_models) finish_ray(); else st = TS_ENTER;
    };
    // The bitmap probe of the lane's cell (the bitmap of its arrival direction: cells whose entries were all tested one
    // step ago count as empty; the counting build, and scenes whose bitmaps do not fit in LDS together, use the plain
    // occupancy -- the host encodes which in occ_shift / occ_m1 / occ_m2).
    // (ALL_LDS is a kernel parameter, not a branch: if the LDS probe and the HBM probe met in one block, the wait before
    // the bit test would also wait for outstanding global loads.)  The cell's record is read by the next test phase, not
    // here: a load into registers that an earlier step's load may still be filling makes the compiler wait for all of them.
    auto probe = [&]() __attribute__((always_inline)) {
        if (STATS) cnt.v[C_CELLS]++;
        const uint32_t wi = ((cv >> 8) << A.occ_shift) + ((cv & A.occ_m1) | A.occ_m2);
        uint32_t word;
        if (ALL_LDS) word = occ_lds[wi];
        else word = occ_glb[wi];
        if ((word >> ((cv >> 3) & 31u)) & 1u) fl |= F_WINIT | F_OPEN;  // non-empty
    };
    // Position of a ray that enters cell (ix, iy, iz) of a model with n cells per side whose first scene cell is base32 * 32
    auto place = [&](const int ix, const int iy, const int iz, const bool px, const bool py, const bool pz, const uint32_t n,
                     const uint32_t base32) __attribute__((always_inline)) {
        const uint32_t cell = (base32 << 5) + (((uint32_t)iz * n + (uint32_t)iy) * n + (uint32_t)ix);
        cv = (cell << 3) | 6u;
        const uint32_t rx = px ? n - 1u - (uint32_t)ix : (uint32_t)ix, ry = py ? n - 1u - (uint32_t)iy : (uint32_t)iy,
                       rz = pz ? n - 1u - (uint32_t)iz : (uint32_t)iz;
        rem = rx | (ry << 10) | (rz << 20) | kRemGuards;
        const uint32_t nn = n * n;
        wx = px ? ((1u << 3) | 1u) : ((0u - 1u) << 3);
        wy = py ? ((n << 3) | 3u) : (((0u - n) << 3) | 2u);
        wz = pz ? ((nn << 3) | 5u) : (((0u - nn) << 3) | 4u);
        fl = 0u;
        probe();  // the entry cell: no DDA step before it
    };
    // One step of the walk for the calling lanes: DDA advance (core.rs:151-167), bounds (core.rs:129-134), bitmap probe.
    auto walk_step = [&]() __attribute__((always_inline)) {
        const bool ax_ = (tmx < tmy) && (tmx < tmz);
        const bool ay_ = !(tmx < tmy) && (tmy < tmz);
        const bool az_ = !ax_ && !ay_;
        const double nx_ = tmx + tdx, ny_ = tmy + tdy, nz_ = tmz + tdz;
        tmx = ax_ ? nx_ : tmx;
        tmy = ay_ ? ny_ : tmy;
        tmz = az_ ? nz_ : tmz;
        // (selects of VALUES: a ?: between captured variables is lowered to a select of their addresses and sends them to the stack)
        const uint32_t wx_ = wx, wy_ = wy, wz_ = wz;
        const uint32_t wyz_ = ay_ ? wy_ : wz_;
        const uint32_t w_ = ax_ ? wx_ : wyz_;
        const uint32_t ryz_ = ay_ ? (1u << 10) : (1u << 20);
        const uint32_t r_ = ax_ ? 1u : ryz_;
        cv = (cv & ~7u) + w_;
        rem -= r_;
        if ((rem & kRemGuards) == kRemGuards) probe();
        else leave_model();  // left the grid without a hit
    };

    for (;;) {
        unsigned long long idle = __ballot(st == TS_IDLE);
        unsigned long long busy = __ballot(st != TS_IDLE);
        // ---- refill: idle lanes take the next items of the wave's block, in order
        if (!exhausted && ((uint32_t)__popcll(idle) >= A.refill || !busy)) {
            if (STATS) { dg_refills++; dg_refill_lanes += (uint64_t)__popcll(idle); dg_t0 = __builtin_amdgcn_s_memtime(); }
            for (;;) {
                idle = __ballot(st == TS_IDLE);
                if (!idle) break;
                if (pool_next == pool_end) {  // the next run of items (wave-uniform)
                    if (sq_items) {
                        uint32_t at = 0;
                        if (lane == 0) at = atomicAdd(A.ctr + (size_t)sq * CT_STRIDE + CT_CLAIM, A.claim);
                        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
                        if (at < sq_items) {
                            pool_next = at;
                            pool_end = min(at + A.claim, sq_items);
                            continue;
                        }
                    }
                    if (!stealing) {  // own sub-queues first
                        sq += trace_waves;
                        if (sq >= A.nq) stealing = true;
                    }
                    if (stealing) {
                        if (stolen >= steal_max) { exhausted = true; break; }
                        ++stolen;
                        sq = (wave_id + stolen * 2731u) % A.nq;  // (an odd stride: different waves try different sub-queues)
                    }
                    open_queue();
                    continue;
                }
                const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                const uint32_t avail = pool_end - pool_next;
#ifdef RTM_WAVE_TIMES
                wt_rays += min((uint32_t)__popcll(idle), avail);
#endif
                if (st == TS_IDLE && rank < avail) {
                    const uint32_t li_ = pool_next + rank;  // position in the sub-queue -> region, position in the region
                    const uint32_t rg = (li_ >= sq_n0 ? 1u : 0u) + (li_ >= sq_n01 ? 1u : 0u);
                    const size_t it = (size_t)sq * A.cap_i + (size_t)rg * A.cap_l + (li_ - (rg == 0u ? 0u : (rg == 1u ? sq_n0 : sq_n01)));
                    const size_t ps = IU[at64(IU_COUNT, IU_P, it)];
                    if (rg == 0u) d = mk(Pst[at64((int)A.pf_n, PF_D0 + 0, ps)], Pst[at64((int)A.pf_n, PF_D0 + 1, ps)], Pst[at64((int)A.pf_n, PF_D0 + 2, ps)]);
                    else d = mk(I[at64(IT_DCOUNT, IT_D + 0, it)], I[at64(IT_DCOUNT, IT_D + 1, it)], I[at64(IT_DCOUNT, IT_D + 2, it)]);
                    tmx = I[at64(IT_DCOUNT, IT_TM + 0, it)]; tmy = I[at64(IT_DCOUNT, IT_TM + 1, it)]; tmz = I[at64(IT_DCOUNT, IT_TM + 2, it)];
                    const uint32_t cw = IU[at64(IU_COUNT, IU_CELL, it)];
                    q = (uint32_t)it;
                    const uint32_t iw = IU[at64(IU_COUNT, IU_M, it)];
                    o = mk(Pst[at64((int)A.pf_n, PF_O + 0, ps)], Pst[at64((int)A.pf_n, PF_O + 1, ps)], Pst[at64((int)A.pf_n, PF_O + 2, ps)]);
                    model_i = iw & 0xFFFFu;
                    {   // tDelta = (step sign * cell size) / direction (core.rs:119-125; the product with +-1 is exact)
                        const gptr<double> mc = (gptr<double>)models_g[model_i].cell;
                        const double cx_ = mc[0], cy_ = mc[1], cz_ = mc[2];
                        tdx = ((iw & IM_SX) ? cx_ : -cx_) / d.x;
                        tdy = ((iw & IM_SY) ? cy_ : -cy_) / d.y;
                        tdz = ((iw & IM_SZ) ? cz_ : -cz_) / d.z;
                    }
                    const uint32_t mw = model_i < (uint32_t)kModelTab ? mtab[model_i] : (models_g[model_i].n | ((models_g[model_i].cell_base >> 5) << 10));
                    bm = kNoId; btri = kNoId; bt = 0.0;
                    if (STATS) dg_ray_its = 0;
                    st = TS_TRAV;
                    place((int)(cw & 1023u) - 1, (int)((cw >> 10) & 1023u) - 1, (int)((cw >> 20) & 1023u) - 1, (iw & IM_SX) != 0u,
                          (iw & IM_SY) != 0u, (iw & IM_SZ) != 0u, mw & 1023u, mw >> 10);
                }
                const uint32_t want = (uint32_t)__popcll(idle);
                pool_next += want < avail ? want : avail;
            }
            busy = __ballot(st != TS_IDLE);
            if (STATS) dg_tr += __builtin_amdgcn_s_memtime() - dg_t0;
        }
#ifdef RTM_WAVE_TIMES
        if (exhausted && !wt_exh) wt_exh = __builtin_amdgcn_s_memrealtime();
#endif
        if (!busy) {
            if (exhausted) break;
            continue;
        }
        // ---- rays that have left a model and have further models to visit (scenes with several models)
        if (__ballot(st == TS_ENTER)) {
            for (uint32_t m = 0; m < S.n_models; ++m) {
                const __attribute__((address_space(4))) RtmDevModel& M = as_const(S.models)[m];  // (wave-uniform: scalar loads)
                if (st == TS_ENTER && model_i == m) {
                    if (STATS) cnt.v[C_BBOX]++;
                    const double fx = 1.0 / d.x, fy = 1.0 / d.y, fz = 1.0 / d.z;
                    const double t1 = (M.bmin[0] - o.x) * fx, t2 = (M.bmax[0] - o.x) * fx;
                    const double t3 = (M.bmin[1] - o.y) * fy, t4 = (M.bmax[1] - o.y) * fy;
                    const double t5 = (M.bmin[2] - o.z) * fz, t6 = (M.bmax[2] - o.z) * fz;
                    const double tmin = fmax(fmax(fmin(t1, t2), fmin(t3, t4)), fmin(t5, t6));
                    const double tmax = fmin(fmin(fmax(t1, t2), fmax(t3, t4)), fmax(t5, t6));
                    if (tmax < 0.0 || tmin > tmax) {
                        model_i = m + 1;
                    } else {
                        const V3 P0 = ray_at(o, d, tmin);
                        const int n = (int)M.n;
                        const int ix = cell_coord((P0.x - M.bmin[0]) / M.cell[0], n);
                        const int iy = cell_coord((P0.y - M.bmin[1]) / M.cell[1], n);
                        const int iz = cell_coord((P0.z - M.bmin[2]) / M.cell[2], n);
                        const int sx = d.x > 0.0 ? 1 : -1, sy = d.y > 0.0 ? 1 : -1, sz = d.z > 0.0 ? 1 : -1;
                        const double nbx = (M.bmin[0] + M.hc2[0] * ((double)(ix + (sx > 0 ? 1 : 0)) + 0.5)) - M.hc[0];
                        const double nby = (M.bmin[1] + M.hc2[1] * ((double)(iy + (sy > 0 ? 1 : 0)) + 0.5)) - M.hc[1];
                        const double nbz = (M.bmin[2] + M.hc2[2] * ((double)(iz + (sz > 0 ? 1 : 0)) + 0.5)) - M.hc[2];
                        tmx = (nbx - P0.x) / d.x; tmy = (nby - P0.y) / d.y; tmz = (nbz - P0.z) / d.z;
                        tdx = ((double)sx * M.cell[0]) / d.x;
                        tdy = ((double)sy * M.cell[1]) / d.y;
                        tdz = ((double)sz * M.cell[2]) / d.z;
                        st = TS_TRAV;
                        place(ix, iy, iz, sx > 0, sy > 0, sz > 0, M.n, M.cell_base >> 5);
                    }
                }
            }
            if (st == TS_ENTER) finish_ray();  // no model left
        }
        if (!__ballot(st == TS_TRAV)) continue;
        // ---- traversal: all traversing lanes step together, whichever model each is in (what differs is per-lane state: cv, rem, wx, wy, wz)
        for (;;) {
            // A traversing lane is either WALKING (one DDA step and one bitmap probe per cell) or TESTING (its
            // cell's list is open); each iteration runs one of the two activities, whichever has the lanes.
            bool mine = st == TS_TRAV;
            if (mine && (fl & (F_OPEN | F_HAVE)) == F_HAVE) {  // core.rs:147-149: the first cell with any accepted hit ends the walk
                if (STATS) cnt.v[C_MHIT]++;
                model_result();
                leave_model();
                mine = false;
            }
            const uint32_t nW = (uint32_t)__popcll(__ballot(mine && !(fl & F_OPEN)));
            const uint32_t nT = (uint32_t)__popcll(__ballot(mine && (fl & F_OPEN)));
            if (nW + nT == 0u) break;
            if (STATS) {  // what the lanes that do not take part in this step are doing
                dg_idle += (uint64_t)__popcll(__ballot(st == TS_IDLE));
                dg_other += (uint64_t)__popcll(__ballot(st != TS_IDLE && !mine));
            }
            if (nT == 0u || (nW != 0u && nT * 64u < A.t_test_frac * (nW + nT))) {
                if (STATS) { dg_walks++; dg_walk_lanes += nW; dg_wait_t += nT; dg_t0 = __builtin_amdgcn_s_memtime(); }
                if (STATS && mine && !(fl & F_OPEN)) dg_ray_its++;
#pragma unroll
                for (int u = 0; u < RTM_FWALK_UNROLL; ++u) {
                    if (st == TS_TRAV && !(fl & F_OPEN)) walk_step();
                }
                if (STATS) dg_tw += __builtin_amdgcn_s_memtime() - dg_t0;
            } else {
                if (STATS) { dg_steps++; dg_step_lanes += nT; dg_wait_w += nW; dg_t0 = __builtin_amdgcn_s_memtime(); }
                const bool testing = mine && (fl & F_OPEN);
                if (STATS && testing) dg_ray_its += 1u << 16;
                uint32_t frame_i = 0u;    // long cells: index of the cell's frame (word 2 of its record)
                bool frame_known = false;
                if (testing && (fl & F_WINIT)) {  // a newly opened cell: entries already tested in the cell the ray came from are skipped
                    // list range [k, k_end), k < k_end, and the arrival mask: one 32-byte record
                    const gptr<uint32_t> ci = cell_info + (size_t)(cv >> 3) * 8;
                    const uint32_t arr = cv & 7u;
                    const uint32_t am = ci[2 + (arr < 6u ? arr : 0u)], w2 = ci[2];
                    k_end = ci[1];
                    k = ci[0];
                    const uint32_t len = k_end - k;
                    const uint32_t all = len >= 32u ? 0xFFFFFFFFu : ((1u << len) - 1u);
                    wmask = (A.cull && len <= 32u && arr < 6u) ? (all & am) : all;
                    fl &= ~F_HI;
                    if (len > 32u) { frame_i = w2; frame_known = true; fl |= F_LONG; }  // a long cell
                    fl &= ~F_WINIT;
                }
                // Long cells: the window's entries are first tested against their 8-byte boxes (rtm_internal.h cell_box8) and
                // only the entries whose box the ray's LINE meets keep their bit.  All arithmetic is f32 in the cell's own
                // frame -- origin at the point of the ray closest to the region's centre (computed in f64), one unit per
                // quantum -- where coordinates stay below a few hundred, so f32 rounding (1e-4 quantum) is far inside the
                // one-quantum enlargement of the boxes.  A non-finite ray rejects nothing.
                if (HAS_LONG && A.cull && __ballot(testing && (fl & F_LONG))) {
                    const bool scan = testing && (fl & F_LONG) != 0u;
                    fl &= ~F_LONG;
                    float olx = 0.f, oly = 0.f, olz = 0.f, ivx = 0.f, ivy = 0.f, ivz = 0.f;
                    uint32_t boff = 0u;
                    bool finite = false;
                    if (scan) {
                        if (!frame_known) frame_i = cell_info[(size_t)(cv >> 3) * 8 + 2];  // a later window of a cell with more than 64 entries
                        const gptr<double> F = cell_frame + (size_t)frame_i * 8;
                        const V3 ctr = ld3(F), iq = ld3(F + 3);
                        boff = (uint32_t)__double_as_longlong(F[6]);  // where the cell's boxes are, relative to its list positions
                        const V3 w = ctr - o;
                        // (any point of the ray near the region will do as the frame's origin: an approximate quotient)
                        const double ts = dot(w, d) * (double)__builtin_amdgcn_rcpf((float)dot(d, d));
                        const V3 pl = (ray_at(o, d, ts) - ctr) * iq;
                        olx = (float)(pl.x + 127.5); oly = (float)(pl.y + 127.5); olz = (float)(pl.z + 127.5);
                        const float dlx = (float)(d.x * iq.x), dly = (float)(d.y * iq.y), dlz = (float)(d.z * iq.z);
                        // (the screen is used only where f32 is good for 1e-2 quantum: frame origin within 2^16 quanta of the region)
                        finite = fabsf(olx) + fabsf(oly) + fabsf(olz) < 65536.0f && fabsf(dlx) + fabsf(dly) + fabsf(dlz) < __builtin_inff();
                        // 1 / direction, kept finite: a (near-)zero component makes its slab [-big, +big] or empty, as it should
                        // (the hardware's approximate reciprocal: one ulp of iv moves a face by 1e-5 quantum at most)
                        ivx = fabsf(dlx) >= 1e-30f ? __builtin_amdgcn_rcpf(dlx) : __builtin_copysignf(1e30f, dlx);
                        ivy = fabsf(dly) >= 1e-30f ? __builtin_amdgcn_rcpf(dly) : __builtin_copysignf(1e30f, dly);
                        ivz = fabsf(dlz) >= 1e-30f ? __builtin_amdgcn_rcpf(dlz) : __builtin_copysignf(1e30f, dlz);
                    }
                    const float cx = -olx * ivx, cy = -oly * ivy, cz = -olz * ivz;
                    const uint32_t nwin = scan ? min(k_end - k, 64u) : 0u;  // the window and the 32 entries after it
                    // Which corner of a box the ray's line meets first is the same for every entry: the sign of the direction
                    // picks the near and the far face per axis ONCE, as two byte selectors, and an entry's six coordinates are
                    // put in (near x, near y, near z, far x | far y, far z) order by two byte permutes -- the slab test is then
                    // max3 of the near times against min3 of the far ones, without a min and a max per axis and entry.
                    // (iv is finite and t = x * iv + c is monotonic in x, so the near face's time IS min(t(x0), t(x1)).)
                    const bool px_ = !(ivx < 0.f), py_ = !(ivy < 0.f), pz_ = !(ivz < 0.f);
                    const uint32_t sel0 = (px_ ? 0u : 3u) | ((py_ ? 1u : 4u) << 8) | ((pz_ ? 2u : 5u) << 16) | ((px_ ? 3u : 0u) << 24);
                    const uint32_t sel1 = (py_ ? 4u : 1u) | ((pz_ ? 5u : 2u) << 8) | 0x0C0C0000u;
                    // the cell's boxes are packed from a 16-byte boundary (two entries per load) and the array is padded by a
                    // window's length, so a pass reads 64 entries from the cell's first whatever the list's length (the bits of
                    // entries beyond it are dropped below); lanes that do not scan read entry 0
                    const uint32_t bbase = scan ? k + boff : 0u;
                    uint32_t pass_h[2] = {0u, 0u};
                    typedef float v2f __attribute__((ext_vector_type(2)));
                    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                    const __attribute__((address_space(1))) u4* const box4 = (const __attribute__((address_space(1))) u4*)cell_box8 + (bbase >> 1);
#pragma unroll
                    for (uint32_t half = 0; half < 2u; ++half) {
                        uint32_t pass = 0u, n_done = 0u;  // entries of this half looked at (wave-uniform); their bits enter `pass` from below
                        for (uint32_t j0 = half * 32u; j0 < half * 32u + 32u && __ballot(j0 < nwin); j0 += 8u) {
                            n_done += 8u;
                            u4 bx[4];
#pragma unroll
                            for (uint32_t u = 0; u < 4u; ++u) bx[u] = box4[(j0 >> 1) + u];
#pragma unroll
                            for (uint32_t u = 0; u < 8u; ++u) {
                                const uint32_t w0 = (u & 1u) ? bx[u >> 1].z : bx[u >> 1].x, w1 = (u & 1u) ? bx[u >> 1].w : bx[u >> 1].y;
                                const uint32_t q0 = __builtin_amdgcn_perm(w1, w0, sel0), q1 = __builtin_amdgcn_perm(w1, w0, sel1);
                                const float nx = (float)(q0 & 255u), ny = (float)((q0 >> 8) & 255u), nz = (float)((q0 >> 16) & 255u);
                                const float fx = (float)(q0 >> 24), fy = (float)(q1 & 255u), fz = (float)((q1 >> 8) & 255u);
                                // (x - ol) * iv as one fused multiply-add with c = -ol * iv: its rounding moves a face by 1e-5 quantum
                                const v2f tx = __builtin_elementwise_fma(v2f{nx, fx}, v2f{ivx, ivx}, v2f{cx, cx});
                                const v2f ty = __builtin_elementwise_fma(v2f{ny, fy}, v2f{ivy, ivy}, v2f{cy, cy});
                                const v2f tz = __builtin_elementwise_fma(v2f{nz, fz}, v2f{ivz, ivz}, v2f{cz, cz});
                                const float tlo = fmaxf(fmaxf(tx.x, ty.x), tz.x);
                                const float thi = fminf(fminf(tx.y, ty.y), tz.y);
                                // pass = pass << 1 | !(tlo > thi), as a compare into the carry and one add-with-carry (a comparison
                                // with NaN is false: the entry stays); the entries' bits arrive in reverse order
                                asm("v_cmp_ngt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(pass) : "v"(tlo), "v"(thi) : "vcc");
                            }
                        }
                        pass_h[half] = n_done ? __builtin_bitreverse32(pass) >> (32u - n_done) : 0u;
                    }
                    unsigned long long pass = ((unsigned long long)pass_h[1] << 32) | pass_h[0];
                    pass &= nwin >= 64u ? ~0ull : ((1ull << nwin) - 1ull);
                    if (scan) {  // (a window left empty moves on at the end of this step)
                        if (!finite) pass = nwin >= 64u ? ~0ull : ((1ull << nwin) - 1ull);
                        wmask &= (uint32_t)pass;
                        wmask_hi = (uint32_t)(pass >> 32);
                        fl |= F_HI;
                    }
                }
                if (nT <= 4u && A.coop) {
                    // A few rays are testing (the end of a launch: some long rays are left).  The other lanes test the
                    // window's entries for them -- 32 lanes per ray and one entry per lane with one or two rays, 16 lanes
                    // and two entries with three or four -- so a cell costs one step whatever its length; the accepted
                    // entries are then applied to the ray in list order, exactly as the sequential loop does
                    // (core.rs:140-142).  (Eight rays with four entries per lane spill registers: measured slower.)
                    unsigned long long tm = __ballot(testing);
                    const uint32_t gsh = nT <= 2u ? 5u : 4u;            // log2 of the lanes per ray
                    const uint32_t grp = lane >> gsh;                   // the ray this lane helps
                    int owner[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        owner[h] = tm ? __ffsll((long long)tm) - 1 : 0;
                        tm &= tm - 1ull;
                    }
                    const int src = grp == 0u ? owner[0] : grp == 1u ? owner[1] : grp == 2u ? owner[2] : owner[3];
                    const V3 so = mk(__shfl(o.x, src, 64), __shfl(o.y, src, 64), __shfl(o.z, src, 64));
                    const V3 sd = mk(__shfl(d.x, src, 64), __shfl(d.y, src, 64), __shfl(d.z, src, 64));
                    const uint32_t sk = (uint32_t)__shfl((int)k, src, 64), swm = (uint32_t)__shfl((int)wmask, src, 64);
                    const uint32_t passes = 32u >> gsh;                 // entries per lane
                    bool okc[2] = {false, false};
                    double ttc[2] = {0.0, 0.0};
                    uint32_t tic[2] = {0u, 0u};
#pragma unroll
                    for (uint32_t ps = 0; ps < 2u; ++ps) {
                        if (ps >= passes) break;
                        const uint32_t e = (lane & ((1u << gsh) - 1u)) + (ps << gsh);
                        if (grp < nT && ((swm >> e) & 1u)) {
                            const gptr<double> tp = cell_rec + (uint32_t)((sk + e) * 10u);
                            const V3 v0 = ld3(tp), e1 = ld3(tp + 3), e2 = ld3(tp + 6);
                            tic[ps] = (uint32_t)__double_as_longlong(tp[9]);
                            const V3 pv = cross(sd, e2);
                            const double det = dot(e1, pv);
                            const double inv = 1.0 / det;
                            const V3 tv = so - v0;
                            const double u = dot(tv, pv) * inv;
                            const V3 qv = cross(tv, e1);
                            const double v = dot(sd, qv) * inv;
                            ttc[ps] = dot(e2, qv) * inv;
                            okc[ps] = !(fabs(det) < kF64Eps) && !(u < 0.0 || u > 1.0) && !(v < 0.0 || u + v > 1.0) && !(ttc[ps] < 0.0);
                        }
                    }
                    const unsigned long long okm[2] = {__ballot(okc[0]), __ballot(okc[1])};
                    const unsigned long long gmask = gsh == 5u ? 0xFFFFFFFFull : 0xFFFFull;
                    for (uint32_t h = 0; h < nT; ++h) {  // (wave-uniform)
                        const uint32_t own = (uint32_t)(h == 0u ? owner[0] : h == 1u ? owner[1] : h == 2u ? owner[2] : owner[3]);
#pragma unroll
                        for (uint32_t ps = 0; ps < 2u; ++ps) {  // pass 0 holds the lower entries: ascending = list order
                            unsigned long long om = okm[ps] & (gmask << (h << gsh));
                            while (om) {
                                const int c = __ffsll((long long)om) - 1;
                                om &= om - 1ull;
                                const double tc = __shfl(ttc[ps], c, 64);
                                const uint32_t trc = (uint32_t)__shfl((int)tic[ps], c, 64);
                                if (lane == own) {
                                    const bool take = !(fl & F_HAVE) || tc < m_t;
                                    m_t = take ? tc : m_t;
                                    m_tri = take ? trc : m_tri;
                                    fl |= F_HAVE;
                                }
                            }
                        }
                    }
                    if (testing) {
                        if (STATS) cnt.v[C_TRI] += (uint32_t)__popc(wmask);
                        wmask = 0u;
                    }
                } else if (testing) {
                    bool have = (fl & F_HAVE) != 0u;
                    // Up to kWide consecutive list entries, evaluated branch-free (model.rs:45-72): the
                    // reference's early returns become one predicate built from the same comparisons (so NaNs take
                    // the same side).  Entries are applied in list order (core.rs:140-142).
                    constexpr int kWide = RTM_FTRI_WIDE;
                    uint32_t ti[kWide];
                    double tt[kWide];
                    bool ok[kWide];
                    uint32_t cntw = 0;
#pragma unroll
                    for (int j = 0; j < kWide; ++j) {  // the next untested entries, in list order
                        const bool any = wmask != 0u && (uint32_t)j < A.wide;  // (A.wide slots per step: see render.hip)
                        const uint32_t b = any ? (uint32_t)__builtin_ctz(wmask) : 0u;
                        ti[j] = k + b;
                        wmask = any ? (wmask & (wmask - 1u)) : wmask;
                        cntw += any ? 1u : 0u;
                    }
#pragma unroll
                    for (int j = 0; j < kWide; ++j) {
                        tt[j] = 0.0;
                        ok[j] = false;
                        // slots beyond the second are evaluated only when some lane of the wave has that many
                        // entries left in its cell (wave-uniform branch: lists along a ray are mostly short)
                        const unsigned long long wants = __ballot((uint32_t)j < cntw);
                        if (j >= 2 && !wants) continue;
                        if (STATS) { dg_slots++; dg_slot_lanes += (uint64_t)__popcll(wants); }
                        const gptr<double> tp = cell_rec + (uint32_t)(ti[j] * 10u);  // (at most kMaxListEntries positions: the product fits 32 bits)
                        const V3 v0 = ld3(tp), e1 = ld3(tp + 3), e2 = ld3(tp + 6);
                        ti[j] = (uint32_t)__double_as_longlong(tp[9]);  // the record ends with the triangle's index
                        const V3 pv = cross(d, e2);
                        const double det = dot(e1, pv);
                        const double inv = 1.0 / det;
                        const V3 tv = o - v0;
                        const double u = dot(tv, pv) * inv;
                        const V3 qv = cross(tv, e1);
                        const double v = dot(d, qv) * inv;
                        tt[j] = dot(e2, qv) * inv;
                        ok[j] = (uint32_t)j < cntw && !(fabs(det) < kF64Eps) && !(u < 0.0 || u > 1.0) && !(v < 0.0 || u + v > 1.0) &&
                                !(tt[j] < 0.0);
                    }
#pragma unroll
                    for (int j = 0; j < kWide; ++j) {
                        const bool take = ok[j] && (!have || tt[j] < m_t);
                        m_t = take ? tt[j] : m_t;
                        m_tri = take ? ti[j] : m_tri;
                        have = have || take;
                    }
                    fl = (fl & ~F_HAVE) | (have ? F_HAVE : 0u);
                    if (STATS) cnt.v[C_TRI] += cntw;
                }
                if (testing && wmask == 0u && (fl & F_OPEN)) {  // window done: the next 32 entries of a long list, or the cell is finished
                    const uint32_t adv = (HAS_LONG && (fl & F_HI)) ? 64u : 32u;  // entries dealt with so far, counted from k
                    if (HAS_LONG && (fl & F_HI) && wmask_hi != 0u) {  // the second half of what the screen covered
                        k += 32u;
                        wmask = wmask_hi;
                        wmask_hi = 0u;
                        fl &= ~F_HI;
                    } else if (k_end - k > adv) {
                        k += adv;
                        const uint32_t len = k_end - k;
                        wmask = len >= 32u ? 0xFFFFFFFFu : ((1u << len) - 1u);
                        fl = (fl & ~F_HI) | F_LONG;  // (only long cells have further windows: their entries are screened by their boxes first)
                    } else {
                        fl &= ~(F_OPEN | F_HI);
                    }
                }
                if (STATS) dg_tt += __builtin_amdgcn_s_memtime() - dg_t0;
            }
            const unsigned long long still = __ballot(st == TS_TRAV);
            if (!still) break;
            if (!exhausted && (uint32_t)__popcll(__ballot(st == TS_IDLE)) >= A.refill) break;
            if ((uint32_t)__popcll(still) < A.t_min_lanes && __ballot(st == TS_ENTER)) break;
        }
        // a hit accepted in the last iteration (the loop may leave before its next top)
        if (st == TS_TRAV && (fl & (F_OPEN | F_HAVE)) == F_HAVE) {
            if (STATS) cnt.v[C_MHIT]++;
            model_result();
            leave_model();
        }
    }
#ifdef RTM_WAVE_TIMES
    if (A.dbg && lane == 0) {
        unsigned long long* const w = A.dbg + (size_t)wave_id * 4;
        w[0] = wt_begin; w[1] = wt_exh; w[2] = __builtin_amdgcn_s_memrealtime(); w[3] = wt_rays;
    }
#endif
    if (STATS) {  // longest ray of the wave (one atomic per wave, not per ray)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            dg_max_walks = max(dg_max_walks, (uint32_t)__shfl_xor((int)dg_max_walks, off, 64));
            dg_max_tests = max(dg_max_tests, (uint32_t)__shfl_xor((int)dg_max_tests, off, 64));
        }
        if (lane == 0) {
            atomicMax(out_cnt + 26, (unsigned long long)dg_max_walks);
            atomicMax(out_cnt + 27, (unsigned long long)dg_max_tests);
        }
    }
    if (STATS && lane == 0) {
        const unsigned long long el = (unsigned long long)(__builtin_amdgcn_s_memtime() - dg_begin);
        atomicAdd(out_cnt + 20, el);
        { const unsigned long long its = dg_steps + dg_walks; atomicMax(out_cnt + 21, (el << 24) | (its > 0xFFFFFFull ? 0xFFFFFFull : its)); }
        atomicAdd(out_cnt + 22, 1ull);
        atomicAdd(out_cnt + 23, (unsigned long long)dg_tr);
        atomicAdd(out_cnt + 24, (unsigned long long)dg_tw);
        atomicAdd(out_cnt + 25, (unsigned long long)dg_tt);
    }
    flush_counters<STATS>(cnt, out_cnt);
    if (STATS && lane == 0) {
        const uint64_t dv[8] = {dg_steps, dg_step_lanes, dg_walks, dg_walk_lanes, dg_slots, dg_slot_lanes, dg_refills, dg_refill_lanes};
        for (int i = 0; i < 8; ++i)
            if (dv[i]) atomicAdd(out_cnt + 9 + i, (unsigned long long)dv[i]);
        const uint64_t dw[4] = {dg_idle, dg_other, dg_wait_t, dg_wait_w};
        for (int i = 0; i < 4; ++i)
            if (dw[i]) atomicAdd(out_cnt + 28 + i, (unsigned long long)dw[i]);
    }
}

// ================================================================================================
// resolve: pixel = sum over samples, in sample order, of colour / n (main.rs:169)
__global__ __launch_bounds__(kBlock) void front_resolve_kernel(FrontArgs A, double* __restrict__ out_rgb) {
    const uint32_t pid = blockIdx.x * kBlock + threadIdx.x;
    if (pid >= A.npix) return;
    const uint32_t tile = pid >> 6, in = pid & 63u;
    const uint32_t x = (tile % A.tiles_x) * 8u + (in & 7u);
    const uint32_t lyc = (tile / A.tiles_x) * 8u + (in >> 3);
    if (!(x < A.W && lyc < A.rows)) return;
    const double n_spp = (double)A.spp;
    V3 px = mk(0, 0, 0);
    for (uint32_t s = 0; s < A.spp; ++s) {
        const size_t p = (size_t)(s * A.n_tiles + tile) * 64u + in;  // where the shade kernel put the sample's colour
        px = px + mk(A.fin[at64(3, 0, p)], A.fin[at64(3, 1, p)], A.fin[at64(3, 2, p)]) / n_spp;
    }
    const size_t oi = (size_t)(A.row0 + lyc) * A.W + x;
    out_rgb[oi * 3] = px.x;
    out_rgb[oi * 3 + 1] = px.y;
    out_rgb[oi * 3 + 2] = px.z;
}

}  // namespace rtf
