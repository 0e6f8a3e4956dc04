// rt_wave.h -- the single-kernel formulation: a wave-scheduled path-state machine for gfx950.  NOT the product
// path since round 1: rtm_render* run the ray-front pipeline of rt_front.h.  This kernel stays selectable
// (RTM_KERNEL=wave, rtm_set_option "kernel") as the fallback for scenes the front path does not take (more than
// 2^26 list entries, more than 1023 cells per side) and as a second device implementation the parity tests hold
// against the oracle.
//
// Why this shape (DESIGN.md "Kernel"): the per-ray arithmetic is fixed by bit parity with the reference, so
// throughput is decided by (a) how many of a wavefront's 64 lanes do useful work per issued f64 instruction
// and (b) how many wavefronts a SIMD can hold to cover L2 latency and the long f64 divide/sqrt chains.
// A straight per-lane transliteration of cast()/hit() (round 1's `v1` kernel, retired in round 3) runs at ~10 % lane
// utilisation on the bunny scene: shadow rays, bounce rays and the 3D-DDA sit at different code addresses
// and every DDA loop runs for the slowest lane.  Here
//   * every lane owns one pixel at a time and walks the reference's sequence
//       camera ray -> hit -> [shadow ray per light] -> bounce -> ...      (main.rs:57-132,159-170)
//     as a state machine with exactly ONE copy of each expensive block;
//   * the wave alternates, by ballot/popcount thresholds, between
//       S "shade/set-up": consume the finished query, produce the next ray, analytic primitives, slab test,
//                         DDA initialisation (core.rs:68-125, main.rs:65-125) for all waiting lanes at once,
//       T "traverse"    : one triangle test per traversing lane per step, cell advance folded in
//                         (core.rs:129-167, model.rs:43-74);
//   * lanes are persistent: a lane that finishes its pixel claims the next one (8x8 tiles handed to waves by
//     one atomic each), so there is no tail of idle lanes inside a wave;
//   * per-lane state that only the S round touches (pixel/sample accumulators, shading normal, bounce
//     direction, diffuse sum) lives in LDS, not VGPRs: the T loop then fits the register budget of
//     RTM_WAVE_OCC waves per SIMD;
//   * empty cells are skipped through a cell-occupancy bitmap staged in LDS (one bit per cell, derived from
//     the offset array at upload time) -- the same DDA walk, without touching the lists.
// The arithmetic of every block is the same op sequence as rt_device.h / the reference: results are
// bit-identical, which tests/test_parity_gpu.py checks for both kernels.
#pragma once
#include "rt_device.h"

#pragma clang fp contract(off)

namespace rtw {
using namespace rtd;

#ifndef RTM_WAVE_OCC
#define RTM_WAVE_OCC 4  // waves per SIMD the register allocator must leave room for
#endif
#ifndef RTM_WALK_UNROLL
#define RTM_WALK_UNROLL 2  // DDA steps per walk iteration of the traversal loop
#endif
#ifndef RTM_TRI_WIDE
#define RTM_TRI_WIDE 4  // list entries tested per traversal step
#endif
constexpr int kBlock = 256;
constexpr int kColdFields = 17;                       // doubles of S-only state per lane, see enum below
constexpr int kColdBytes = kColdFields * kBlock * 8;  // 34,816 B per workgroup
constexpr int kOccLdsWordsMax = 8192;                 // at most 32 KiB of bitmaps in LDS; the rest stay in HBM

enum : uint32_t { ST_CONSUME = 0, ST_ENTER = 1, ST_TRAV = 2, ST_DEAD = 3, ST_FETCH = 4 };
enum : uint32_t { PH_START = 0, PH_PATH = 1, PH_SHADOW = 2 };
enum : int { F_PX = 0, F_COLOR = 3, F_ATT = 6, F_NRM = 7, F_BD = 10, F_DIFF = 13, F_DL2 = 16 };

struct RenderArgsW {
    uint32_t W, H, spp, max_bounces;
    uint64_t seed;
    uint32_t row_begin, row_stride, n_rows, flags;
    uint32_t s_threshold;  // per 64 live lanes: run an S round when at least this many wait (or nobody traverses)
    uint32_t t_test_frac;  // per 64 traversing lanes: run a test step when at least this many have an open list, else a walk step
    uint32_t t_min_lanes;  // per 64 live lanes: leave the T loop below this many traversing lanes (if S is due)
    uint32_t tiles_x, n_tiles;  // 8x8 pixel tiles over width x n_rows; pixel ids are tile-major
    uint32_t occ_words;         // bitmap words staged in LDS (models in order, as many as fit)
    uint32_t* work_counter;     // zeroed before the launch: next unclaimed tile (persistent fetch)
};

template <bool STATS>
__global__ __launch_bounds__(kBlock, RTM_WAVE_OCC) void render_wave_kernel(RtmDevScene S, RenderArgsW A, double* __restrict__ out_rgb,
                                                                            uint32_t* __restrict__ out_obj, uint32_t* __restrict__ out_tri,
                                                                            unsigned long long* __restrict__ out_cnt) {
    extern __shared__ double lds_raw[];
    double* const cold = lds_raw;                                          // [kColdFields][kBlock]
    uint32_t* const occ_lds = reinterpret_cast<uint32_t*>(lds_raw + kColdFields * kBlock);  // [A.occ_words]
    const uint32_t tid = threadIdx.x;
    auto ldc = [&](int f) -> double { return cold[f * kBlock + tid]; };
    auto stc = [&](int f, double v) { cold[f * kBlock + tid] = v; };
    auto ldc3 = [&](int f) -> V3 { return mk(ldc(f), ldc(f + 1), ldc(f + 2)); };
    auto stc3 = [&](int f, V3 v) { stc(f, v.x); stc(f + 1, v.y); stc(f + 2, v.z); };

    // ---- stage the occupancy bitmaps (models back to back; those that do not fit stay in HBM)
    {
        uint32_t base = 0;
        for (uint32_t m = 0; m < S.n_models; ++m) {
            const uint32_t n = S.models[m].n;
            const uint32_t words = (n * n * n + 31u) >> 5;
            if (base + words > A.occ_words) break;
            const gptr<uint32_t> src = as_global(S.models[m].occupancy);
            for (uint32_t i = tid; i < words; i += kBlock) occ_lds[base + i] = src[i];
            base += words;
        }
    }
    __syncthreads();

    const uint32_t lane = tid & 63u;
    Counters cnt;
    if (STATS)
        for (int i = 0; i < 9; ++i) cnt.v[i] = 0;
    uint32_t pool_next = 0, pool_end = 0;  // wave-uniform: the wave's claimed tile
    const double n_spp = (double)A.spp;
    const int K = (int)A.max_bounces;

    // ---- hot lane state (VGPRs) ---------------------------------------------------------------------
    uint32_t st = ST_FETCH, phase = PH_START;
    uint32_t x = 0, ly = 0, pixel = 0, s = 0, li = 0, model_i = 0, mat = 0;  // mat: material of the surface being shaded
    int bounce = 0;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);  // current query ray; during shadow queries o is the shaded point P
    uint32_t obj0 = kNoId, tri0 = kNoId;
    double best = 0.0;                    // query result so far (core.rs:69-85)
    bool found = false;
    uint32_t h_obj = kNoId, h_tri = kNoId, h_mat = 0, h_kp = 0;  // h_kp = kind << 30 | prim
    int ix = 0, iy = 0, iz = 0;           // traversal state (core.rs:99-128)
    double tmx = 0, tmy = 0, tmz = 0, tdx = 0, tdy = 0, tdz = 0;
    uint32_t k = 0, k_end = 0, m_tri = 0;
    double m_t = 0.0;
    bool have = false, fresh = false, open = false;  // open: the current cell's list [k, k_end) is not exhausted

    auto consider = [&](double t, uint32_t obj, uint32_t mt, uint32_t kp, uint32_t tri) {
        if (t <= kEps) return;                                  // core.rs:75-77
        if (t < best || (found && t == best && obj < h_obj)) {  // core.rs:79, ties to the earlier object
            best = t; found = true;
            h_obj = obj; h_mat = mt; h_kp = kp; h_tri = tri;
        }
    };
    // core.rs:151-167 with the step signs of core.rs:102-104.  Written with selects, not branches: hipcc turns the
    // branchy form into an indexed update of a stack array {ix,iy,iz}/{tmx,tmy,tmz}, i.e. scratch traffic per cell.
#define RTW_DDA_ADVANCE()                                                      \
    do {                                                                       \
        const bool ax_ = (tmx < tmy) && (tmx < tmz);                           \
        const bool ay_ = !(tmx < tmy) && (tmy < tmz);                          \
        const bool az_ = !ax_ && !ay_;                                         \
        ix += ax_ ? (d.x > 0.0 ? 1 : -1) : 0;                                  \
        iy += ay_ ? (d.y > 0.0 ? 1 : -1) : 0;                                  \
        iz += az_ ? (d.z > 0.0 ? 1 : -1) : 0;                                  \
        const double nx_ = tmx + tdx, ny_ = tmy + tdy, nz_ = tmz + tdz;        \
        tmx = ax_ ? nx_ : tmx;                                                 \
        tmy = ay_ ? ny_ : tmy;                                                 \
        tmz = az_ ? nz_ : tmz;                                                 \
    } while (0)

    uint64_t dg_s_rounds = 0, dg_s_lanes = 0, dg_t_steps = 0, dg_t_lanes = 0;  // scheduler diagnostics (STATS build only)
    uint64_t dg_s_cyc = 0, dg_t_cyc = 0, dg_t0 = 0;
    uint64_t dg_sb[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg_tm = 0;  // S-round sub-block cycles
#define RTW_MARK(i) do { if (STATS) { const uint64_t n_ = __builtin_amdgcn_s_memtime(); dg_sb[i] += n_ - dg_tm; dg_tm = n_; } } while (0)
    uint64_t dg_walk_iters = 0, dg_walk_lanes = 0, dg_slots = 0, dg_slot_lanes = 0, dg_walk_steps = 0;
    const uint64_t dg_begin = STATS ? __builtin_amdgcn_s_memtime() : 0;
    for (;;) {
        // ---- hand out pixels to idle lanes
        for (;;) {
            const unsigned long long need = __ballot(st == ST_FETCH);
            if (!need) break;
            if (pool_next == pool_end) {
                const int first = __ffsll((long long)need) - 1;
                uint32_t base = 0;
                if (lane == (uint32_t)first) base = atomicAdd(A.work_counter, 1u);
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                if (base >= A.n_tiles) {  // frame exhausted
                    if (st == ST_FETCH) st = ST_DEAD;
                    break;
                }
                pool_next = base * 64u;
                pool_end = pool_next + 64u;
            }
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const uint32_t avail = pool_end - pool_next;
            if (st == ST_FETCH && rank < avail) {
                const uint32_t id = pool_next + rank, tile = id >> 6, in = id & 63u;
                x = (tile % A.tiles_x) * 8u + (in & 7u);
                ly = (tile / A.tiles_x) * 8u + (in >> 3);
                if (x < A.W && ly < A.n_rows) {  // ids outside a ragged right/bottom tile are skipped
                    pixel = x + A.W * (A.row_begin + ly * A.row_stride);  // main.rs:160-161,177
                    stc3(F_PX, mk(0, 0, 0));
                    s = 0;
                    phase = PH_START;
                    st = ST_CONSUME;
                }
            }
            const uint32_t want = (uint32_t)__popcll(need);
            pool_next += want < avail ? want : avail;
        }
        const unsigned long long waiting = __ballot(st == ST_CONSUME || st == ST_ENTER);
        const unsigned long long traversing = __ballot(st == ST_TRAV);
        if (!waiting && !traversing) break;
        const uint32_t live = (uint32_t)__popcll(waiting | traversing);
        const uint32_t s_need = (A.s_threshold * live + 63u) >> 6;
        if (!traversing || (uint32_t)__popcll(waiting) >= s_need) {
            // =========================== S round ===========================
            if (STATS) { dg_s_rounds++; dg_s_lanes += (uint64_t)__popcll(waiting); dg_t0 = __builtin_amdgcn_s_memtime(); dg_tm = dg_t0; }
            if (st == ST_CONSUME) {
                // "Decide, then do": lanes of a wave sit in different phases, so every expensive block below exists
                // once and serves whichever lanes need it this round (one Philox pair, one normal normalisation, one
                // direction normalisation), instead of once per phase.
                // ---- 0. cheap per-lane decisions (integer state only)
                const bool path = phase == PH_PATH, shadow = phase == PH_SHADOW;
                const bool path_hit = path && found, path_miss = path && !found;
                if (path && bounce == 0 && s == 0) { obj0 = found ? h_obj : kNoId; tri0 = found ? h_tri : kNoId; }
                if (path_hit) mat = h_mat;
                const bool metal = S.materials[(size_t)mat * 4 + 3] != 0.0;  // material of the surface being shaded
                const bool plastic_hit = path_hit && !metal;
                const bool finish = (path_hit && S.n_lights == 0u) || (shadow && li + 1u >= S.n_lights);
                const bool next_shadow = (path_hit && S.n_lights != 0u) || (shadow && li + 1u < S.n_lights);
                const bool end_sample = path_miss || (finish && bounce + 1 >= K);
                const bool last_sample = end_sample && s + 1u >= A.spp;
                const bool gen = phase == PH_START || (end_sample && !last_sample);
                // ---- 1. random numbers: Vec3::random() of this bounce (vec3.rs:160-162) or the next sample's jitter
                //         (main.rs:162-163); both are pure functions of (seed, pixel, sample, stream)
                uint32_t ra[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
                if (plastic_hit || gen) {
                    const uint32_t smp = gen ? (phase == PH_START ? s : s + 1u) : s;
                    philox(A.seed, pixel, smp, gen ? 0u : 1u + 2u * (uint32_t)bounce, ra);
                }
                if (plastic_hit) philox(A.seed, pixel, s, 2u + 2u * (uint32_t)bounce, rb);
                RTW_MARK(0);
                // ---- 2. surface point and shading normal of a path hit (core.rs:89,169-177,195,209; main.rs:68-73)
                if (path_hit) {
                    const V3 P = ray_at(o, d, best);
                    const uint32_t kind = h_kp >> 30, prim = h_kp & 0x3FFFFFFFu;
                    V3 raw;
                    if (kind == 0u) raw = ray_at(o, d, best) - ld3(S.spheres + (size_t)prim * 4);
                    else if (kind == 1u) raw = ld3(S.planes + (size_t)prim * 4);  // used as written, not normalised
                    else raw = model_normal_raw(S.models[prim], h_tri, P);
                    V3 nn = normalize(raw);
                    if (kind == 1u) nn = raw;
                    if (dot(d, nn) > 0.001) nn = nn * -1.0;
                    // bounce direction (main.rs:50-55): metal reflects (not renormalised, vec3.rs:172-175); plastic is
                    // normalize(n + random()), whose normalisation is done when the bounce ray is issued (step 6)
                    const V3 rnd = mk(u01(ra[0], ra[1]) * (1.0 - 0.0) + 0.0, u01(ra[2], ra[3]) * (1.0 - 0.0) + 0.0,
                                      u01(rb[0], rb[1]) * (1.0 - 0.0) + 0.0);
                    const V3 refl = d - nn * 2.0 * dot(d, nn);
                    const V3 diff = nn + rnd;
                    stc3(F_NRM, nn);
                    stc3(F_BD, sel3(metal, refl, diff));
                    stc3(F_DIFF, mk(0, 0, 0));
                    o = P;  // the shadow rays start here and so does the bounce (main.rs:82,121)
                    li = 0;
                }
                RTW_MARK(1);
                // ---- 3. result of a shadow query (main.rs:86-104), light li; o == P
                if (shadow) {
                    const double* L = S.lights + (size_t)li * 8;
                    const double dl2 = ldc(F_DL2);
                    bool occluded = false;
                    if (found) occluded = sqdist(ray_at(o, d, best), o) <= dl2;
                    if (!occluded) {
                        const double eff = dot(d, ldc3(F_NRM));
                        if (eff > 0.001) {
                            const double dd = fmax(dl2 / (L[6] * L[6]), 1.0);
                            const double decay = 1.0 / dd;
                            stc3(F_DIFF, ldc3(F_DIFF) + ld3(L + 3) * decay * eff);
                        }
                    }
                    ++li;
                }
                // ---- 4. end of the light loop: add the surface's colour, attenuate (main.rs:107-125)
                V3 newdir = mk(0, 0, 1);
                bool need_norm = false;
                if (path_miss) stc3(F_COLOR, ldc3(F_COLOR) + mk(1, 1, 1) * ldc(F_ATT));  // main.rs:127
                if (finish) {
                    const double* mp = S.materials + (size_t)mat * 4;
                    const double att = ldc(F_ATT);
                    stc3(F_COLOR, ldc3(F_COLOR) + ld3(mp) * att * (ldc3(F_DIFF) * (metal ? 0.2 : 0.9)));
                    stc(F_ATT, att * (metal ? 0.8 : 0.1));
                    newdir = ldc3(F_BD);
                    need_norm = !metal;
                    ++bounce;
                    phase = PH_PATH;
                }
                // ---- 5. end of a sample: pixel accumulation in sample order (main.rs:169)
                if (end_sample) {
                    const V3 px = ldc3(F_PX) + ldc3(F_COLOR) / n_spp;
                    ++s;
                    if (last_sample) {
                        const size_t oi = (size_t)ly * A.W + x;
                        out_rgb[oi * 3] = px.x;
                        out_rgb[oi * 3 + 1] = px.y;
                        out_rgb[oi * 3 + 2] = px.z;
                        if (out_obj) out_obj[oi] = obj0;
                        if (out_tri) out_tri[oi] = tri0;
                        st = ST_FETCH;
                    } else {
                        stc3(F_PX, px);
                    }
                }
                RTW_MARK(2);
                // ---- 6. the next ray: shadow ray to light li (main.rs:80-85), bounce ray, or camera ray (main.rs:160-168)
                if (next_shadow) {
                    const V3 lp = ld3(S.lights + (size_t)li * 8);
                    newdir = lp - o;
                    need_norm = true;
                    stc(F_DL2, sqdist(o, lp));
                    phase = PH_SHADOW;
                    if (STATS) cnt.v[C_SHADOW]++;
                }
                if (gen) {
                    double xo = 0.0, yo = 0.0;
                    if (!(A.flags & RTM_FLAG_ZERO_JITTER)) {
                        xo = u01(ra[0], ra[1]) * (0.5 - -0.5) + -0.5;
                        yo = u01(ra[2], ra[3]) * (0.5 - -0.5) + -0.5;
                    }
                    const uint32_t yy = A.row_begin + ly * A.row_stride;
                    const double u = ((double)x + xo - (double)A.W * 0.5) / (double)A.W;   // main.rs:165
                    const double v = ((double)yy + yo - (double)A.H * 0.5) / (double)A.H;  // main.rs:166
                    newdir = ld3(S.cam_fwd) * S.cam_zoom + ld3(S.cam_up) * v + ld3(S.cam_right) * u;  // camera.rs:47
                    need_norm = true;
                    o = ld3(S.cam_pos);
                    stc3(F_COLOR, mk(0, 0, 0));
                    stc(F_ATT, 1.0);
                    bounce = 0;
                    phase = PH_PATH;
                }
                {
                    const V3 nd = normalize(newdir);  // vec3.rs:213-215: three divisions by the length
                    d = sel3(need_norm, nd, newdir);
                }
                RTW_MARK(3);
                if (st != ST_FETCH) {
                    // ---- new query: analytic primitives (core.rs:183-210)
                    if (STATS && phase == PH_PATH) cnt.v[bounce == 0 ? C_CAM : C_BOUNCE]++;
                    best = __builtin_inf();
                    found = false;
                    h_obj = kNoId; h_tri = kNoId; h_kp = 0;
                    const double a = dot(d, d);
                    for (uint32_t i = 0; i < S.n_spheres; ++i) {
                        if (STATS) cnt.v[C_SPHERE]++;
                        const double* sp = S.spheres + (size_t)i * 4;
                        const V3 oc = o - ld3(sp);
                        const double b = 2.0 * dot(d, oc);
                        const double c = dot(oc, oc) - sp[3] * sp[3];
                        const double disc = b * b - 4.0 * a * c;
                        if (disc < 0.0) continue;
                        const double t = (-b - __builtin_sqrt(disc)) / (2.0 * a);
                        consider(t, S.sph_ids[i * 2 + 1], S.sph_ids[i * 2], i, kNoId);
                    }
                    for (uint32_t i = 0; i < S.n_planes; ++i) {
                        if (STATS) cnt.v[C_PLANE]++;
                        const double* pp = S.planes + (size_t)i * 4;
                        const V3 pn = ld3(pp);
                        const double dv = dot(pn, d);
                        if (fabs(dv) < kEps) continue;
                        const V3 center = pn * pp[3];
                        const double t = dot(center - o, pn) / dv;
                        if (t < kEps) continue;
                        consider(t, S.pl_ids[i * 2 + 1], S.pl_ids[i * 2], (1u << 30) | i, kNoId);
                    }
                    model_i = 0;
                    st = ST_ENTER;
                }
            }
            RTW_MARK(4);
            // ---- enter the next model whose box the ray meets (box3D.rs:64-93, core.rs:95-125); m is wave-uniform
            for (uint32_t m = 0; m < S.n_models; ++m) {
                const RtmDevModel& M = S.models[m];
                if (st == ST_ENTER && model_i == m) {
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
                        ix = cell_coord((P0.x - M.bmin[0]) / M.cell[0], n);
                        iy = cell_coord((P0.y - M.bmin[1]) / M.cell[1], n);
                        iz = cell_coord((P0.z - M.bmin[2]) / M.cell[2], n);
                        const int sx = d.x > 0.0 ? 1 : -1, sy = d.y > 0.0 ? 1 : -1, sz = d.z > 0.0 ? 1 : -1;
                        const double nbx = (M.bmin[0] + M.hc2[0] * ((double)(ix + (sx > 0 ? 1 : 0)) + 0.5)) - M.hc[0];
                        const double nby = (M.bmin[1] + M.hc2[1] * ((double)(iy + (sy > 0 ? 1 : 0)) + 0.5)) - M.hc[1];
                        const double nbz = (M.bmin[2] + M.hc2[2] * ((double)(iz + (sz > 0 ? 1 : 0)) + 0.5)) - M.hc[2];
                        tmx = (nbx - P0.x) / d.x; tmy = (nby - P0.y) / d.y; tmz = (nbz - P0.z) / d.z;
                        tdx = ((double)sx * M.cell[0]) / d.x;
                        tdy = ((double)sy * M.cell[1]) / d.y;
                        tdz = ((double)sz * M.cell[2]) / d.z;
                        have = false; fresh = true; open = false;
                        st = ST_TRAV;
                    }
                }
            }
            if (st == ST_ENTER) st = ST_CONSUME;  // no model left: the query is complete
            RTW_MARK(5);
            if (STATS) dg_s_cyc += __builtin_amdgcn_s_memtime() - dg_t0;
        } else {
            // =========================== T loop ===========================
            // lanes traversing the same model as the first traversing lane step together (mm is wave-uniform)
            if (STATS) dg_t0 = __builtin_amdgcn_s_memtime();
            const int first = __ffsll((long long)traversing) - 1;
            const uint32_t mm = (uint32_t)__builtin_amdgcn_readlane((int)model_i, first);
            uint32_t occ_base = 0;  // LDS word offset of this model's bitmap, or 0xFFFFFFFF when it stayed in HBM
            for (uint32_t j = 0; j <= mm; ++j) {
                const uint32_t nj = S.models[j].n;
                const uint32_t words = (nj * nj * nj + 31u) >> 5;
                if (occ_base + words > A.occ_words) { occ_base = 0xFFFFFFFFu; break; }
                if (j < mm) occ_base += words;
            }
            const RtmDevModel& M = S.models[mm];
            const uint32_t n = M.n;
            const gptr<uint32_t> cell_start = as_global(M.cell_start), tri_idx = as_global(M.tri_idx);
            const gptr<double> tri_pos = as_global(M.tri_pos);
            const gptr<uint32_t> occ_glb = as_global(M.occupancy);
            const bool occ_in_lds = occ_base != 0xFFFFFFFFu;
            // a ray leaving the last model has completed its query: it needs no further model-enter pass
            const uint32_t after_model = mm + 1u < S.n_models ? ST_ENTER : ST_CONSUME;
            for (;;) {
                // A traversing lane is either WALKING (looking for the next non-empty cell; one DDA step and one
                // bitmap probe per iteration) or TESTING (its cell's list is open).  The two activities have very
                // different lengths per ray -- a few lanes cross dozens of empty cells while the others have lists to
                // scan -- so each loop iteration runs ONE of them, whichever has the lanes, instead of making every
                // lane wait for the longest walk of the wave.
                bool mine = st == ST_TRAV && model_i == mm;
                if (mine && !open && have) {  // core.rs:147-149: first cell with any accepted hit ends the walk
                    if (STATS) cnt.v[C_MHIT]++;
                    consider(m_t, M.obj, M.mat, (2u << 30) | mm, m_tri);
                    model_i = mm + 1;
                    st = after_model;
                    mine = false;
                }
                const uint32_t nW = (uint32_t)__popcll(__ballot(mine && !open));
                const uint32_t nT = (uint32_t)__popcll(__ballot(mine && open));
                if (nT == 0u || (nW != 0u && nT * 64u < A.t_test_frac * (nW + nT))) {
                    if (STATS) { dg_walk_iters++; dg_walk_lanes += nW; }
#pragma unroll
                    for (int u = 0; u < RTM_WALK_UNROLL; ++u) {
                        if (st == ST_TRAV && model_i == mm && !open) {
                            if (!fresh) RTW_DDA_ADVANCE();  // core.rs:151-167
                            fresh = false;
                            if ((unsigned)ix < n && (unsigned)iy < n && (unsigned)iz < n) {  // core.rs:129-134
                                if (STATS) cnt.v[C_CELLS]++;
                                const uint32_t cell = ((uint32_t)iz * n + (uint32_t)iy) * n + (uint32_t)ix;
                                uint32_t word;
                                if (occ_in_lds) word = occ_lds[occ_base + (cell >> 5)];
                                else word = occ_glb[cell >> 5];
                                if ((word >> (cell & 31u)) & 1u) {  // non-empty: the list is [k, k_end), k < k_end
                                    k = cell_start[cell];
                                    k_end = cell_start[cell + 1];
                                    open = true;
                                }
                            } else {  // left the grid without a hit
                                model_i = mm + 1;
                                st = after_model;
                            }
                        }
                    }
                } else {
                    if (STATS) { dg_t_steps++; dg_t_lanes += nT; }
                    if (mine && open) {
                        // Up to RTM_TRI_WIDE consecutive list entries per step, evaluated branch-free (model.rs:45-72):
                        // the reference's early returns become one predicate built from the same comparisons (so
                        // NaNs take the same side), which removes four exec-mask regions per test and gives the
                        // scheduler independent f64 chains.  Entries are applied in list order (core.rs:140-142).
                        const uint32_t left = k_end - k;
                        const uint32_t cntw = left < (uint32_t)RTM_TRI_WIDE ? left : (uint32_t)RTM_TRI_WIDE;
                        uint32_t ti[RTM_TRI_WIDE];
                        double tt[RTM_TRI_WIDE];
                        bool ok[RTM_TRI_WIDE];
#pragma unroll
                        for (int j = 0; j < RTM_TRI_WIDE; ++j) ti[j] = tri_idx[k + ((uint32_t)j < cntw ? (uint32_t)j : 0u)];
#pragma unroll
                        for (int j = 0; j < RTM_TRI_WIDE; ++j) {
                            tt[j] = 0.0;
                            ok[j] = false;
                            // slots beyond the second are evaluated only when some lane of the wave has that many
                            // entries left in its cell (wave-uniform branch: lists along a ray are mostly short)
                            const unsigned long long wants = __ballot((uint32_t)j < cntw);
                            if (j >= 2 && !wants) continue;
                            if (STATS) { dg_slots++; dg_slot_lanes += (uint64_t)__popcll(wants); }
                            const gptr<double> tp = tri_pos + (size_t)ti[j] * 9;
                            const V3 v0 = ld3(tp), e1 = ld3(tp + 3), e2 = ld3(tp + 6);
                            const V3 p = cross(d, e2);
                            const double det = dot(e1, p);
                            const double inv = 1.0 / det;
                            const V3 tv = o - v0;
                            const double u = dot(tv, p) * inv;
                            const V3 q = cross(tv, e1);
                            const double v = dot(d, q) * inv;
                            tt[j] = dot(e2, q) * inv;
                            ok[j] = (uint32_t)j < cntw && !(fabs(det) < kF64Eps) && !(u < 0.0 || u > 1.0) && !(v < 0.0 || u + v > 1.0) &&
                                    !(tt[j] < 0.0);
                        }
#pragma unroll
                        for (int j = 0; j < RTM_TRI_WIDE; ++j) {
                            const bool take = ok[j] && (!have || tt[j] < m_t);
                            m_t = take ? tt[j] : m_t;
                            m_tri = take ? ti[j] : m_tri;
                            have = have || take;
                        }
                        if (STATS) cnt.v[C_TRI] += cntw;
                        k += cntw;
                        open = k < k_end;
                    }
                }
                const unsigned long long still = __ballot(st == ST_TRAV && model_i == mm);
                if (!still) break;
                const unsigned long long w2 = __ballot(st == ST_CONSUME || st == ST_ENTER);
                if ((uint32_t)__popcll(still) * 64u < A.t_min_lanes * live && (uint32_t)__popcll(w2) >= s_need) break;
            }
            if (STATS) dg_t_cyc += __builtin_amdgcn_s_memtime() - dg_t0;
        }
    }
    if (STATS) {
        for (int i = 0; i < 9; ++i) {
            uint64_t v = cnt.v[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t lo = __shfl_xor((uint32_t)v, off, 64);
                const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
                v += ((uint64_t)hi << 32) | lo;
            }
            if (lane == 0 && v) atomicAdd(out_cnt + i, (unsigned long long)v);
        }
        if (lane == 0) {
            atomicAdd(out_cnt + 9, (unsigned long long)dg_s_rounds);
            atomicAdd(out_cnt + 10, (unsigned long long)dg_s_lanes);
            atomicAdd(out_cnt + 11, (unsigned long long)dg_t_steps);
            atomicAdd(out_cnt + 12, (unsigned long long)dg_t_lanes);
            atomicAdd(out_cnt + 13, (unsigned long long)dg_s_cyc);
            atomicAdd(out_cnt + 14, (unsigned long long)dg_t_cyc);
            atomicAdd(out_cnt + 15, (unsigned long long)(__builtin_amdgcn_s_memtime() - dg_begin));
        }
        if (lane == 0) {  // wave-uniform traversal diagnostics
            uint64_t dv[5] = {dg_walk_iters, dg_walk_lanes, dg_slots, dg_slot_lanes, dg_walk_steps};
            for (int i = 0; i < 5; ++i)
                if (dv[i]) atomicAdd(out_cnt + 16 + i, (unsigned long long)dv[i]);
            for (int i = 0; i < 6; ++i) atomicAdd(out_cnt + 24 + i, (unsigned long long)dg_sb[i]);
        }
    }
}

}  // namespace rtw
