// rt_multi.h -- wave-scheduled path-state machine with THREE queries in flight per lane.
//
// rt_wave.h keeps one ray per lane.  Its measured limits (DESIGN.md "Kernel", RTM_DIAG): a wave's lanes split into
// three populations -- set-up/shading, cell walking, triangle testing -- so each block of code runs at 25-45 % lane
// utilisation, and every path vertex costs three set-up rounds (hit, shadow ray 1, shadow ray 2).
//
// The reference's cast() (main.rs:57-132) fixes the ORDER in which results are combined, not the order in which the
// rays are traced: at a path vertex the shadow rays (main.rs:80-104) and the bounce ray (main.rs:50-55,121-125) all
// start at the hit point and depend only on the vertex (the random numbers are a pure function of pixel, sample and
// bounce).  So each lane owns three query SLOTS
//     slot 0: the camera / bounce query of the path,   slots 1,2: shadow queries towards lights li, li+1
// which are issued together and consumed together, in the reference's order: the diffuse sum over the lights in
// light order, then the surface term and the attenuation, then the next vertex.  Per vertex there is ONE set-up
// round instead of three, and in the traversal loop a lane offers whichever of its slots is in the state the wave
// is executing (walk or test), which roughly doubles the lanes per issued instruction.
// The arithmetic of every block is the op sequence of rt_device.h / the reference: bit-identical results.
#pragma once
#include "rt_wave.h"

#pragma clang fp contract(off)

namespace rtq {
using namespace rtd;

#ifndef RTM_MULTI_OCC
#define RTM_MULTI_OCC 2  // waves per SIMD the register allocator must leave room for
#endif
#ifndef RTM_MWALK_UNROLL
#define RTM_MWALK_UNROLL 4  // DDA steps per walk iteration
#endif
#ifndef RTM_MTRI_WIDE
#define RTM_MTRI_WIDE 4  // list entries tested per test iteration
#endif
constexpr int kBlock = 256;
constexpr int kColdFields = 15;                       // doubles of S-only state per lane
constexpr int kColdBytes = kColdFields * kBlock * 8;  // 30,720 B per workgroup
enum : int { F_PX = 0, F_COLOR = 3, F_ATT = 6, F_NRM = 7, F_DIFF = 10, F_DL2 = 13 /* two: slots 1 and 2 */ };
enum : uint32_t { L_FETCH = 0, L_ACTIVE = 1, L_DEAD = 2 };
// slot flag word: state, per-query flags, DDA step signs (core.rs:102-104), model index in the high half
enum : uint32_t {
    SL_IDLE = 0, SL_ENTER = 1, SL_TRAV = 2, SL_ST_MASK = 3,
    SL_HAVE = 4,    // a triangle of the current cell was accepted (core.rs:140-149)
    SL_FRESH = 8,   // the entry cell has not been visited yet: no DDA step before it
    SL_OPEN = 16,   // the current cell's list [k, k_end) is not exhausted
    SL_FOUND = 32,  // the query has a closest hit so far (core.rs:79-83)
    SL_SX = 64, SL_SY = 128, SL_SZ = 256,
    SL_MODEL_SHIFT = 16
};

struct Slot {
    V3 d;
    double best, m_t;
    double tmx, tmy, tmz, tdx, tdy, tdz;
    int ix, iy, iz;
    uint32_t k, k_end, m_tri, fl;
};

template <bool STATS>
__global__ __launch_bounds__(kBlock, RTM_MULTI_OCC) void render_multi_kernel(RtmDevScene S, rtw::RenderArgsW A, double* __restrict__ out_rgb,
                                                                              uint32_t* __restrict__ out_obj, uint32_t* __restrict__ out_tri,
                                                                              unsigned long long* __restrict__ out_cnt) {
    extern __shared__ double lds_raw[];
    double* const cold = lds_raw;                                                            // [kColdFields][kBlock]
    uint32_t* const occ_lds = reinterpret_cast<uint32_t*>(lds_raw + kColdFields * kBlock);  // [A.occ_words]
    const uint32_t tid = threadIdx.x;
    auto ldc = [&](int f) -> double { return cold[f * kBlock + tid]; };
    auto stc = [&](int f, double v) { cold[f * kBlock + tid] = v; };
    auto ldc3 = [&](int f) -> V3 { return mk(ldc(f), ldc(f + 1), ldc(f + 2)); };
    auto stc3 = [&](int f, V3 v) { stc(f, v.x); stc(f + 1, v.y); stc(f + 2, v.z); };

    {  // stage the occupancy bitmaps (models back to back; those that do not fit stay in HBM)
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
    const uint32_t nL = S.n_lights;

    // ---- lane state ---------------------------------------------------------------------------------
    uint32_t lst = L_FETCH;
    bool start = false;  // a freshly fetched pixel: nothing to consume yet
    bool hasv = false;   // a shaded vertex waits for its shadow results (n_sh of them, lights li..li+n_sh-1)
    uint32_t x = 0, ly = 0, pixel = 0, s = 0, li = 0, n_sh = 0, mat = 0;
    int bounce = 0;
    V3 o = mk(0, 0, 0);  // origin shared by the three slots: the camera position or the vertex being shaded
    uint32_t obj0 = kNoId, tri0 = kNoId;
    uint32_t h_obj = kNoId, h_tri = kNoId, h_mat = 0, h_kp = 0;  // slot 0's closest hit; h_kp = kind << 30 | prim
    Slot s0, s1, s2;
    s0.d = s1.d = s2.d = mk(0, 0, 1);
    s0.best = s1.best = s2.best = 0.0;
    s0.m_t = s1.m_t = s2.m_t = 0.0;
    s0.tmx = s0.tmy = s0.tmz = s0.tdx = s0.tdy = s0.tdz = 0.0;
    s1.tmx = s1.tmy = s1.tmz = s1.tdx = s1.tdy = s1.tdz = 0.0;
    s2.tmx = s2.tmy = s2.tmz = s2.tdx = s2.tdy = s2.tdz = 0.0;
    s0.ix = s0.iy = s0.iz = s1.ix = s1.iy = s1.iz = s2.ix = s2.iy = s2.iz = 0;
    s0.k = s0.k_end = s0.m_tri = s1.k = s1.k_end = s1.m_tri = s2.k = s2.k_end = s2.m_tri = 0;
    s0.fl = s1.fl = s2.fl = SL_IDLE;

    // core.rs:75-83.  Ties go to the earlier object (core.rs:79 is strict and objects are visited in file order);
    // only slot 0 keeps the identity of its hit -- a shadow query needs the distance alone (main.rs:88-92).
    auto consider = [&](Slot& q, const bool primary, double t, uint32_t obj, uint32_t mt, uint32_t kp, uint32_t tri) __attribute__((always_inline)) {
        if (t <= kEps) return;
        const bool fnd = (q.fl & SL_FOUND) != 0u;
        if (t < q.best || (primary && fnd && t == q.best && obj < h_obj)) {
            q.best = t;
            q.fl |= SL_FOUND;
            if (primary) { h_obj = obj; h_mat = mt; h_kp = kp; h_tri = tri; }
        }
    };
    // a new query in slot q: analytic primitives at once (core.rs:183-210), models through the enter pass
    auto issue = [&](Slot& q, const bool primary, const bool go, const V3 dir) __attribute__((always_inline)) {
        if (go) {
            q.d = dir;
            q.best = __builtin_inf();
            q.fl = SL_ENTER;  // model 0, no flags
            if (primary) { h_obj = kNoId; h_tri = kNoId; h_kp = 0; }
            const double a = dot(dir, dir);
            for (uint32_t i = 0; i < S.n_spheres; ++i) {
                if (STATS) cnt.v[C_SPHERE]++;
                const double* sp = S.spheres + (size_t)i * 4;
                const V3 oc = o - ld3(sp);
                const double b = 2.0 * dot(dir, oc);
                const double c = dot(oc, oc) - sp[3] * sp[3];
                const double disc = b * b - 4.0 * a * c;
                if (disc < 0.0) continue;
                const double t = (-b - __builtin_sqrt(disc)) / (2.0 * a);
                consider(q, primary, t, S.sph_ids[i * 2 + 1], S.sph_ids[i * 2], i, kNoId);
            }
            for (uint32_t i = 0; i < S.n_planes; ++i) {
                if (STATS) cnt.v[C_PLANE]++;
                const double* pp = S.planes + (size_t)i * 4;
                const V3 pn = ld3(pp);
                const double dv = dot(pn, dir);
                if (fabs(dv) < kEps) continue;
                const V3 center = pn * pp[3];
                const double t = dot(center - o, pn) / dv;
                if (t < kEps) continue;
                consider(q, primary, t, S.pl_ids[i * 2 + 1], S.pl_ids[i * 2], (1u << 30) | i, kNoId);
            }
        }
    };
    // slab test and DDA initialisation against model m (box3D.rs:64-93, core.rs:95-125)
    auto enter = [&](Slot& q, const RtmDevModel& M, const uint32_t m) __attribute__((always_inline)) {
        if ((q.fl & SL_ST_MASK) == SL_ENTER && (q.fl >> SL_MODEL_SHIFT) == m) {
            if (STATS) cnt.v[C_BBOX]++;
            const V3 d = q.d;
            const double fx = 1.0 / d.x, fy = 1.0 / d.y, fz = 1.0 / d.z;
            const double t1 = (M.bmin[0] - o.x) * fx, t2 = (M.bmax[0] - o.x) * fx;
            const double t3 = (M.bmin[1] - o.y) * fy, t4 = (M.bmax[1] - o.y) * fy;
            const double t5 = (M.bmin[2] - o.z) * fz, t6 = (M.bmax[2] - o.z) * fz;
            const double tmin = fmax(fmax(fmin(t1, t2), fmin(t3, t4)), fmin(t5, t6));
            const double tmax = fmin(fmin(fmax(t1, t2), fmax(t3, t4)), fmax(t5, t6));
            if (tmax < 0.0 || tmin > tmax) {
                q.fl += 1u << SL_MODEL_SHIFT;
            } else {
                const V3 P0 = ray_at(o, d, tmin);
                const int n = (int)M.n;
                q.ix = cell_coord((P0.x - M.bmin[0]) / M.cell[0], n);
                q.iy = cell_coord((P0.y - M.bmin[1]) / M.cell[1], n);
                q.iz = cell_coord((P0.z - M.bmin[2]) / M.cell[2], n);
                const int sx = d.x > 0.0 ? 1 : -1, sy = d.y > 0.0 ? 1 : -1, sz = d.z > 0.0 ? 1 : -1;
                const double nbx = (M.bmin[0] + M.hc2[0] * ((double)(q.ix + (sx > 0 ? 1 : 0)) + 0.5)) - M.hc[0];
                const double nby = (M.bmin[1] + M.hc2[1] * ((double)(q.iy + (sy > 0 ? 1 : 0)) + 0.5)) - M.hc[1];
                const double nbz = (M.bmin[2] + M.hc2[2] * ((double)(q.iz + (sz > 0 ? 1 : 0)) + 0.5)) - M.hc[2];
                q.tmx = (nbx - P0.x) / d.x; q.tmy = (nby - P0.y) / d.y; q.tmz = (nbz - P0.z) / d.z;
                q.tdx = ((double)sx * M.cell[0]) / d.x;
                q.tdy = ((double)sy * M.cell[1]) / d.y;
                q.tdz = ((double)sz * M.cell[2]) / d.z;
                q.fl = (q.fl & SL_FOUND) | SL_TRAV | SL_FRESH | (sx > 0 ? SL_SX : 0u) | (sy > 0 ? SL_SY : 0u) | (sz > 0 ? SL_SZ : 0u) |
                       (m << SL_MODEL_SHIFT);
            }
        }
    };

    uint64_t dg_s_rounds = 0, dg_s_lanes = 0, dg_t_steps = 0, dg_t_lanes = 0;  // scheduler diagnostics (STATS build only)
    uint64_t dg_s_cyc = 0, dg_t_cyc = 0, dg_t0 = 0, dg_walk_iters = 0, dg_walk_lanes = 0;
    const uint64_t dg_begin = STATS ? __builtin_amdgcn_s_memtime() : 0;
    for (;;) {
        // ---- hand out pixels to idle lanes
        for (;;) {
            const unsigned long long need = __ballot(lst == L_FETCH);
            if (!need) break;
            if (pool_next == pool_end) {
                const int first = __ffsll((long long)need) - 1;
                uint32_t base = 0;
                if (lane == (uint32_t)first) base = atomicAdd(A.work_counter, 1u);
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                if (base >= A.n_tiles) {  // frame exhausted
                    if (lst == L_FETCH) lst = L_DEAD;
                    break;
                }
                pool_next = base * 64u;
                pool_end = pool_next + 64u;
            }
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const uint32_t avail = pool_end - pool_next;
            if (lst == L_FETCH && rank < avail) {
                const uint32_t id = pool_next + rank, tile = id >> 6, in = id & 63u;
                x = (tile % A.tiles_x) * 8u + (in & 7u);
                ly = (tile / A.tiles_x) * 8u + (in >> 3);
                if (x < A.W && ly < A.n_rows) {  // ids outside a ragged right/bottom tile are skipped
                    pixel = x + A.W * (A.row_begin + ly * A.row_stride);  // main.rs:160-161,177
                    stc3(F_PX, mk(0, 0, 0));
                    s = 0;
                    start = true;
                    hasv = false;
                    lst = L_ACTIVE;
                }
            }
            const uint32_t want = (uint32_t)__popcll(need);
            pool_next += want < avail ? want : avail;
        }
        const uint32_t st0 = s0.fl & SL_ST_MASK, st1 = s1.fl & SL_ST_MASK, st2 = s2.fl & SL_ST_MASK;
        const bool active = lst == L_ACTIVE;
        const bool all_idle = st0 == SL_IDLE && st1 == SL_IDLE && st2 == SL_IDLE;
        const bool any_enter = st0 == SL_ENTER || st1 == SL_ENTER || st2 == SL_ENTER;
        const bool any_trav = st0 == SL_TRAV || st1 == SL_TRAV || st2 == SL_TRAV;
        const unsigned long long waiting = __ballot(active && (all_idle || any_enter));
        const unsigned long long traversing = __ballot(active && any_trav);
        if (!waiting && !traversing) break;
        const uint32_t live = (uint32_t)__popcll(__ballot(active));
        const uint32_t s_need = (A.s_threshold * live + 63u) >> 6;
        if (!traversing || (uint32_t)__popcll(waiting) >= s_need) {
            // =========================== S round ===========================
            if (STATS) { dg_s_rounds++; dg_s_lanes += (uint64_t)__popcll(waiting); dg_t0 = __builtin_amdgcn_s_memtime(); }
            if (active && all_idle) {
                // ---- 0. decisions (integer state only); every expensive block below exists once
                const bool a_v = hasv;
                const uint32_t li0 = li;                      // first light of the batch whose results are in
                const uint32_t n_in = a_v ? n_sh : 0u;        // shadow results to consume
                const bool a_more = a_v && li0 + n_in < nL;   // further lights: another batch, slot 0 keeps its result
                const bool a_fin = a_v && !a_more;            // the vertex is complete (main.rs:107-125)
                const bool a_end = a_fin && bounce + 1 >= K;  // ... and it was the last one allowed
                const bool b_take = !start && !a_more && !a_end;  // slot 0 holds the next path query's result
                const bool found0 = (s0.fl & SL_FOUND) != 0u;
                const bool b_hit = b_take && found0, b_miss = b_take && !found0;
                const bool end_sample = a_end || b_miss;
                const bool last_sample = end_sample && s + 1u >= A.spp;
                const bool gen = start || (end_sample && !last_sample);
                const bool metal_old = S.materials[(size_t)mat * 4 + 3] != 0.0;
                const uint32_t mat_new = b_hit ? h_mat : mat;
                const bool metal_new = S.materials[(size_t)mat_new * 4 + 3] != 0.0;
                const int bounce_new = bounce + (a_fin ? 1 : 0);  // index of the vertex a hit creates
                const bool plastic_hit = b_hit && !metal_new;
                // ---- 1. random numbers: Vec3::random() of this bounce (vec3.rs:160-162) or the next sample's jitter
                uint32_t ra[4] = {0, 0, 0, 0}, rb[4] = {0, 0, 0, 0};
                if (plastic_hit || gen) {
                    const uint32_t smp = gen ? (start ? s : s + 1u) : s;
                    philox(A.seed, pixel, smp, gen ? 0u : 1u + 2u * (uint32_t)bounce_new, ra);
                }
                if (plastic_hit) philox(A.seed, pixel, s, 2u + 2u * (uint32_t)bounce_new, rb);
                // ---- 2. shadow results of the pending vertex, in light order (main.rs:86-104); o == P
                if (n_in > 0u) {
                    V3 diff = ldc3(F_DIFF);
                    const V3 nn = ldc3(F_NRM);
                    auto shade = [&](const Slot& q, const uint32_t j) __attribute__((always_inline)) {
                        if (j < n_in) {
                            const double* L = S.lights + (size_t)(li0 + j) * 8;
                            const double dl2 = ldc(F_DL2 + (int)j);
                            bool occluded = false;
                            if (q.fl & SL_FOUND) occluded = sqdist(ray_at(o, q.d, q.best), o) <= dl2;
                            if (!occluded) {
                                const double eff = dot(q.d, nn);
                                if (eff > 0.001) {
                                    const double dd = fmax(dl2 / (L[6] * L[6]), 1.0);
                                    const double decay = 1.0 / dd;
                                    diff = diff + ld3(L + 3) * decay * eff;
                                }
                            }
                        }
                    };
                    shade(s1, 0u);
                    shade(s2, 1u);
                    stc3(F_DIFF, diff);
                    li = li0 + n_in;
                }
                // ---- 3. end of the light loop: the surface's colour, attenuation (main.rs:107-125)
                if (a_fin) {
                    const double* mp = S.materials + (size_t)mat * 4;
                    const double att = ldc(F_ATT);
                    stc3(F_COLOR, ldc3(F_COLOR) + ld3(mp) * att * (ldc3(F_DIFF) * (metal_old ? 0.2 : 0.9)));
                    stc(F_ATT, att * (metal_old ? 0.8 : 0.1));
                    ++bounce;
                    hasv = false;
                }
                // ---- 4. the path query's result
                if (b_take && bounce == 0 && s == 0) { obj0 = found0 ? h_obj : kNoId; tri0 = found0 ? h_tri : kNoId; }
                if (b_miss) stc3(F_COLOR, ldc3(F_COLOR) + mk(1, 1, 1) * ldc(F_ATT));  // main.rs:127
                V3 dir0 = mk(0, 0, 1);
                bool norm0 = false;
                if (b_hit) {
                    // surface point and shading normal (core.rs:89,169-177,195,209; main.rs:68-73)
                    const V3 d = s0.d;
                    const V3 P = ray_at(o, d, s0.best);
                    const uint32_t kind = h_kp >> 30, prim = h_kp & 0x3FFFFFFFu;
                    V3 raw;
                    if (kind == 0u) raw = ray_at(o, d, s0.best) - ld3(S.spheres + (size_t)prim * 4);
                    else if (kind == 1u) raw = ld3(S.planes + (size_t)prim * 4);  // used as written, not normalised
                    else raw = model_normal_raw(S.models[prim], h_tri, P);
                    V3 nn = normalize(raw);
                    if (kind == 1u) nn = raw;
                    if (dot(d, nn) > 0.001) nn = nn * -1.0;
                    // bounce direction (main.rs:50-55): metal reflects (not renormalised, vec3.rs:172-175); plastic is
                    // normalize(n + random()), normalised in step 6
                    const V3 rnd = mk(u01(ra[0], ra[1]) * (1.0 - 0.0) + 0.0, u01(ra[2], ra[3]) * (1.0 - 0.0) + 0.0,
                                      u01(rb[0], rb[1]) * (1.0 - 0.0) + 0.0);
                    const V3 refl = d - nn * 2.0 * dot(d, nn);
                    const V3 dif = nn + rnd;
                    stc3(F_NRM, nn);
                    stc3(F_DIFF, mk(0, 0, 0));
                    dir0 = metal_new ? refl : dif;
                    norm0 = !metal_new;
                    mat = mat_new;
                    o = P;  // the shadow rays start here and so does the bounce (main.rs:82,121)
                    li = 0;
                    hasv = true;
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
                        lst = L_FETCH;
                    } else {
                        stc3(F_PX, px);
                    }
                }
                if (gen) {  // camera ray of the next sample (main.rs:160-168)
                    double xo = 0.0, yo = 0.0;
                    if (!(A.flags & RTM_FLAG_ZERO_JITTER)) {
                        xo = u01(ra[0], ra[1]) * (0.5 - -0.5) + -0.5;
                        yo = u01(ra[2], ra[3]) * (0.5 - -0.5) + -0.5;
                    }
                    const uint32_t yy = A.row_begin + ly * A.row_stride;
                    const double u = ((double)x + xo - (double)A.W * 0.5) / (double)A.W;   // main.rs:165
                    const double v = ((double)yy + yo - (double)A.H * 0.5) / (double)A.H;  // main.rs:166
                    dir0 = ld3(S.cam_fwd) * S.cam_zoom + ld3(S.cam_up) * v + ld3(S.cam_right) * u;  // camera.rs:47
                    norm0 = true;
                    o = ld3(S.cam_pos);
                    stc3(F_COLOR, mk(0, 0, 0));
                    stc(F_ATT, 1.0);
                    bounce = 0;
                    start = false;
                    hasv = false;
                }
                // ---- 6. the new queries: slot 0 = camera ray or bounce ray, slots 1,2 = shadow rays to lights li, li+1
                const bool go0 = gen || (b_hit && bounce + 1 < K);
                const bool sh = b_hit || a_more;
                n_sh = sh ? (nL - li < 2u ? nL - li : 2u) : 0u;
                const bool go1 = n_sh > 0u, go2 = n_sh > 1u;
                {
                    const V3 nd = normalize(dir0);  // vec3.rs:213-215: three divisions by the length
                    dir0 = norm0 ? nd : dir0;
                }
                if (STATS && go0) cnt.v[gen ? C_CAM : C_BOUNCE]++;
                if (STATS) cnt.v[C_SHADOW] += n_sh;
                issue(s0, true, go0, dir0);
                auto shadow_dir = [&](const bool go, const uint32_t j) __attribute__((always_inline)) -> V3 {
                    V3 dir = mk(0, 0, 1);
                    if (go) {
                        const V3 lp = ld3(S.lights + (size_t)(li + j) * 8);
                        dir = normalize(lp - o);  // main.rs:80-85
                        stc(F_DL2 + (int)j, sqdist(o, lp));
                    }
                    return dir;
                };
                issue(s1, false, go1, shadow_dir(go1, 0u));
                issue(s2, false, go2, shadow_dir(go2, 1u));
            }
            // ---- enter the next model whose box the ray meets; m is wave-uniform
            for (uint32_t m = 0; m < S.n_models; ++m) {
                const RtmDevModel& M = S.models[m];
                enter(s0, M, m);
                enter(s1, M, m);
                enter(s2, M, m);
            }
            // no model left: the query is complete
            if ((s0.fl & SL_ST_MASK) == SL_ENTER) s0.fl &= ~SL_ST_MASK;
            if ((s1.fl & SL_ST_MASK) == SL_ENTER) s1.fl &= ~SL_ST_MASK;
            if ((s2.fl & SL_ST_MASK) == SL_ENTER) s2.fl &= ~SL_ST_MASK;
            if (STATS) dg_s_cyc += __builtin_amdgcn_s_memtime() - dg_t0;
        } else {
            // =========================== T loop ===========================
            // slots traversing the same model as the first traversing slot step together (mm is wave-uniform)
            if (STATS) dg_t0 = __builtin_amdgcn_s_memtime();
            const int first = __ffsll((long long)traversing) - 1;
            const uint32_t my_m = st0 == SL_TRAV ? s0.fl >> SL_MODEL_SHIFT : (st1 == SL_TRAV ? s1.fl >> SL_MODEL_SHIFT : s2.fl >> SL_MODEL_SHIFT);
            const uint32_t mm = (uint32_t)__builtin_amdgcn_readlane((int)my_m, first);
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
            const uint32_t after_fl = (mm + 1u < S.n_models ? SL_ENTER : SL_IDLE) | ((mm + 1u) << SL_MODEL_SHIFT);
            const uint32_t trav_mm = SL_TRAV | (mm << SL_MODEL_SHIFT);
            constexpr uint32_t kWhere = SL_ST_MASK | (0xFFFFu << SL_MODEL_SHIFT);  // state and model bits of a flag word
            const uint32_t M_obj = M.obj, M_mat = M.mat;
            auto complete = [&](Slot& q, const bool primary) __attribute__((always_inline)) {
                // core.rs:147-149: the first cell with any accepted hit ends the walk
                if ((q.fl & (kWhere | SL_OPEN | SL_HAVE)) == (trav_mm | SL_HAVE)) {
                    if (STATS) cnt.v[C_MHIT]++;
                    consider(q, primary, q.m_t, M_obj, M_mat, (2u << 30) | mm, q.m_tri);
                    q.fl = (q.fl & SL_FOUND) | after_fl;
                }
            };
#define RTQ_PICK(f) (pick == 0u ? s0.f : (pick == 1u ? s1.f : s2.f))
#define RTQ_PUT(f, v)                        \
    do {                                     \
        s0.f = put0 ? (v) : s0.f;            \
        s1.f = put1 ? (v) : s1.f;            \
        s2.f = put2 ? (v) : s2.f;            \
    } while (0)
            for (;;) {
                complete(s0, true);
                complete(s1, false);
                complete(s2, false);
                // A traversing slot is either WALKING (one DDA step and one bitmap probe per cell) or TESTING (its
                // cell's list is open).  Each iteration runs ONE of the two activities, whichever has the lanes; a
                // lane takes part with the first of its slots that is in that state.
                const bool w0 = (s0.fl & (kWhere | SL_OPEN)) == trav_mm, w1 = (s1.fl & (kWhere | SL_OPEN)) == trav_mm,
                           w2 = (s2.fl & (kWhere | SL_OPEN)) == trav_mm;
                const bool t0 = (s0.fl & (kWhere | SL_OPEN)) == (trav_mm | SL_OPEN), t1 = (s1.fl & (kWhere | SL_OPEN)) == (trav_mm | SL_OPEN),
                           t2 = (s2.fl & (kWhere | SL_OPEN)) == (trav_mm | SL_OPEN);
                const bool anyW = w0 || w1 || w2, anyT = t0 || t1 || t2;
                const uint32_t nW = (uint32_t)__popcll(__ballot(anyW));
                const uint32_t nT = (uint32_t)__popcll(__ballot(anyT));
                if (nW + nT == 0u) break;
                if (nT == 0u || (nW != 0u && nT * 64u < A.t_test_frac * (nW + nT))) {
                    if (STATS) { dg_walk_iters++; dg_walk_lanes += nW; }
                    if (anyW) {
                        const uint32_t pick = w0 ? 0u : (w1 ? 1u : 2u);
                        const bool put0 = pick == 0u, put1 = pick == 1u, put2 = pick == 2u;
                        double tmx = RTQ_PICK(tmx), tmy = RTQ_PICK(tmy), tmz = RTQ_PICK(tmz);
                        const double tdx = RTQ_PICK(tdx), tdy = RTQ_PICK(tdy), tdz = RTQ_PICK(tdz);
                        int ix = RTQ_PICK(ix), iy = RTQ_PICK(iy), iz = RTQ_PICK(iz);
                        uint32_t fl = RTQ_PICK(fl), k = RTQ_PICK(k), k_end = RTQ_PICK(k_end);
#pragma unroll
                        for (int u = 0; u < RTM_MWALK_UNROLL; ++u) {
                            if ((fl & (SL_ST_MASK | SL_OPEN)) == SL_TRAV) {
                                if (!(fl & SL_FRESH)) {  // core.rs:151-167 with the step signs of core.rs:102-104, as selects
                                    const bool ax_ = (tmx < tmy) && (tmx < tmz);
                                    const bool ay_ = !(tmx < tmy) && (tmy < tmz);
                                    const bool az_ = !ax_ && !ay_;
                                    ix += ax_ ? ((fl & SL_SX) ? 1 : -1) : 0;
                                    iy += ay_ ? ((fl & SL_SY) ? 1 : -1) : 0;
                                    iz += az_ ? ((fl & SL_SZ) ? 1 : -1) : 0;
                                    const double nx_ = tmx + tdx, ny_ = tmy + tdy, nz_ = tmz + tdz;
                                    tmx = ax_ ? nx_ : tmx;
                                    tmy = ay_ ? ny_ : tmy;
                                    tmz = az_ ? nz_ : tmz;
                                }
                                fl &= ~SL_FRESH;
                                if ((unsigned)ix < n && (unsigned)iy < n && (unsigned)iz < n) {  // core.rs:129-134
                                    if (STATS) cnt.v[C_CELLS]++;
                                    const uint32_t cell = ((uint32_t)iz * n + (uint32_t)iy) * n + (uint32_t)ix;
                                    uint32_t word;
                                    if (occ_in_lds) word = occ_lds[occ_base + (cell >> 5)];
                                    else word = occ_glb[cell >> 5];
                                    if ((word >> (cell & 31u)) & 1u) {  // non-empty: the list is [k, k_end), k < k_end
                                        k = cell_start[cell];
                                        k_end = cell_start[cell + 1];
                                        fl |= SL_OPEN;
                                    }
                                } else {  // left the grid without a hit
                                    fl = (fl & SL_FOUND) | after_fl;
                                }
                            }
                        }
                        RTQ_PUT(tmx, tmx); RTQ_PUT(tmy, tmy); RTQ_PUT(tmz, tmz);
                        RTQ_PUT(ix, ix); RTQ_PUT(iy, iy); RTQ_PUT(iz, iz);
                        RTQ_PUT(fl, fl); RTQ_PUT(k, k); RTQ_PUT(k_end, k_end);
                    }
                } else {
                    if (STATS) { dg_t_steps++; dg_t_lanes += nT; }
                    if (anyT) {
                        // Up to RTM_MTRI_WIDE consecutive list entries, evaluated branch-free (model.rs:45-72): the
                        // reference's early returns become one predicate built from the same comparisons (so NaNs take
                        // the same side).  Entries are applied in list order (core.rs:140-142).
                        const uint32_t pick = t0 ? 0u : (t1 ? 1u : 2u);
                        const bool put0 = pick == 0u, put1 = pick == 1u, put2 = pick == 2u;
                        const V3 d = mk(RTQ_PICK(d.x), RTQ_PICK(d.y), RTQ_PICK(d.z));
                        uint32_t k = RTQ_PICK(k);
                        const uint32_t k_end = RTQ_PICK(k_end);
                        double m_t = RTQ_PICK(m_t);
                        uint32_t m_tri = RTQ_PICK(m_tri), fl = RTQ_PICK(fl);
                        bool have = (fl & SL_HAVE) != 0u;
                        const uint32_t left = k_end - k;
                        const uint32_t cntw = left < (uint32_t)RTM_MTRI_WIDE ? left : (uint32_t)RTM_MTRI_WIDE;
                        uint32_t ti[RTM_MTRI_WIDE];
                        double tt[RTM_MTRI_WIDE];
                        bool ok[RTM_MTRI_WIDE];
#pragma unroll
                        for (int j = 0; j < RTM_MTRI_WIDE; ++j) ti[j] = tri_idx[k + ((uint32_t)j < cntw ? (uint32_t)j : 0u)];
#pragma unroll
                        for (int j = 0; j < RTM_MTRI_WIDE; ++j) {
                            tt[j] = 0.0;
                            ok[j] = false;
                            // slots beyond the second are evaluated only when some lane of the wave has that many
                            // entries left in its cell (wave-uniform branch: lists along a ray are mostly short)
                            if (j >= 2 && !__ballot((uint32_t)j < cntw)) continue;
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
                        for (int j = 0; j < RTM_MTRI_WIDE; ++j) {
                            const bool take = ok[j] && (!have || tt[j] < m_t);
                            m_t = take ? tt[j] : m_t;
                            m_tri = take ? ti[j] : m_tri;
                            have = have || take;
                        }
                        if (STATS) cnt.v[C_TRI] += cntw;
                        k += cntw;
                        fl = (fl & ~(SL_HAVE | SL_OPEN)) | (have ? SL_HAVE : 0u) | (k < k_end ? SL_OPEN : 0u);
                        RTQ_PUT(k, k); RTQ_PUT(m_t, m_t); RTQ_PUT(m_tri, m_tri); RTQ_PUT(fl, fl);
                    }
                }
                const bool tr0 = (s0.fl & kWhere) == trav_mm, tr1 = (s1.fl & kWhere) == trav_mm, tr2 = (s2.fl & kWhere) == trav_mm;
                const unsigned long long still = __ballot(tr0 || tr1 || tr2);
                // (slots that just finished with a hit are completed at the top of the next iteration)
                const uint32_t e0 = s0.fl & SL_ST_MASK, e1 = s1.fl & SL_ST_MASK, e2 = s2.fl & SL_ST_MASK;
                const unsigned long long wt = __ballot(active && ((e0 == SL_IDLE && e1 == SL_IDLE && e2 == SL_IDLE) || e0 == SL_ENTER ||
                                                                   e1 == SL_ENTER || e2 == SL_ENTER));
                if (!still) break;
                if ((uint32_t)__popcll(still) * 64u < A.t_min_lanes * live && (uint32_t)__popcll(wt) >= s_need) break;
            }
#undef RTQ_PICK
#undef RTQ_PUT
            // hits accepted in the last iteration (the loop may leave before its next top)
            complete(s0, true);
            complete(s1, false);
            complete(s2, false);
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
            atomicAdd(out_cnt + 16, (unsigned long long)dg_walk_iters);
            atomicAdd(out_cnt + 17, (unsigned long long)dg_walk_lanes);
        }
    }
}

}  // namespace rtq
