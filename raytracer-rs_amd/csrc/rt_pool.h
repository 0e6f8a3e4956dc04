// rt_pool.h -- the trace kernel of the ray-front pipeline with its rays parked in LDS.
//
// rt_front.h's first trace kernel keeps one ray per lane in registers; a wave step is either a walk step or a test step, and
// the lanes whose ray is in the other activity (or idle, waiting for a refill) are masked off: 33 % of the lane slots of its
// walk steps and 54 % of its test slots do work on the bunny frame (RTM_DIAG, profiles/r03_diag_*).  Here every WAVE owns a
// pool of R > 64 rays whose whole state lives in LDS (structure of arrays over the ray slot: ~140 bytes per ray), and a step
// first PICKS up to 64 rays that are ready for the activity it runs -- walk-ready, test-ready, waiting to enter their next model,
// or idle slots to refill -- loads the half of their state that the activity needs, runs it with (nearly) all lanes, and
// stores what changed.  The pool is private to its wave: no cross-wave atomics, no barrier after the bitmaps are staged, the
// same bounded termination as before (a wave stops when its queues are exhausted and its pool is empty).
//   pick:  every lane reads the status word of ray `lane` (and of ray 64 + lane), four ballots turn them into per-activity
//          masks over ray slots; the lane whose ray is the r-th ready one writes its slot number to pick[r]; lane r reads it.
//   walk:  tMax, tDelta, position (cv, rem) in; up to kPoolWalkSteps DDA steps (core.rs:129-167), ending early at a
//          non-empty cell or the grid's boundary; tMax, position and the status out.
//   test:  origin, direction, window (k, k_end, mask), closest so far in; up to A.wide entries (model.rs:43-74); window and
//          closest out; a closed cell makes the ray walk-ready again or -- core.rs:147-149 -- ends its visit of the model.
// Which ray a lane works on changes from step to step; every ray's own sequence of operations is the one of rt_front.h
// (same expressions on the same operands), so results are bit-identical -- tests run both kernels against the oracle.
#pragma once
#include "rt_front.h"

#pragma clang fp contract(off)

namespace rtf {

constexpr int kPoolBlockMax = 1024;  // one workgroup per CU (the bitmaps are staged once per CU), as many waves as the pools leave LDS for
constexpr int kPoolWalkSteps = 4;   // DDA steps of a walk batch before the lanes still walking are counted
constexpr int kPoolWalkMore = 4;    // ... further pairs of steps while enough of them are left

// per-ray state in LDS, structure of arrays over the ray slot: doubles, then words
enum : int { PD_OX = 0, PD_OY, PD_OZ, PD_DX, PD_DY, PD_DZ, PD_TMX, PD_TMY, PD_TMZ, PD_TDX, PD_TDY, PD_TDZ, PD_MT, PD_BT };
enum : int { PW_CV = 0, PW_REM, PW_K, PW_KEND, PW_WMASK, PW_MTRI, PW_Q, PW_MISC, PW_FL, PW_EXTRA };
// PW_MISC: model (16 bits) | step signs at IM_SX.. (bits 16-18) | cells per side of the model (bits 19-28)
enum : uint32_t {
    PF_HAVE = 1u,     // a triangle of the current cell was accepted (core.rs:140-149)
    PF_OPEN = 4u,     // the current cell's list is not exhausted: the ray is test-ready
    PF_WINIT = 8u,    // the window of a newly opened cell has not been set up yet
    PF_LONG = 16u,    // the window belongs to a long cell and has not been screened yet
    PF_HI = 32u,      // the box screen also covered the 32 entries after the window
    PF_IDLE = 64u,    // the slot holds no ray
    PF_ENTER = 128u   // the ray has left a model and visits the next one (scenes with several models)
};
constexpr int pool_nd(bool multi) { return multi ? 14 : 13; }
constexpr int pool_nu(bool has_long, bool multi) { return (int)PW_EXTRA + (has_long ? 1 : 0) + (multi ? 2 : 0); }
constexpr size_t pool_wave_bytes(int R, bool has_long, bool multi) { return (size_t)R * (size_t)(pool_nd(multi) * 8 + pool_nu(has_long, multi) * 4) + 256u; }
constexpr size_t pool_lds_bytes(uint32_t occ_words, int R, bool has_long, bool multi, uint32_t waves) {
    return (((size_t)occ_words * 4 + 15) & ~(size_t)15) + (size_t)waves * pool_wave_bytes(R, has_long, multi);
}

template <bool STATS, bool HAS_LONG, bool MULTI, int R>
__global__ __launch_bounds__(kPoolBlockMax) void pool_trace_kernel(RtmDevScene S, FrontArgs A, uint32_t round, unsigned long long* __restrict__ out_cnt) {
    static_assert(R > 64 && R <= 128 && R % 8 == 0, "pool size");
    extern __shared__ __attribute__((aligned(16))) uint32_t pool_lds[];  // [A.occ_words] bitmaps, then one pool per wave
    __shared__ uint32_t mtab[kModelTab];                                  // per model: cells per side | first scene cell / 32 << 10
    constexpr int ND = pool_nd(MULTI), NU = pool_nu(HAS_LONG, MULTI);
    constexpr int PW_WHI = PW_EXTRA, PW_BM = PW_EXTRA + (HAS_LONG ? 1 : 0), PW_BTRI = PW_BM + 1;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave_in_wg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    for (uint32_t m = tid; m < min(S.n_models, (uint32_t)kModelTab); m += blockDim.x)
        mtab[m] = S.models[m].n | ((S.models[m].cell_base >> 5) << 10);
    {
        const gptr<uint32_t> src = as_global(S.g_occ7);
        for (uint32_t i = tid; i < A.occ_words; i += blockDim.x) pool_lds[i] = src[A.occ_plain ? i * 8u + 6u : i];
    }
    const uint32_t* const occ_lds = pool_lds;
    char* const pool_base = (char*)pool_lds + (((size_t)A.occ_words * 4 + 15) & ~(size_t)15) + (size_t)wave_in_wg * pool_wave_bytes(R, HAS_LONG, MULTI);
    double* const PD = (double*)pool_base;                    // PD[f * R + j]
    uint32_t* const PU = (uint32_t*)(pool_base + (size_t)R * ND * 8);  // PU[f * R + j]
    uint32_t* const PK = PU + (size_t)R * NU;                 // pick[64]
    for (uint32_t j = lane; j < (uint32_t)R; j += 64u) PU[PW_FL * R + j] = PF_IDLE;
    PK[lane] = 0u;
    __syncthreads();

    Counters cnt;
    if (STATS)
        for (int i = 0; i < 9; ++i) cnt.v[i] = 0;
    const uint64_t dg_begin = STATS ? __builtin_amdgcn_s_memtime() : 0;
    uint64_t dg_tr = 0, dg_tw = 0, dg_tt = 0, dg_t0 = 0;
    uint64_t dg_steps = 0, dg_step_lanes = 0, dg_walks = 0, dg_walk_lanes = 0, dg_slots = 0, dg_slot_lanes = 0, dg_refills = 0, dg_refill_lanes = 0;

    // the wave's position in the queue (as in front_trace_kernel: own sub-queues first, claimed A.claim items at a time, then
    // up to A.steal sub-queues of other waves)
    const uint32_t trace_waves = gridDim.x * (blockDim.x / 64u);
    const uint32_t wave_id = blockIdx.x * (blockDim.x / 64u) + wave_in_wg;
    uint32_t sq = wave_id, sq_items = 0, sq_n0 = 0, sq_n01 = 0, stolen = 0;
    bool stealing = false;
    uint32_t pool_next = 0, pool_end = 0;
    auto open_queue = [&]() __attribute__((always_inline)) {
        const uint32_t* const c = A.ctr + (size_t)sq * CT_STRIDE + CT_ITEMS0;
        sq_n0 = c[0]; sq_n01 = sq_n0 + c[1]; sq_items = sq_n01 + c[2];
    };
    if (sq < A.nq) open_queue();
    const uint32_t steal_max = sq_items >= A.claim ? A.steal : 0u;
    bool exhausted = sq >= A.nq;
    const gptr<double> I = as_global(A.itD);
    const gptr<uint32_t> IU = as_global(A.itU);
    const gptr<double> Pst = as_global(A.D[(round + 1u) & 1u]);
    const gptr<RtmDevModel> models_g = (gptr<RtmDevModel>)S.models;
    const gptr<uint32_t> cell_info = as_global(S.g_cell_info);
    const gptr<double> cell_rec = as_global(S.g_cell_rec);
    const gptr<double> cell_frame = as_global(S.g_cell_frame);
    const gptr<uint32_t> cell_box8 = as_global(S.g_cell_box8);
    const unsigned long long below = (1ull << lane) - 1ull;

    // lane r gets the slot of the r-th ray of the set (lo: slots 0..63, hi: slots 64..R-1), counted from the bottom or -- every
    // other step, so that no ray waits for ever behind busier ones -- from the top; n = rays picked (at most 64)
    auto pick = [&](const unsigned long long lo, const unsigned long long hi, const bool top, uint32_t& n) __attribute__((always_inline)) -> uint32_t {
        const uint32_t nlo = (uint32_t)__popcll(lo), tot = nlo + (uint32_t)__popcll(hi);
        n = tot < 64u ? tot : 64u;
        const uint32_t skip = top ? tot - n : 0u;
        if ((lo >> lane) & 1ull) {
            const uint32_t r = (uint32_t)__popcll(lo & below) - skip;  // (wraps below `skip`: not picked)
            if (r < 64u) PK[r] = lane;
        }
        if ((hi >> lane) & 1ull) {
            const uint32_t r = nlo + (uint32_t)__popcll(hi & below) - skip;
            if (r < 64u) PK[r] = 64u + lane;
        }
        __builtin_amdgcn_wave_barrier();
        const uint32_t j = PK[lane];
        __builtin_amdgcn_wave_barrier();
        return lane < n ? j : 0u;
    };
    auto probe_bit = [&](const uint32_t cv) __attribute__((always_inline)) -> bool {  // see front_trace_kernel's probe()
        if (STATS) cnt.v[C_CELLS]++;
        const uint32_t wi = ((cv >> 8) << A.occ_shift) + ((cv & A.occ_m1) | A.occ_m2);
        return ((occ_lds[wi] >> ((cv >> 3) & 31u)) & 1u) != 0u;
    };
    // the ray in slot j is done: its answer goes where the shade kernel of the next round looks for it
    auto write_result = [&](const uint32_t q, const double t, const uint32_t tri, const uint32_t model) __attribute__((always_inline)) {
        FrontResult r;
        r.t = t; r.tri = tri; r.model = model;
        A.res[q] = r;
    };

    uint32_t iter = 0;
    for (;; ++iter) {
        if (iter > (1u << 22)) break;  // (a safety net, far beyond any launch: a wave never spins for ever)
        // ---- status of the pool
        const uint32_t f0 = PU[PW_FL * R + lane];
        const bool has1 = lane < (uint32_t)(R - 64);
        const uint32_t f1 = has1 ? PU[PW_FL * R + 64 + lane] : PF_IDLE;
        const unsigned long long i_lo = __ballot((f0 & PF_IDLE) != 0u), i_hi = __ballot(has1 && (f1 & PF_IDLE) != 0u);
        const unsigned long long t_lo = __ballot((f0 & (PF_IDLE | PF_ENTER | PF_OPEN)) == PF_OPEN), t_hi = __ballot(has1 && (f1 & (PF_IDLE | PF_ENTER | PF_OPEN)) == PF_OPEN);
        const unsigned long long w_lo = __ballot((f0 & (PF_IDLE | PF_ENTER | PF_OPEN)) == 0u), w_hi = __ballot(has1 && (f1 & (PF_IDLE | PF_ENTER | PF_OPEN)) == 0u);
        unsigned long long e_lo = 0ull, e_hi = 0ull;
        if (MULTI) { e_lo = __ballot((f0 & PF_ENTER) != 0u); e_hi = __ballot(has1 && (f1 & PF_ENTER) != 0u); }
        const uint32_t nI = (uint32_t)(__popcll(i_lo) + __popcll(i_hi)), nT = (uint32_t)(__popcll(t_lo) + __popcll(t_hi));
        const uint32_t nW = (uint32_t)(__popcll(w_lo) + __popcll(w_hi)), nE = (uint32_t)(__popcll(e_lo) + __popcll(e_hi));
        const bool top = (iter & 1u) != 0u;

        // ---- refill: the next items of the wave's queue go to idle slots, 64 at a time
        if (!exhausted && (nI >= A.pool_refill || nW + nT + nE == 0u)) {
            if (pool_next == pool_end) {  // the next run of items (wave-uniform)
                for (;;) {
                    if (sq_items) {
                        uint32_t at = 0;
                        if (lane == 0) at = atomicAdd(A.ctr + (size_t)sq * CT_STRIDE + CT_CLAIM, A.claim);
                        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
                        if (at < sq_items) {
                            pool_next = at;
                            pool_end = min(at + A.claim, sq_items);
                            break;
                        }
                    }
                    if (!stealing) {
                        sq += trace_waves;
                        if (sq >= A.nq) stealing = true;
                    }
                    if (stealing) {
                        if (stolen >= steal_max) { exhausted = true; break; }
                        ++stolen;
                        sq = (wave_id + stolen * 2731u) % A.nq;
                    }
                    open_queue();
                }
                if (exhausted) continue;
            }
            if (STATS) dg_t0 = __builtin_amdgcn_s_memtime();
            uint32_t n_slots;
            const uint32_t j = pick(i_lo, i_hi, false, n_slots);
            const uint32_t avail = pool_end - pool_next;
            const uint32_t n_take = n_slots < avail ? n_slots : avail;
            if (STATS) { dg_refills++; dg_refill_lanes += n_take; }
            if (lane < n_take) {
                const uint32_t li_ = pool_next + lane;  // position in the sub-queue -> region, position in the region
                const uint32_t rg = (li_ >= sq_n0 ? 1u : 0u) + (li_ >= sq_n01 ? 1u : 0u);
                const size_t it = (size_t)sq * A.cap_i + (size_t)rg * A.cap_l + (li_ - (rg == 0u ? 0u : (rg == 1u ? sq_n0 : sq_n01)));
                const size_t ps = IU[at64(IU_COUNT, IU_P, it)];
                V3 d;
                if (rg == 0u) d = mk(Pst[at64((int)A.pf_n, PF_D0 + 0, ps)], Pst[at64((int)A.pf_n, PF_D0 + 1, ps)], Pst[at64((int)A.pf_n, PF_D0 + 2, ps)]);
                else d = mk(I[at64(IT_DCOUNT, IT_D + 0, it)], I[at64(IT_DCOUNT, IT_D + 1, it)], I[at64(IT_DCOUNT, IT_D + 2, it)]);
                const double tmx = I[at64(IT_DCOUNT, IT_TM + 0, it)], tmy = I[at64(IT_DCOUNT, IT_TM + 1, it)], tmz = I[at64(IT_DCOUNT, IT_TM + 2, it)];
                const uint32_t cw = IU[at64(IU_COUNT, IU_CELL, it)];
                const uint32_t iw = IU[at64(IU_COUNT, IU_M, it)];
                const V3 o = mk(Pst[at64((int)A.pf_n, PF_O + 0, ps)], Pst[at64((int)A.pf_n, PF_O + 1, ps)], Pst[at64((int)A.pf_n, PF_O + 2, ps)]);
                const uint32_t model_i = iw & 0xFFFFu;
                // tDelta = (step sign * cell size) / direction (core.rs:119-125; the product with +-1 is exact)
                const gptr<double> mc = (gptr<double>)models_g[model_i].cell;
                const double cx_ = mc[0], cy_ = mc[1], cz_ = mc[2];
                const double tdx = ((iw & IM_SX) ? cx_ : -cx_) / d.x;
                const double tdy = ((iw & IM_SY) ? cy_ : -cy_) / d.y;
                const double tdz = ((iw & IM_SZ) ? cz_ : -cz_) / d.z;
                const uint32_t mw = model_i < (uint32_t)kModelTab ? mtab[model_i] : (models_g[model_i].n | ((models_g[model_i].cell_base >> 5) << 10));
                const uint32_t n = mw & 1023u, base32 = mw >> 10;
                const uint32_t ix = (cw & 1023u) - 1u, iy = ((cw >> 10) & 1023u) - 1u, iz = ((cw >> 20) & 1023u) - 1u;
                const bool px = (iw & IM_SX) != 0u, py = (iw & IM_SY) != 0u, pz = (iw & IM_SZ) != 0u;
                const uint32_t cv = (((base32 << 5) + ((iz * n + iy) * n + ix)) << 3) | 6u;
                const uint32_t rem = (px ? n - 1u - ix : ix) | ((py ? n - 1u - iy : iy) << 10) | ((pz ? n - 1u - iz : iz) << 20) | kRemGuards;
                const uint32_t fl = probe_bit(cv) ? (PF_WINIT | PF_OPEN) : 0u;  // the entry cell: no DDA step before it
                PD[PD_OX * R + j] = o.x; PD[PD_OY * R + j] = o.y; PD[PD_OZ * R + j] = o.z;
                PD[PD_DX * R + j] = d.x; PD[PD_DY * R + j] = d.y; PD[PD_DZ * R + j] = d.z;
                PD[PD_TMX * R + j] = tmx; PD[PD_TMY * R + j] = tmy; PD[PD_TMZ * R + j] = tmz;
                PD[PD_TDX * R + j] = tdx; PD[PD_TDY * R + j] = tdy; PD[PD_TDZ * R + j] = tdz;
                PU[PW_CV * R + j] = cv; PU[PW_REM * R + j] = rem;
                PU[PW_Q * R + j] = (uint32_t)it;
                PU[PW_MISC * R + j] = model_i | (iw & (IM_SX | IM_SY | IM_SZ)) | (n << 19);
                PU[PW_FL * R + j] = fl;
                if (MULTI) { PD[PD_BT * R + j] = 0.0; PU[PW_BM * R + j] = kNoId; PU[PW_BTRI * R + j] = kNoId; }
            }
            pool_next += n_take;
            if (STATS) dg_tr += __builtin_amdgcn_s_memtime() - dg_t0;
            continue;
        }
        if (nW + nT + nE == 0u) {
            if (exhausted) break;
            continue;  // (not reached: an empty pool with a live queue refills above)
        }

        // ---- rays that have left a model and have further models to visit (scenes with several models)
        if (MULTI && nE != 0u && (nE >= 32u || nW + nT == 0u)) {
            uint32_t n_p;
            const uint32_t j = pick(e_lo, e_hi, top, n_p);
            const bool mine = lane < n_p;
            V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
            uint32_t model_i = 0xFFFFu;
            if (mine) {
                o = mk(PD[PD_OX * R + j], PD[PD_OY * R + j], PD[PD_OZ * R + j]);
                d = mk(PD[PD_DX * R + j], PD[PD_DY * R + j], PD[PD_DZ * R + j]);
                model_i = PU[PW_MISC * R + j] & 0xFFFFu;
            }
            bool placed = false;
            for (uint32_t m = 0; m < S.n_models; ++m) {
                if (!__ballot(mine && !placed && model_i == m)) continue;
                const __attribute__((address_space(4))) RtmDevModel& M = as_const(S.models)[m];  // (wave-uniform: scalar loads)
                if (mine && !placed && model_i == m) {
                    if (STATS) cnt.v[C_BBOX]++;
                    const double fx = 1.0 / d.x, fy = 1.0 / d.y, fz = 1.0 / d.z;
                    const double t1 = (M.bmin[0] - o.x) * fx, t2 = (M.bmax[0] - o.x) * fx;
                    const double t3 = (M.bmin[1] - o.y) * fy, t4 = (M.bmax[1] - o.y) * fy;
                    const double t5 = (M.bmin[2] - o.z) * fz, t6 = (M.bmax[2] - o.z) * fz;
                    const double tmin = fmax(fmax(fmin(t1, t2), fmin(t3, t4)), fmin(t5, t6));
                    const double tmax = fmin(fmin(fmax(t1, t2), fmax(t3, t4)), fmax(t5, t6));
                    if (tmax < 0.0 || tmin > tmax) {
                        model_i = m + 1u;
                    } else {
                        const V3 P0 = ray_at(o, d, tmin);
                        const int n = (int)M.n;
                        const int ix = cell_coord((P0.x - M.bmin[0]) / M.cell[0], n);
                        const int iy = cell_coord((P0.y - M.bmin[1]) / M.cell[1], n);
                        const int iz = cell_coord((P0.z - M.bmin[2]) / M.cell[2], n);
                        const bool px = d.x > 0.0, py = d.y > 0.0, pz = d.z > 0.0;
                        const double nbx = (M.bmin[0] + M.hc2[0] * ((double)(ix + (px ? 1 : 0)) + 0.5)) - M.hc[0];
                        const double nby = (M.bmin[1] + M.hc2[1] * ((double)(iy + (py ? 1 : 0)) + 0.5)) - M.hc[1];
                        const double nbz = (M.bmin[2] + M.hc2[2] * ((double)(iz + (pz ? 1 : 0)) + 0.5)) - M.hc[2];
                        PD[PD_TMX * R + j] = (nbx - P0.x) / d.x; PD[PD_TMY * R + j] = (nby - P0.y) / d.y; PD[PD_TMZ * R + j] = (nbz - P0.z) / d.z;
                        PD[PD_TDX * R + j] = ((px ? 1.0 : -1.0) * M.cell[0]) / d.x;
                        PD[PD_TDY * R + j] = ((py ? 1.0 : -1.0) * M.cell[1]) / d.y;
                        PD[PD_TDZ * R + j] = ((pz ? 1.0 : -1.0) * M.cell[2]) / d.z;
                        const uint32_t un = M.n, uix = (uint32_t)ix, uiy = (uint32_t)iy, uiz = (uint32_t)iz;
                        const uint32_t cv = ((M.cell_base + ((uiz * un + uiy) * un + uix)) << 3) | 6u;
                        PU[PW_CV * R + j] = cv;
                        PU[PW_REM * R + j] = (px ? un - 1u - uix : uix) | ((py ? un - 1u - uiy : uiy) << 10) | ((pz ? un - 1u - uiz : uiz) << 20) | kRemGuards;
                        PU[PW_MISC * R + j] = m | (px ? IM_SX : 0u) | (py ? IM_SY : 0u) | (pz ? IM_SZ : 0u) | (un << 19);
                        PU[PW_FL * R + j] = probe_bit(cv) ? (PF_WINIT | PF_OPEN) : 0u;
                        placed = true;
                    }
                }
            }
            if (mine && !placed) {  // no model left: the ray is done
                write_result(PU[PW_Q * R + j], PD[PD_BT * R + j], PU[PW_BTRI * R + j], PU[PW_BM * R + j]);
                PU[PW_FL * R + j] = PF_IDLE;
            }
            continue;
        }

        const bool do_test = nT != 0u && (nW == 0u || nT >= 64u || (nW < 64u && nT * 64u >= A.t_test_frac * (nW + nT)));
        if (!do_test) {
            // ---- walk: DDA advance (core.rs:151-167), bounds (core.rs:129-134), bitmap probe
            if (STATS) dg_t0 = __builtin_amdgcn_s_memtime();
            uint32_t n_p;
            const uint32_t j = pick(w_lo, w_hi, top, n_p);
            const bool mine = lane < n_p;
            double tmx = 0, tmy = 0, tmz = 0, tdx = 0, tdy = 0, tdz = 0;
            uint32_t cv = 6u, rem = kRemGuards, misc = 0;
            if (mine) {
                tmx = PD[PD_TMX * R + j]; tmy = PD[PD_TMY * R + j]; tmz = PD[PD_TMZ * R + j];
                tdx = PD[PD_TDX * R + j]; tdy = PD[PD_TDY * R + j]; tdz = PD[PD_TDZ * R + j];
                cv = PU[PW_CV * R + j]; rem = PU[PW_REM * R + j]; misc = PU[PW_MISC * R + j];
            }
            // what a step along each axis adds to cv: (cell index stride, signed) << 3 | arrival code
            const uint32_t n = (misc >> 19) & 1023u, nn = n * n;
            const uint32_t wx = (misc & IM_SX) ? ((1u << 3) | 1u) : ((0u - 1u) << 3);
            const uint32_t wy = (misc & IM_SY) ? ((n << 3) | 3u) : (((0u - n) << 3) | 2u);
            const uint32_t wz = (misc & IM_SZ) ? ((nn << 3) | 5u) : (((0u - nn) << 3) | 4u);
            bool walking = mine, left = false, found = false;
            auto step = [&]() __attribute__((always_inline)) {
                if (STATS) { dg_walks++; dg_walk_lanes += (uint64_t)__popcll(__ballot(walking)); }
                if (walking) {
                    const bool ax_ = (tmx < tmy) && (tmx < tmz);
                    const bool ay_ = !(tmx < tmy) && (tmy < tmz);
                    const bool az_ = !ax_ && !ay_;
                    const double nx_ = tmx + tdx, ny_ = tmy + tdy, nz_ = tmz + tdz;
                    tmx = ax_ ? nx_ : tmx;
                    tmy = ay_ ? ny_ : tmy;
                    tmz = az_ ? nz_ : tmz;
                    const uint32_t wx_ = wx, wy_ = wy, wz_ = wz;  // (selects of values, not of captured variables: see rt_front.h)
                    const uint32_t wyz_ = ay_ ? wy_ : wz_;
                    const uint32_t w_ = ax_ ? wx_ : wyz_;
                    const uint32_t ryz_ = ay_ ? (1u << 10) : (1u << 20);
                    const uint32_t r_ = ax_ ? 1u : ryz_;
                    cv = (cv & ~7u) + w_;
                    rem -= r_;
                    if ((rem & kRemGuards) == kRemGuards) {
                        if (probe_bit(cv)) { found = true; walking = false; }
                    } else {
                        left = true; walking = false;  // left the grid without a hit
                    }
                }
            };
#pragma unroll
            for (int u = 0; u < kPoolWalkSteps; ++u) step();
            for (int more = 0; more < kPoolWalkMore && (uint32_t)__popcll(__ballot(walking)) >= A.pool_walk_more; ++more) { step(); step(); }
            if (mine) {
                PD[PD_TMX * R + j] = tmx; PD[PD_TMY * R + j] = tmy; PD[PD_TMZ * R + j] = tmz;
                PU[PW_CV * R + j] = cv; PU[PW_REM * R + j] = rem;
                uint32_t fl = found ? (PF_WINIT | PF_OPEN) : 0u;
                if (left) {
                    if (MULTI && (misc & 0xFFFFu) + 1u < S.n_models) {
                        PU[PW_MISC * R + j] = (misc & ~0xFFFFu) | ((misc & 0xFFFFu) + 1u);
                        fl = PF_ENTER;
                    } else {
                        if (MULTI) write_result(PU[PW_Q * R + j], PD[PD_BT * R + j], PU[PW_BTRI * R + j], PU[PW_BM * R + j]);
                        else write_result(PU[PW_Q * R + j], 0.0, kNoId, kNoId);
                        fl = PF_IDLE;
                    }
                }
                PU[PW_FL * R + j] = fl;
            }
            if (STATS) dg_tw += __builtin_amdgcn_s_memtime() - dg_t0;
            continue;
        }

        // ---- test: up to A.wide entries of the open cell's window per ray (model.rs:43-74), in list order (core.rs:140-142)
        {
            if (STATS) dg_t0 = __builtin_amdgcn_s_memtime();
            uint32_t n_p;
            const uint32_t j = pick(t_lo, t_hi, top, n_p);
            const bool testing = lane < n_p;
            if (STATS) { dg_steps++; dg_step_lanes += n_p; }
            V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
            uint32_t k = 0, k_end = 0, wmask = 0, wmask_hi = 0, m_tri = 0, fl = 0, cv = 6u;
            double m_t = 0.0;
            if (testing) {
                o = mk(PD[PD_OX * R + j], PD[PD_OY * R + j], PD[PD_OZ * R + j]);
                d = mk(PD[PD_DX * R + j], PD[PD_DY * R + j], PD[PD_DZ * R + j]);
                fl = PU[PW_FL * R + j];
                cv = PU[PW_CV * R + j];
                k = PU[PW_K * R + j]; k_end = PU[PW_KEND * R + j]; wmask = PU[PW_WMASK * R + j];
                m_t = PD[PD_MT * R + j]; m_tri = PU[PW_MTRI * R + j];
                if (HAS_LONG) wmask_hi = PU[PW_WHI * R + j];
            }
            uint32_t frame_i = 0u;
            bool frame_known = false;
            if (testing && (fl & PF_WINIT)) {  // a newly opened cell: entries already tested in the cell the ray came from are skipped
                const gptr<uint32_t> ci = cell_info + (size_t)(cv >> 3) * 8;
                const uint32_t arr = cv & 7u;
                const uint32_t am = ci[2 + (arr < 6u ? arr : 0u)], w2 = ci[2];
                k_end = ci[1];
                k = ci[0];
                const uint32_t len = k_end - k;
                const uint32_t all = len >= 32u ? 0xFFFFFFFFu : ((1u << len) - 1u);
                wmask = (A.cull && len <= 32u && arr < 6u) ? (all & am) : all;
                fl &= ~(PF_HI | PF_HAVE);
                if (len > 32u) { frame_i = w2; frame_known = true; fl |= PF_LONG; }
                fl &= ~PF_WINIT;
            }
            // long cells: the box screen of front_trace_kernel (rt_front.h), unchanged
            if (HAS_LONG && A.cull && __ballot(testing && (fl & PF_LONG))) {
                const bool scan = testing && (fl & PF_LONG) != 0u;
                fl &= ~PF_LONG;
                float olx = 0.f, oly = 0.f, olz = 0.f, ivx = 0.f, ivy = 0.f, ivz = 0.f;
                uint32_t boff = 0u;
                bool finite = false;
                if (scan) {
                    if (!frame_known) frame_i = cell_info[(size_t)(cv >> 3) * 8 + 2];
                    const gptr<double> F = cell_frame + (size_t)frame_i * 8;
                    const V3 ctr = ld3(F), iq = ld3(F + 3);
                    boff = (uint32_t)__double_as_longlong(F[6]);
                    const V3 w = ctr - o;
                    const double ts = dot(w, d) * (double)__builtin_amdgcn_rcpf((float)dot(d, d));
                    const V3 pl = (ray_at(o, d, ts) - ctr) * iq;
                    olx = (float)(pl.x + 127.5); oly = (float)(pl.y + 127.5); olz = (float)(pl.z + 127.5);
                    const float dlx = (float)(d.x * iq.x), dly = (float)(d.y * iq.y), dlz = (float)(d.z * iq.z);
                    finite = fabsf(olx) + fabsf(oly) + fabsf(olz) < 65536.0f && fabsf(dlx) + fabsf(dly) + fabsf(dlz) < __builtin_inff();
                    ivx = fabsf(dlx) >= 1e-30f ? __builtin_amdgcn_rcpf(dlx) : __builtin_copysignf(1e30f, dlx);
                    ivy = fabsf(dly) >= 1e-30f ? __builtin_amdgcn_rcpf(dly) : __builtin_copysignf(1e30f, dly);
                    ivz = fabsf(dlz) >= 1e-30f ? __builtin_amdgcn_rcpf(dlz) : __builtin_copysignf(1e30f, dlz);
                }
                const float cx = -olx * ivx, cy = -oly * ivy, cz = -olz * ivz;
                const uint32_t nwin = scan ? min(k_end - k, 64u) : 0u;
                const bool px_ = !(ivx < 0.f), py_ = !(ivy < 0.f), pz_ = !(ivz < 0.f);
                const uint32_t sel0 = (px_ ? 0u : 3u) | ((py_ ? 1u : 4u) << 8) | ((pz_ ? 2u : 5u) << 16) | ((px_ ? 3u : 0u) << 24);
                const uint32_t sel1 = (py_ ? 4u : 1u) | ((pz_ ? 5u : 2u) << 8) | 0x0C0C0000u;
                const uint32_t bbase = scan ? k + boff : 0u;
                uint32_t pass_h[2] = {0u, 0u};
                typedef float v2f __attribute__((ext_vector_type(2)));
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                const __attribute__((address_space(1))) u4* const box4 = (const __attribute__((address_space(1))) u4*)cell_box8 + (bbase >> 1);
#pragma unroll
                for (uint32_t half = 0; half < 2u; ++half) {
                    uint32_t pass = 0u, n_done = 0u;
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
                            const v2f tx = __builtin_elementwise_fma(v2f{nx, fx}, v2f{ivx, ivx}, v2f{cx, cx});
                            const v2f ty = __builtin_elementwise_fma(v2f{ny, fy}, v2f{ivy, ivy}, v2f{cy, cy});
                            const v2f tz = __builtin_elementwise_fma(v2f{nz, fz}, v2f{ivz, ivz}, v2f{cz, cz});
                            const float tlo = fmaxf(fmaxf(tx.x, ty.x), tz.x);
                            const float thi = fminf(fminf(tx.y, ty.y), tz.y);
                            asm("v_cmp_ngt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(pass) : "v"(tlo), "v"(thi) : "vcc");
                        }
                    }
                    pass_h[half] = n_done ? __builtin_bitreverse32(pass) >> (32u - n_done) : 0u;
                }
                unsigned long long pass = ((unsigned long long)pass_h[1] << 32) | pass_h[0];
                pass &= nwin >= 64u ? ~0ull : ((1ull << nwin) - 1ull);
                if (scan) {
                    if (!finite) pass = nwin >= 64u ? ~0ull : ((1ull << nwin) - 1ull);
                    wmask &= (uint32_t)pass;
                    wmask_hi = (uint32_t)(pass >> 32);
                    fl |= PF_HI;
                }
            }
            if (testing) {
                bool have = (fl & PF_HAVE) != 0u;
                constexpr int kWide = RTM_FTRI_WIDE;
                uint32_t ti[kWide];
                double tt[kWide];
                bool ok[kWide];
                uint32_t cntw = 0;
#pragma unroll
                for (int s = 0; s < kWide; ++s) {  // the next untested entries, in list order
                    const bool any = wmask != 0u && (uint32_t)s < A.wide;
                    const uint32_t b = any ? (uint32_t)__builtin_ctz(wmask) : 0u;
                    ti[s] = k + b;
                    wmask = any ? (wmask & (wmask - 1u)) : wmask;
                    cntw += any ? 1u : 0u;
                }
#pragma unroll
                for (int s = 0; s < kWide; ++s) {
                    tt[s] = 0.0;
                    ok[s] = false;
                    const unsigned long long wants = __ballot((uint32_t)s < cntw);
                    if (s >= 1 && !wants) continue;  // (wave-uniform)
                    if (STATS) { dg_slots++; dg_slot_lanes += (uint64_t)__popcll(wants); }
                    const gptr<double> tp = cell_rec + (uint32_t)(ti[s] * 10u);  // (the host keeps list positions below 2^32 / 10)
                    const V3 v0 = ld3(tp), e1 = ld3(tp + 3), e2 = ld3(tp + 6);
                    ti[s] = (uint32_t)__double_as_longlong(tp[9]);
                    const V3 pv = cross(d, e2);
                    const double det = dot(e1, pv);
                    const double inv = 1.0 / det;
                    const V3 tv = o - v0;
                    const double u = dot(tv, pv) * inv;
                    const V3 qv = cross(tv, e1);
                    const double v = dot(d, qv) * inv;
                    tt[s] = dot(e2, qv) * inv;
                    ok[s] = (uint32_t)s < cntw && !(fabs(det) < kF64Eps) && !(u < 0.0 || u > 1.0) && !(v < 0.0 || u + v > 1.0) && !(tt[s] < 0.0);
                }
#pragma unroll
                for (int s = 0; s < kWide; ++s) {
                    const bool take = ok[s] && (!have || tt[s] < m_t);
                    m_t = take ? tt[s] : m_t;
                    m_tri = take ? ti[s] : m_tri;
                    have = have || take;
                }
                fl = (fl & ~PF_HAVE) | (have ? PF_HAVE : 0u);
                if (STATS) cnt.v[C_TRI] += cntw;
                if (wmask == 0u) {  // window done: the next 32 entries of a long list, or the cell is finished
                    const uint32_t adv = (HAS_LONG && (fl & PF_HI)) ? 64u : 32u;
                    if (HAS_LONG && (fl & PF_HI) && wmask_hi != 0u) {
                        k += 32u;
                        wmask = wmask_hi;
                        wmask_hi = 0u;
                        fl &= ~PF_HI;
                    } else if (k_end - k > adv) {
                        k += adv;
                        const uint32_t len = k_end - k;
                        wmask = len >= 32u ? 0xFFFFFFFFu : ((1u << len) - 1u);
                        fl = (fl & ~PF_HI) | PF_LONG;
                    } else {
                        fl &= ~(PF_OPEN | PF_HI);
                    }
                }
                if (fl & PF_OPEN) {  // the rest of the window waits for the next test step
                    PU[PW_K * R + j] = k; PU[PW_KEND * R + j] = k_end; PU[PW_WMASK * R + j] = wmask;
                    PD[PD_MT * R + j] = m_t; PU[PW_MTRI * R + j] = m_tri;
                    if (HAS_LONG) PU[PW_WHI * R + j] = wmask_hi;
                    PU[PW_FL * R + j] = fl;
                } else if (fl & PF_HAVE) {  // core.rs:147-149: the first cell with any accepted hit ends the walk
                    if (STATS) cnt.v[C_MHIT]++;
                    const uint32_t misc = PU[PW_MISC * R + j];
                    const uint32_t model_i = misc & 0xFFFFu;
                    if (MULTI) {
                        double bt = PD[PD_BT * R + j];
                        uint32_t bm = PU[PW_BM * R + j], btri = PU[PW_BTRI * R + j];
                        if (!(m_t <= kEps)) {  // core.rs:75-83 over the models
                            bool take = bm == kNoId || m_t < bt;
                            if (!take && m_t == bt) take = models_g[model_i].obj < models_g[bm].obj;
                            if (take) { bt = m_t; bm = model_i; btri = m_tri; }
                        }
                        if (model_i + 1u < S.n_models) {
                            PD[PD_BT * R + j] = bt; PU[PW_BM * R + j] = bm; PU[PW_BTRI * R + j] = btri;
                            PU[PW_MISC * R + j] = (misc & ~0xFFFFu) | (model_i + 1u);
                            PU[PW_FL * R + j] = PF_ENTER;
                        } else {
                            write_result(PU[PW_Q * R + j], bt, btri, bm);
                            PU[PW_FL * R + j] = PF_IDLE;
                        }
                    } else {
                        const bool good = !(m_t <= kEps);
                        write_result(PU[PW_Q * R + j], good ? m_t : 0.0, good ? m_tri : kNoId, good ? model_i : kNoId);
                        PU[PW_FL * R + j] = PF_IDLE;
                    }
                } else {
                    PU[PW_FL * R + j] = 0u;  // nothing accepted in this cell: the ray walks on
                }
            }
            if (STATS) dg_tt += __builtin_amdgcn_s_memtime() - dg_t0;
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
        const uint64_t dv[8] = {dg_steps, dg_step_lanes, dg_walks, dg_walk_lanes, dg_slots, dg_slot_lanes, dg_refills, dg_refill_lanes};
        for (int i = 0; i < 8; ++i)
            if (dv[i]) atomicAdd(out_cnt + 9 + i, (unsigned long long)dv[i]);
    }
    flush_counters<STATS>(cnt, out_cnt);
}

}  // namespace rtf
