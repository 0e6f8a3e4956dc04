// rt_device.h -- device-side vector math and the per-lane closest-hit query for gfx950 (CDNA4), f64 throughout: the building
// blocks of rt_front.h / rt_wave.h and, as scene_hit(), the straight per-lane form of hit() that answers rtm_trace_rays.
//
// Op for op the reference's (citations relative to /root/reference/src/):
//   hit raytracing/core.rs:68-92, Solid::hit core.rs:180-213,
//   ray_intersect core.rs:94-178, Triangle::intersect model.rs:43-74, normal_on_point model.rs:18-39,
//   Box3::intersect math/box3D.rs:64-93, Camera::shoot_to camera.rs:46-52, Vec3 ops math/vec3.rs.
// Bit-exactness rules: every expression keeps the reference's grouping; FMA contraction is OFF for
// this whole translation unit (pragma below and -ffp-contract=off); `/` and sqrt are the IEEE
// correctly-rounded f64 forms; fmin/fmax are the NaN-ignoring forms Rust's f64::min/max use.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rtm_internal.h"

#pragma clang fp contract(off)

#define RTM_DEV __device__ __forceinline__

namespace rtd {

constexpr double kEps = 1e-5;                       // core.rs:10
constexpr double kF64Eps = 2.220446049250313e-16;   // f64::EPSILON, model.rs:51
constexpr uint32_t kNoId = 0xFFFFFFFFu;

struct V3 {
    double x, y, z;
};
// Pointers read out of RtmDevModel records are generic to the compiler (it would emit flat_load, which also
// ties up the LDS counter); they always point into HBM, so say so.
template <typename T>
using gptr = const T __attribute__((address_space(1)))*;
template <typename T>
RTM_DEV gptr<T> as_global(const T* p) { return (gptr<T>)p; }
// Scene data that nothing writes during a frame, read at wave-uniform addresses: through the constant address space the
// compiler issues scalar loads (s_load into scalar registers, scalar data cache) instead of one vector load per lane.
template <typename T>
using cptr = const T __attribute__((address_space(4)))*;
template <typename T>
RTM_DEV cptr<T> as_const(const T* p) { return (cptr<T>)p; }

RTM_DEV V3 mk(double x, double y, double z) { return V3{x, y, z}; }
RTM_DEV V3 ld3(const double* p) { return V3{p[0], p[1], p[2]}; }
RTM_DEV V3 ld3(gptr<double> p) { return V3{p[0], p[1], p[2]}; }
RTM_DEV V3 ld3(cptr<double> p) { return V3{p[0], p[1], p[2]}; }
RTM_DEV V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RTM_DEV V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
RTM_DEV V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
RTM_DEV V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
RTM_DEV V3 operator/(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
RTM_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // (x*x' + y*y') + z*z', vec3.rs:179
RTM_DEV V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RTM_DEV V3 normalize(V3 a) { return a / __builtin_sqrt(dot(a, a)); }  // vec3.rs:213-215
RTM_DEV V3 ray_at(V3 o, V3 d, double t) { return o + d * t; }         // ray.rs:10-12
// component-wise select (a struct-typed ?: is lowered to a select of stack addresses, i.e. scratch traffic)
RTM_DEV V3 sel3(bool c, V3 a, V3 b) { return {c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
RTM_DEV double sqdist(V3 a, V3 b) { V3 e = a - b; return dot(e, e); }  // vec3.rs:192-194

// Rust `as u32` on f64 (saturating, NaN -> 0) followed by clamp(0, n-1)   model.rs:268-279
RTM_DEV int cell_coord(double rel, int n) {
    if (!(rel > 0.0)) return 0;
    if (rel >= (double)n) return n - 1;
    return (int)rel;
}

// ---- Philox4x32-10 (same stream layout as the oracle; DESIGN.md "RNG") ---------------------------
RTM_DEV void philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), c3 = 0u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // (one 32 x 32 -> 64 multiply per product: written as a high and a low half it becomes two quarter-rate instructions)
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
        const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
RTM_DEV double u01(uint32_t lo, uint32_t hi) {  // 52 mantissa bits -> [1,2) - 1.0
    const uint64_t u = ((uint64_t)hi << 32) | (uint64_t)lo;
    return __longlong_as_double((long long)((u >> 12) | 0x3FF0000000000000ull)) - 1.0;
}

// ---- work counters (exact; only in the STATS instantiation) -----------------------------------
struct Counters {
    uint64_t v[9];  // rays_camera, rays_bounce, rays_shadow, sphere, plane, bbox, cells, tri, model_hits
};
enum { C_CAM = 0, C_BOUNCE, C_SHADOW, C_SPHERE, C_PLANE, C_BBOX, C_CELLS, C_TRI, C_MHIT };

struct Hit {
    double t;
    uint32_t obj, tri, mat;
    uint32_t kind;  // 0 sphere, 1 plane, 2 model
    uint32_t prim;  // index inside its kind
};

// ---- one model: slab test + 3D-DDA (core.rs:94-178) ----------------------------------------------
template <bool STATS>
RTM_DEV bool model_intersect(const RtmDevModel& M, V3 o, V3 d, double& t_out, uint32_t& tri_out, Counters& cnt) {
    if (STATS) cnt.v[C_BBOX]++;
    // box3D.rs:66-91
    const double fx = 1.0 / d.x, fy = 1.0 / d.y, fz = 1.0 / d.z;
    const double t1 = (M.bmin[0] - o.x) * fx, t2 = (M.bmax[0] - o.x) * fx;
    const double t3 = (M.bmin[1] - o.y) * fy, t4 = (M.bmax[1] - o.y) * fy;
    const double t5 = (M.bmin[2] - o.z) * fz, t6 = (M.bmax[2] - o.z) * fz;
    const double tmin = fmax(fmax(fmin(t1, t2), fmin(t3, t4)), fmin(t5, t6));
    const double tmax = fmin(fmin(fmax(t1, t2), fmax(t3, t4)), fmax(t5, t6));
    if (tmax < 0.0) return false;
    if (tmin > tmax) return false;
    const V3 P0 = ray_at(o, d, tmin);  // core.rs:96 -- may lie BEHIND the origin (tmin < 0)
    const int n = (int)M.n;
    int ix = cell_coord((P0.x - M.bmin[0]) / M.cell[0], n);  // model.rs:267-280
    int iy = cell_coord((P0.y - M.bmin[1]) / M.cell[1], n);
    int iz = cell_coord((P0.z - M.bmin[2]) / M.cell[2], n);
    const int sx = d.x > 0.0 ? 1 : -1, sy = d.y > 0.0 ? 1 : -1, sz = d.z > 0.0 ? 1 : -1;  // core.rs:102-104
    // core.rs:107-119 with cell_box(i).min() = (bmin + hc2*(i+0.5)) - hc   (model.rs:231-240, box3D.rs:56)
    const double nbx = (M.bmin[0] + M.hc2[0] * ((double)(ix + (sx > 0 ? 1 : 0)) + 0.5)) - M.hc[0];
    const double nby = (M.bmin[1] + M.hc2[1] * ((double)(iy + (sy > 0 ? 1 : 0)) + 0.5)) - M.hc[1];
    const double nbz = (M.bmin[2] + M.hc2[2] * ((double)(iz + (sz > 0 ? 1 : 0)) + 0.5)) - M.hc[2];
    double tmx = (nbx - P0.x) / d.x, tmy = (nby - P0.y) / d.y, tmz = (nbz - P0.z) / d.z;
    const double tdx = ((double)sx * M.cell[0]) / d.x;  // core.rs:122-125
    const double tdy = ((double)sy * M.cell[1]) / d.y;
    const double tdz = ((double)sz * M.cell[2]) / d.z;
    bool have = false;
    double best_t = 0.0;
    uint32_t best_i = 0;
    const gptr<uint32_t> cell_start = as_global(M.cell_start), tri_idx = as_global(M.tri_idx);
    const gptr<double> tri_pos = as_global(M.tri_pos);
    while ((unsigned)ix < (unsigned)n && (unsigned)iy < (unsigned)n && (unsigned)iz < (unsigned)n) {
        if (STATS) cnt.v[C_CELLS]++;
        const uint32_t cell = ((uint32_t)iz * (uint32_t)n + (uint32_t)iy) * (uint32_t)n + (uint32_t)ix;
        const uint32_t k_begin = cell_start[cell], k_end = cell_start[cell + 1];  // [start, end)
        for (uint32_t k = k_begin; k < k_end; ++k) {
            const uint32_t ti = tri_idx[k];
            if (STATS) cnt.v[C_TRI]++;
            const gptr<double> tp = tri_pos + (size_t)ti * 9;
            // model.rs:45-72; the record holds v0 and the two edges v1-v0, v2-v0 (model.rs:46-47 evaluated at upload)
            const V3 v0 = ld3(tp), e1 = ld3(tp + 3), e2 = ld3(tp + 6);
            const V3 p = cross(d, e2);
            const double det = dot(e1, p);
            if (fabs(det) < kF64Eps) continue;
            const double inv = 1.0 / det;
            const V3 tv = o - v0;
            const double u = dot(tv, p) * inv;
            if (u < 0.0 || u > 1.0) continue;
            const V3 q = cross(tv, e1);
            const double v = dot(d, q) * inv;
            if (v < 0.0 || u + v > 1.0) continue;
            const double t = dot(e2, q) * inv;
            if (t < 0.0) continue;
            if (!have || t < best_t) {  // core.rs:140-142: strict, lowest list position wins ties
                have = true;
                best_t = t;
                best_i = ti;
            }
        }
        if (have) break;  // core.rs:147-149
        if (tmx < tmy) {  // core.rs:151-167
            if (tmx < tmz) { ix += sx; tmx += tdx; } else { iz += sz; tmz += tdz; }
        } else {
            if (tmy < tmz) { iy += sy; tmy += tdy; } else { iz += sz; tmz += tdz; }
        }
    }
    if (!have) return false;
    if (STATS) cnt.v[C_MHIT]++;
    t_out = best_t;
    tri_out = best_i;
    return true;
}

// model.rs:18-39 at hit_point = ray.at(t)  (core.rs:169-177); _raw stops before the final normalize
RTM_DEV V3 model_normal_raw(const RtmDevModel& M, uint32_t ti, V3 point) {
    const gptr<double> tp = as_global(M.tri_pos) + (size_t)ti * 9;
    const gptr<double> np = as_global(M.tri_nrm) + (size_t)ti * 9;
    const V3 v0 = ld3(tp), s0 = ld3(tp + 3), s1 = ld3(tp + 6), s2 = point - v0;  // s0 = v1-v0, s1 = v2-v0 (model.rs:20-21)
    const double d00 = dot(s0, s0), d01 = dot(s0, s1), d11 = dot(s1, s1), d20 = dot(s2, s0), d21 = dot(s2, s1);
    const double den = d00 * d11 - d01 * d01;
    const double v = (d11 * d20 - d01 * d21) / den;
    const double w = (d00 * d21 - d01 * d20) / den;
    const double u = 1.0 - v - w;
    return ld3(np) * u + ld3(np + 3) * v + ld3(np + 6) * w;
}
RTM_DEV V3 model_normal(const RtmDevModel& M, uint32_t ti, V3 point) { return normalize(model_normal_raw(M, ti, point)); }

// ---- hit (core.rs:68-92).  Objects are visited per kind, not in file order: the reference's
// "strictly smaller t, so the earliest object wins a tie" (core.rs:79) is kept by breaking ties on
// the object index.  The normal is evaluated only for the winner (same inputs -> same bits).
template <bool STATS>
RTM_DEV bool scene_hit(const RtmDevScene& S, V3 o, V3 d, Hit& h, Counters& cnt) {
    double best = __builtin_inf();
    bool found = false;
    h.obj = kNoId; h.tri = kNoId; h.mat = 0; h.kind = 0; h.prim = 0;
    auto consider = [&](double t, uint32_t obj, uint32_t mat, uint32_t kind, uint32_t prim, uint32_t tri) {
        if (t <= kEps) return;  // core.rs:75-77 (NaN falls through to the compares below, which are false)
        if (t < best || (found && t == best && obj < h.obj)) {
            best = t; found = true;
            h.obj = obj; h.mat = mat; h.kind = kind; h.prim = prim; h.tri = tri;
        }
    };
    const double a = dot(d, d);  // core.rs:185
    for (uint32_t i = 0; i < S.n_spheres; ++i) {  // core.rs:183-197
        if (STATS) cnt.v[C_SPHERE]++;
        const double* sp = S.spheres + (size_t)i * 4;
        const V3 oc = o - ld3(sp);
        const double b = 2.0 * dot(d, oc);
        const double c = dot(oc, oc) - sp[3] * sp[3];
        const double disc = b * b - 4.0 * a * c;
        if (disc < 0.0) continue;
        const double t = (-b - __builtin_sqrt(disc)) / (2.0 * a);
        consider(t, S.sph_ids[i * 2 + 1], S.sph_ids[i * 2], 0u, i, kNoId);
    }
    for (uint32_t i = 0; i < S.n_planes; ++i) {  // core.rs:198-210
        if (STATS) cnt.v[C_PLANE]++;
        const double* pp = S.planes + (size_t)i * 4;
        const V3 nrm = ld3(pp);
        const double dv = dot(nrm, d);
        if (fabs(dv) < kEps) continue;
        const V3 center = nrm * pp[3];
        const double t = dot(center - o, nrm) / dv;
        if (t < kEps) continue;
        consider(t, S.pl_ids[i * 2 + 1], S.pl_ids[i * 2], 1u, i, kNoId);
    }
    for (uint32_t i = 0; i < S.n_models; ++i) {
        const RtmDevModel& M = S.models[i];
        double t;
        uint32_t tri;
        if (model_intersect<STATS>(M, o, d, t, tri, cnt)) consider(t, M.obj, M.mat, 2u, i, tri);
    }
    h.t = best;
    return found;
}

RTM_DEV V3 hit_normal(const RtmDevScene& S, const Hit& h, V3 o, V3 d, V3 point) {
    if (h.kind == 0u) {  // core.rs:195
        const double* sp = S.spheres + (size_t)h.prim * 4;
        return normalize(ray_at(o, d, h.t) - ld3(sp));
    } else if (h.kind == 1u) {  // core.rs:209
        return ld3(S.planes + (size_t)h.prim * 4);
    }
    return model_normal(S.models[h.prim], h.tri, point);
}

}  // namespace rtd
