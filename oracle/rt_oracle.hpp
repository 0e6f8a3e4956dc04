// rt_oracle.hpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// A C++17 restatement of the colobrodo/raytracer-rs render path, written from the
// reference's Rust sources (cited as file:line relative to /root/reference/).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// PARITY STATUS: "parity unpinned" by the reference itself -- the reference has no
// tests, its golden images are stripped (.MISSING_LARGE_BLOBS:1-4) and no Rust toolchain
// exists here, so the reference binary cannot run anywhere in this pipeline.  The oracle
// is pinned by (1) hand-derived known-answer tests from the cited lines (tests/test_oracle_kat.py),
// (2) structural goldens of the grid (tests/golden/grid_stats.json) and (3) an independent
// pure-Python restatement (oracle/py_restatement.py) that must agree bit for bit.
//
// Arithmetic rules: every expression keeps Rust's left-to-right
// grouping, no FMA contraction (build with -ffp-contract=off), IEEE f64 / and sqrt,
// Rust `f64::min/max` = fmin/fmax (NaN-ignoring), Rust `as` casts saturate.
//
// The reference's thread_rng() (main.rs:162-163, vec3.rs:160-162) is OS-seeded and not
// reproducible, so it is replaced -- here and in the HIP kernels -- by Philox4x32-10
// keyed on (seed; pixel, sample, stream).  The float mapping follows rand 0.8.5's
// UniformFloat::sample_single: 52 random mantissa bits -> [1,2) - 1.0, * (hi-lo) + lo.
#pragma once
#include <algorithm>
#include <array>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace orc {

// ---------------------------------------------------------------- casts (Rust `as`)
static inline uint32_t as_u32(double v) {  // saturating, NaN -> 0
    if (!(v > 0.0)) return 0u;
    if (v >= 4294967295.0) return 4294967295u;
    return (uint32_t)v;
}
static inline uint64_t as_usize(double v) {
    if (!(v > 0.0)) return 0u;
    if (v >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)v;
}
static inline uint8_t as_u8(double v) {
    if (!(v > 0.0)) return 0u;
    if (v >= 255.0) return 255u;
    return (uint8_t)v;
}

// ---------------------------------------------------------------- Vec3 (math/vec3.rs:5-215)
struct Vec3 {
    double x, y, z;
};
static inline Vec3 v3(double x, double y, double z) { return Vec3{x, y, z}; }
static inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }  // vec3.rs:29-35
static inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }  // vec3.rs:51-57
static inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }       // vec3.rs:64-70
static inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }  // vec3.rs:77-83
static inline Vec3 operator/(Vec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }       // vec3.rs:90-96
static inline Vec3 operator/(Vec3 a, Vec3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }  // vec3.rs:103-109
static inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       // vec3.rs:178-180
static inline Vec3 cross(Vec3 a, Vec3 b) {                                                    // vec3.rs:183-189
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline double squared_len(Vec3 a) { return dot(a, a); }                     // vec3.rs:202-204
static inline double len(Vec3 a) { return std::sqrt(squared_len(a)); }            // vec3.rs:207-210
static inline Vec3 normalize(Vec3 a) { return a / len(a); }                        // vec3.rs:213-215
static inline double squared_distance(Vec3 a, Vec3 b) { return squared_len(a - b); }  // vec3.rs:192-194
static inline Vec3 reflect(Vec3 self, Vec3 axis) {                                  // vec3.rs:172-175
    return self - axis * 2.0 * dot(self, axis);
}

// ---------------------------------------------------------------- Ray (math/ray.rs:4-21)
struct Ray {
    Vec3 origin, direction;
    Vec3 at(double t) const { return origin + direction * t; }  // ray.rs:10-12
};
struct HitResult {
    double t;
    Vec3 normal;
};

// ---------------------------------------------------------------- Philox4x32-10 (replaces rand)
struct Philox {
    static inline void round(uint32_t c[4], const uint32_t k[2]) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    static inline void gen(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
        uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t c[4] = {c0, c1, c2, c3};
        for (int i = 0; i < 10; ++i) {
            round(c, k);
            k[0] += 0x9E3779B9u;
            k[1] += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};
// 52 mantissa bits -> [0,1)   (rand 0.8.5 UniformFloat: (u64 >> 12) | exponent(0), minus 1.0)
static inline double u01_from_bits(uint32_t lo, uint32_t hi) {
    const uint64_t u = ((uint64_t)hi << 32) | (uint64_t)lo;
    const uint64_t bits = (u >> 12) | 0x3FF0000000000000ull;
    double v12;
    std::memcpy(&v12, &bits, 8);
    return v12 - 1.0;
}
static inline double gen_range(double u01, double lo, double hi) { return u01 * (hi - lo) + lo; }

// Stream layout shared with the HIP kernels (DESIGN.md "RNG"):
//   counter = (pixel_index, sample, stream, 0)
//   stream 0            -> x_offset (words 0,1), y_offset (words 2,3)             main.rs:162-163
//   stream 1+2b, 2+2b   -> Vec3::random() of bounce b: dx,dy then dz               vec3.rs:160-162
struct SampleRng {
    uint64_t seed;
    uint32_t pixel, sample;
    void jitter(double& xo, double& yo) const {
        uint32_t o[4];
        Philox::gen(seed, pixel, sample, 0u, 0u, o);
        xo = gen_range(u01_from_bits(o[0], o[1]), -0.5, 0.5);
        yo = gen_range(u01_from_bits(o[2], o[3]), -0.5, 0.5);
    }
    Vec3 random_vec(uint32_t bounce) const {
        uint32_t a[4], b[4];
        Philox::gen(seed, pixel, sample, 1u + 2u * bounce, 0u, a);
        Philox::gen(seed, pixel, sample, 2u + 2u * bounce, 0u, b);
        return v3(gen_range(u01_from_bits(a[0], a[1]), 0.0, 1.0), gen_range(u01_from_bits(a[2], a[3]), 0.0, 1.0),
                  gen_range(u01_from_bits(b[0], b[1]), 0.0, 1.0));
    }
};

// ---------------------------------------------------------------- Mat4 (math/mat4.rs:4-231)
struct Mat4 {
    double m[16];
    static Mat4 identity() {  // mat4.rs:9-15
        return Mat4{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    }
    static Mat4 scale(double f) {  // mat4.rs:17-24
        return Mat4{{f, 0, 0, 0, 0, f, 0, 0, 0, 0, f, 0, 0, 0, 0, 1}};
    }
    static Mat4 translate(Vec3 o) {  // mat4.rs:26-33
        return Mat4{{1, 0, 0, o.x, 0, 1, 0, o.y, 0, 0, 1, o.z, 0, 0, 0, 1}};
    }
    static Mat4 rotate(Vec3 axis, double angle) {  // mat4.rs:35-60 -- the nine terms verbatim (quirk Q16)
        const Vec3 u = normalize(axis);
        const double c = std::cos(angle), s = std::sin(angle);
        Mat4 r;
        r.m[0] = c + (u.x * u.x) * (1.0 - c);
        r.m[1] = u.x * u.y * (1.0 - c) - u.z * s;
        r.m[2] = u.x * u.z * (1.0 - c) - u.y * s;
        r.m[3] = 0.0;
        r.m[4] = u.x * u.y * (1.0 - c) - u.x * s;
        r.m[5] = c + (u.y * u.y) * (1.0 - c);
        r.m[6] = u.z * u.y * (1.0 - c) - u.x * s;
        r.m[7] = 0.0;
        r.m[8] = u.x * u.z * (1.0 - c) - u.y * s;
        r.m[9] = u.z * u.y * (1.0 - c) + u.x * s;
        r.m[10] = c + u.z * u.z * (1.0 - c);
        r.m[11] = 0.0;
        r.m[12] = 0.0; r.m[13] = 0.0; r.m[14] = 0.0; r.m[15] = 1.0;
        return r;
    }
    double det3() const {  // mat4.rs:63-67 (first nine slots as a row-major 3x3)
        return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
               m[2] * (m[3] * m[7] - m[4] * m[6]);
    }
    Mat4 sub_matrix(int rr, int cc) const {  // mat4.rs:70-86
        Mat4 s;
        for (double& v : s.m) v = 0.0;
        int idx = 0;
        for (int r = 0; r < 4; ++r) {
            if (r == rr) continue;
            for (int c = 0; c < 4; ++c) {
                if (c == cc) continue;
                s.m[idx++] = m[r * 4 + c];
            }
        }
        return s;
    }
    static double sign_pow(int i) { return (i & 1) ? -1.0 : 1.0; }  // (-1.0f64).powi(i)
    double determinant() const {                                      // mat4.rs:89-96
        double det = 0.0;
        for (int i = 0; i < 4; ++i) det += sign_pow(i) * m[i] * sub_matrix(0, i).det3();
        return det;
    }
    bool inverse(Mat4& out) const {  // mat4.rs:98-121
        const double det = determinant();
        if (det == 0.0) return false;
        double cof[16];
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) cof[c * 4 + r] = sign_pow(r + c) * sub_matrix(r, c).det3();
        for (int i = 0; i < 16; ++i) out.m[i] = cof[i] / det;
        return true;
    }
    Mat4 transpose() const {  // mat4.rs:123-144
        Mat4 t;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) t.m[r * 4 + c] = m[c * 4 + r];
        return t;
    }
    Mat4 then(const Mat4& o) const {  // mat4.rs:146-222 : other * self, sums left to right
        Mat4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                r.m[i * 4 + j] = o.m[i * 4 + 0] * m[0 + j] + o.m[i * 4 + 1] * m[4 + j] + o.m[i * 4 + 2] * m[8 + j] +
                                 o.m[i * 4 + 3] * m[12 + j];
        return r;
    }
    Vec3 apply(Vec3 v) const {  // mat4.rs:224-230
        const double x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
        const double y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7];
        const double z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11];
        const double w = m[12] * v.x + m[13] * v.y + m[14] * v.z + m[15];
        return v3(x / w, y / w, z / w);
    }
};

// ---------------------------------------------------------------- Box3 (math/box3D.rs:4-93)
static inline double rust_signum(double v) {
    if (std::isnan(v)) return v;
    return std::signbit(v) ? -1.0 : 1.0;
}
struct Box3 {
    Vec3 center, half;
    static Box3 from_single_point(Vec3 p) { return {p, v3(0, 0, 0)}; }                          // box3D.rs:17-22
    static Box3 from_min_max(Vec3 mn, Vec3 mx) { return {(mn + mx) * 0.5, (mx - mn) * 0.5}; }  // box3D.rs:24-29
    void include(Vec3 p) {  // box3D.rs:31-45 -- keeps the reference's (d-h)/4 centre update (quirk Q9)
        const Vec3 d = p - center;
        if (std::fabs(d.x) > half.x) {
            half.x = (std::fabs(d.x) + half.x) / 2.0;
            center.x += (d.x - half.x * rust_signum(d.x)) / 2.0;
        }
        if (std::fabs(d.y) > half.y) {
            half.y = (std::fabs(d.y) + half.y) / 2.0;
            center.y += (d.y - half.y * rust_signum(d.y)) / 2.0;
        }
        if (std::fabs(d.z) > half.z) {
            half.z = (std::fabs(d.z) + half.z) / 2.0;
            center.z += (d.z - half.z * rust_signum(d.z)) / 2.0;
        }
    }
    Vec3 min() const { return center - half; }  // box3D.rs:55-57
    Vec3 max() const { return center + half; }  // box3D.rs:59-61
    bool intersect(const Ray& ray, double& t_out) const {  // box3D.rs:64-93
        const Vec3 df = v3(1.0 / ray.direction.x, 1.0 / ray.direction.y, 1.0 / ray.direction.z);
        const Vec3 rmin = min() - ray.origin;
        const Vec3 rmax = max() - ray.origin;
        const double t1 = rmin.x * df.x, t2 = rmax.x * df.x;
        const double t3 = rmin.y * df.y, t4 = rmax.y * df.y;
        const double t5 = rmin.z * df.z, t6 = rmax.z * df.z;
        const double tmin = std::fmax(std::fmax(std::fmin(t1, t2), std::fmin(t3, t4)), std::fmin(t5, t6));
        const double tmax = std::fmin(std::fmin(std::fmax(t1, t2), std::fmax(t3, t4)), std::fmax(t5, t6));
        if (tmax < 0.0) return false;
        if (tmin > tmax) return false;
        t_out = tmin;  // may be negative when the origin is inside (quirk Q5)
        return true;
    }
};

// ---------------------------------------------------------------- Camera (camera.rs:4-53)
struct Camera {
    Vec3 forward, up, right, position;
    double zoom;
    static Camera at(Vec3 pos) { return {v3(0, 0, 1), v3(0, 1, 0), v3(1, 0, 0), pos, 1.0}; }  // camera.rs:13-21
    static Camera look_at(Vec3 pos, Vec3 point) {                                              // camera.rs:30-43
        const Vec3 world_up = v3(0, 1, 0);
        const Vec3 f = normalize(point - pos);
        const Vec3 r = normalize(cross(world_up, f));
        const Vec3 u = normalize(cross(f, r));
        return {f, u, r, pos, 1.0};
    }
    Ray shoot_to(double u, double v) const {  // camera.rs:46-52
        const Vec3 d = forward * zoom + up * v + right * u;
        return {position, normalize(d)};
    }
};

// ---------------------------------------------------------------- OBJ (stand-in for obj-rs 0.7.1 load_obj::<Vertex,_,u16>)
// Behaviour restated from the crate's published semantics (source not in the container):
// `v`/`vn` parsed as f32; every face must be a triangle whose corners carry a normal index;
// vertices are de-duplicated on (position index, normal index) in order of first appearance.
struct ObjVertex {
    float position[3], normal[3];
};
struct Obj {
    std::vector<ObjVertex> vertices;
    std::vector<uint32_t> indices;  // u16 in the reference; values are checked to fit
};
// synth_normals: NOT reference behaviour -- used only for models/teapot.obj (no `vn` lines, which
// the reference loader would refuse); corners then get the geometric normal of their face.
static inline bool load_obj(const std::string& path, Obj& out, std::string& err, bool synth_normals = false) {
    std::ifstream in(path);
    if (!in) { err = "cannot open " + path; return false; }
    std::vector<std::array<float, 3>> pos, nrm;
    std::map<std::pair<long, long>, uint32_t> cache;
    std::string line;
    auto resolve = [](long idx, size_t n) -> long { return idx > 0 ? idx - 1 : (long)n + idx; };
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "v" || tag == "vn") {
            std::string a, b, c;
            if (!(ls >> a >> b >> c)) { err = "bad vertex line: " + line; return false; }
            std::array<float, 3> p = {std::strtof(a.c_str(), nullptr), std::strtof(b.c_str(), nullptr),
                                      std::strtof(c.c_str(), nullptr)};
            (tag == "v" ? pos : nrm).push_back(p);
        } else if (tag == "f") {
            std::vector<std::pair<long, long>> corners;
            std::string tok;
            while (ls >> tok) {
                long pi = 0, ni = 0;
                bool has_n = false;
                const size_t s1 = tok.find('/');
                if (s1 == std::string::npos) {
                    pi = std::strtol(tok.c_str(), nullptr, 10);
                } else {
                    pi = std::strtol(tok.substr(0, s1).c_str(), nullptr, 10);
                    const size_t s2 = tok.find('/', s1 + 1);
                    if (s2 != std::string::npos && s2 + 1 < tok.size()) {
                        ni = std::strtol(tok.substr(s2 + 1).c_str(), nullptr, 10);
                        has_n = true;
                    }
                }
                if (!has_n && !synth_normals) {
                    err = "Tried to extract normal data which are not contained in the model";
                    return false;
                }
                corners.push_back({resolve(pi, pos.size()), has_n ? resolve(ni, nrm.size()) : -1});
            }
            if (corners.size() != 3) { err = "Model should be triangulated first to be loaded properly"; return false; }
            if (synth_normals) {
                // one fresh vertex per corner carrying the face's geometric normal
                Vec3 p[3];
                for (int k = 0; k < 3; ++k) {
                    const auto& q = pos.at((size_t)corners[k].first);
                    p[k] = v3(q[0], q[1], q[2]);
                }
                const Vec3 g = normalize(cross(p[1] - p[0], p[2] - p[0]));
                for (int k = 0; k < 3; ++k) {
                    ObjVertex v;
                    const auto& q = pos.at((size_t)corners[k].first);
                    for (int j = 0; j < 3; ++j) v.position[j] = q[j];
                    v.normal[0] = (float)g.x; v.normal[1] = (float)g.y; v.normal[2] = (float)g.z;
                    out.indices.push_back((uint32_t)out.vertices.size());
                    out.vertices.push_back(v);
                }
                continue;
            }
            for (auto& c : corners) {
                auto it = cache.find(c);
                uint32_t vi;
                if (it == cache.end()) {
                    if (c.first < 0 || (size_t)c.first >= pos.size() || c.second < 0 || (size_t)c.second >= nrm.size()) {
                        err = "face index out of range";
                        return false;
                    }
                    ObjVertex v;
                    for (int j = 0; j < 3; ++j) { v.position[j] = pos[(size_t)c.first][j]; v.normal[j] = nrm[(size_t)c.second][j]; }
                    vi = (uint32_t)out.vertices.size();
                    out.vertices.push_back(v);
                    cache.emplace(c, vi);
                } else {
                    vi = it->second;
                }
                out.indices.push_back(vi);
            }
        }
    }
    if (!synth_normals && out.vertices.size() > 65536) { err = "vertex index does not fit u16"; return false; }
    return true;
}

// ---------------------------------------------------------------- Triangle / Model / ModelGrid (model.rs)
struct Triangle {
    Vec3 v0, v1, v2;
    void barycentric(Vec3 p, double& u, double& v, double& w) const {  // model.rs:18-33
        const Vec3 s0 = v1 - v0, s1 = v2 - v0, s2 = p - v0;
        const double d00 = dot(s0, s0), d01 = dot(s0, s1), d11 = dot(s1, s1), d20 = dot(s2, s0), d21 = dot(s2, s1);
        const double den = d00 * d11 - d01 * d01;
        v = (d11 * d20 - d01 * d21) / den;
        w = (d00 * d21 - d01 * d20) / den;
        u = 1.0 - v - w;
    }
    Vec3 normal_on_point(Vec3 p, const Vec3 n[3]) const {  // model.rs:35-39
        double u, v, w;
        barycentric(p, u, v, w);
        return normalize(n[0] * u + n[1] * v + n[2] * w);
    }
    bool intersect(const Ray& ray, double& t_out) const {  // model.rs:43-74 (Moller-Trumbore)
        const Vec3 e1 = v1 - v0, e2 = v2 - v0;
        const Vec3 p = cross(ray.direction, e2);
        const double det = dot(e1, p);
        if (std::fabs(det) < std::numeric_limits<double>::epsilon()) return false;
        const double inv = 1.0 / det;
        const Vec3 tv = ray.origin - v0;
        const double u = dot(tv, p) * inv;
        if (u < 0.0 || u > 1.0) return false;
        const Vec3 q = cross(tv, e1);
        const double v = dot(ray.direction, q) * inv;
        if (v < 0.0 || u + v > 1.0) return false;
        const double t = dot(e2, q) * inv;
        if (t < 0.0) return false;
        t_out = t;
        return true;
    }
};

struct Counters {  // per-thread work counters
    uint64_t rays_camera = 0, rays_bounce = 0, rays_shadow = 0;
    uint64_t sphere_tests = 0, plane_tests = 0, bbox_tests = 0, cells_visited = 0, tri_tests = 0, model_hits = 0;
    void add(const Counters& o) {
        rays_camera += o.rays_camera; rays_bounce += o.rays_bounce; rays_shadow += o.rays_shadow;
        sphere_tests += o.sphere_tests; plane_tests += o.plane_tests; bbox_tests += o.bbox_tests;
        cells_visited += o.cells_visited; tri_tests += o.tri_tests; model_hits += o.model_hits;
    }
};

struct ModelGrid {  // model.rs:205-210
    Box3 bbox;
    uint32_t n = 0;
    std::vector<uint64_t> offset_array;      // inclusive prefix sums
    std::vector<uint64_t> triangle_indices;
    void triangles(uint32_t ix, uint32_t iy, uint32_t iz, size_t& start, size_t& end) const {  // model.rs:218-228
        const size_t index = (size_t)(iz * n * n + iy * n + ix);
        start = index > 0 ? offset_array[index - 1] : 0;
        end = offset_array[index];
    }
    Vec3 cell_box_min(uint32_t ix, uint32_t iy, uint32_t iz) const {  // model.rs:231-240 + box3D.rs:55-57
        const Vec3 hc = bbox.half / (double)n;
        const Vec3 c = bbox.min() + hc * 2.0 * v3((double)ix + 0.5, (double)iy + 0.5, (double)iz + 0.5);
        return c - hc;
    }
    Vec3 cell_size() const { return (bbox.max() - bbox.min()) / (double)n; }  // model.rs:243-245
    void closest_cell(Vec3 p, uint32_t& ix, uint32_t& iy, uint32_t& iz) const {  // model.rs:267-280
        const Vec3 rel = (p - bbox.min()) / cell_size();
        auto cl = [this](uint32_t v) { return v > n - 1 ? n - 1 : v; };
        ix = cl(as_u32(rel.x)); iy = cl(as_u32(rel.y)); iz = cl(as_u32(rel.z));
    }
};

struct Model {  // model.rs:9-14
    Obj obj;
    Mat4 trasform, normal_trasform;
    Box3 bounding_box;
    // "optimised CPU" variant only: vertices/normals transformed once (bit-identical by construction,
    // get_triangle is a pure function of the vertex).  Faithful-cost mode never reads these.
    std::vector<Triangle> pre_tri;
    std::vector<Vec3> pre_nrm;

    static Vec3 f32v(const float p[3]) { return v3((double)p[0], (double)p[1], (double)p[2]); }  // vec3.rs:18-22
    size_t n_tris() const { return obj.indices.size() / 3; }
    Triangle get_triangle(size_t i) const {  // model.rs:108-122
        return {trasform.apply(f32v(obj.vertices[obj.indices[i * 3 + 0]].position)),
                trasform.apply(f32v(obj.vertices[obj.indices[i * 3 + 1]].position)),
                trasform.apply(f32v(obj.vertices[obj.indices[i * 3 + 2]].position))};
    }
    void get_triangle_normals(size_t i, Vec3 out[3]) const {  // model.rs:125-139
        for (int k = 0; k < 3; ++k)
            out[k] = normalize(normal_trasform.apply(f32v(obj.vertices[obj.indices[i * 3 + k]].normal)));
    }
    static bool make(Obj&& o, const Mat4& t, Model& out, std::string& err) {  // model.rs:91-103, :296-323
        const double inf = std::numeric_limits<double>::infinity();
        Vec3 mx = v3(-inf, -inf, -inf), mn = v3(inf, inf, inf);
        for (const ObjVertex& vtx : o.vertices) {
            const Vec3 v = t.apply(f32v(vtx.position));
            if (mx.x < v.x) mx.x = v.x;
            if (mx.y < v.y) mx.y = v.y;
            if (mx.z < v.z) mx.z = v.z;
            if (mn.x > v.x) mn.x = v.x;
            if (mn.y > v.y) mn.y = v.y;
            if (mn.z > v.z) mn.z = v.z;
        }
        Mat4 inv;
        if (!t.inverse(inv)) { err = "Error: the provided matrix to transform the model is not invertible"; return false; }
        out.obj = std::move(o);
        out.trasform = t;
        out.normal_trasform = inv.transpose();
        out.bounding_box = Box3::from_min_max(mn, mx);
        return true;
    }
    ModelGrid create_grid(size_t cps) const {  // model.rs:147-201 (quirks Q9, Q10)
        std::vector<std::vector<uint64_t>> cells(cps * cps * cps);
        const Vec3 cell_size = (bounding_box.max() - bounding_box.min()) / (double)cps;
        const size_t nt = n_tris();
        for (size_t i = 0; i < nt; ++i) {
            const Triangle tr = get_triangle(i);
            Box3 bb = Box3::from_single_point(tr.v0);
            bb.include(tr.v1);
            bb.include(tr.v2);
            const Vec3 mnp = (bb.min() - bounding_box.min()) / cell_size;
            const size_t min_ix = std::min<uint64_t>(as_usize(mnp.x), cps - 1);
            const size_t min_iy = std::min<uint64_t>(as_usize(mnp.y), cps - 1);
            const size_t min_iz = std::min<uint64_t>(as_usize(mnp.z), cps - 1);
            const Vec3 mxp = (bb.max() - bounding_box.min()) / cell_size;
            const size_t max_ix = std::min<uint64_t>(as_usize(mxp.x), cps - 1);
            const size_t max_iy = std::min<uint64_t>(as_usize(mxp.y), cps - 1);
            const size_t max_iz = std::min<uint64_t>(as_usize(mxp.z), cps - 1);
            for (size_t ix = min_ix; ix <= max_ix; ++ix)
                for (size_t iy = min_iy; iy <= max_iy; ++iy)
                    for (size_t iz = min_iz; iz <= max_iz; ++iz) {
                        auto& cell = cells[iz * cps * cps + iy * cps + ix];
                        if (!cell.empty() && cell.back() == i) continue;
                        cell.push_back(i);
                    }
        }
        ModelGrid g;
        g.bbox = bounding_box;
        g.n = (uint32_t)cps;
        g.offset_array.assign(cps * cps * cps, 0);
        uint64_t off = 0;
        for (size_t i = 0; i < cells.size(); ++i) {
            off += cells[i].size();
            g.offset_array[i] = off;
            g.triangle_indices.insert(g.triangle_indices.end(), cells[i].begin(), cells[i].end());
        }
        return g;
    }
    void precompute() {
        const size_t nt = n_tris();
        pre_tri.resize(nt);
        pre_nrm.resize(nt * 3);
        for (size_t i = 0; i < nt; ++i) {
            pre_tri[i] = get_triangle(i);
            get_triangle_normals(i, &pre_nrm[i * 3]);
        }
    }
};

// ---------------------------------------------------------------- Scene (core.rs:12-53)
enum MaterialType : uint32_t { Plastic = 0, Metal = 1 };  // core.rs:24-27
struct Material {
    Vec3 color;
    MaterialType type;
};
struct Light {
    Vec3 position, color;
    double radius;
};
enum SolidKind : uint32_t { SPHERE = 0, PLANE = 1, MODEL = 2 };
struct SceneObject {
    SolidKind kind;
    Vec3 a;          // sphere centre | plane normal
    double b = 0.0;  // sphere radius | plane distance
    std::shared_ptr<Model> model;
    std::shared_ptr<ModelGrid> grid;
    Material material;
};
struct Scene {
    std::vector<SceneObject> objects;
    std::vector<Light> lights;
};
struct ImageData {  // parser.rs:79-84
    uint32_t width = 0, height = 0;
    Camera camera;
    Scene scene;
};

constexpr double EPSILON = 1e-5;  // core.rs:10
constexpr uint32_t NO_ID = 0xFFFFFFFFu;

struct ModelHit {
    double t;
    Vec3 normal;
    uint32_t tri;
};

// core.rs:94-178.  `faithful` re-fetches and re-transforms the triangle on every test (model.rs:108-122).
static inline bool ray_intersect(const Model& model, const ModelGrid& grid, const Ray& ray, ModelHit& out, bool faithful,
                                 Counters* cnt) {
    if (cnt) cnt->bbox_tests++;
    double tb;
    if (!model.bounding_box.intersect(ray, tb)) return false;
    const Vec3 P0 = ray.at(tb);
    uint32_t ux, uy, uz;
    grid.closest_cell(P0, ux, uy, uz);
    int ix = (int)ux, iy = (int)uy, iz = (int)uz;
    const int step_x = ray.direction.x > 0.0 ? 1 : -1;
    const int step_y = ray.direction.y > 0.0 ? 1 : -1;
    const int step_z = ray.direction.z > 0.0 ? 1 : -1;
    const int nx = ix + std::max(step_x, 0), ny = iy + std::max(step_y, 0), nz = iz + std::max(step_z, 0);
    const Vec3 nb = grid.cell_box_min((uint32_t)nx, (uint32_t)ny, (uint32_t)nz);
    double t_max_x = (nb.x - P0.x) / ray.direction.x;
    double t_max_y = (nb.y - P0.y) / ray.direction.y;
    double t_max_z = (nb.z - P0.z) / ray.direction.z;
    const Vec3 cs = grid.cell_size();
    const double t_delta_x = (double)step_x * cs.x / ray.direction.x;
    const double t_delta_y = (double)step_y * cs.y / ray.direction.y;
    const double t_delta_z = (double)step_z * cs.z / ray.direction.z;
    bool have = false;
    size_t best_i = 0;
    double best_t = 0.0;
    const int n = (int)grid.n;
    while (ix >= 0 && ix < n && iy >= 0 && iy < n && iz >= 0 && iz < n) {
        if (cnt) cnt->cells_visited++;
        size_t s, e;
        grid.triangles((uint32_t)ix, (uint32_t)iy, (uint32_t)iz, s, e);
        for (size_t k = s; k < e; ++k) {
            const size_t ti = (size_t)grid.triangle_indices[k];
            if (cnt) cnt->tri_tests++;
            double t;
            const bool h = faithful ? model.get_triangle(ti).intersect(ray, t) : model.pre_tri[ti].intersect(ray, t);
            if (h && (!have || t < best_t)) {
                have = true;
                best_i = ti;
                best_t = t;
            }
        }
        if (have) break;  // first cell with any accepted hit (quirk Q8)
        if (t_max_x < t_max_y) {
            if (t_max_x < t_max_z) { ix += step_x; t_max_x += t_delta_x; }
            else                   { iz += step_z; t_max_z += t_delta_z; }
        } else {
            if (t_max_y < t_max_z) { iy += step_y; t_max_y += t_delta_y; }
            else                   { iz += step_z; t_max_z += t_delta_z; }
        }
    }
    if (!have) return false;
    if (cnt) cnt->model_hits++;
    const Triangle tri = faithful ? model.get_triangle(best_i) : model.pre_tri[best_i];
    Vec3 nn[3];
    if (faithful) model.get_triangle_normals(best_i, nn);
    else for (int k = 0; k < 3; ++k) nn[k] = model.pre_nrm[best_i * 3 + k];
    out.t = best_t;
    out.normal = tri.normal_on_point(ray.at(best_t), nn);
    out.tri = (uint32_t)best_i;
    return true;
}

struct RaycastResult {  // core.rs:42-46 (+ ids, which the reference only exposes as &SceneObject)
    uint32_t obj = NO_ID, tri = NO_ID;
    Vec3 hit_point, normal;
    double t = 0.0;
};

// core.rs:68-92 with Solid::hit core.rs:180-213 inlined per kind
static inline bool hit(const Scene& scene, const Ray& ray, RaycastResult& out, bool faithful, Counters* cnt) {
    double closest_t = std::numeric_limits<double>::infinity();
    bool found = false;
    Vec3 best_n = v3(0, 0, 0);
    uint32_t best_obj = NO_ID, best_tri = NO_ID;
    for (uint32_t oi = 0; oi < scene.objects.size(); ++oi) {
        const SceneObject& o = scene.objects[oi];
        double t;
        Vec3 nrm;
        uint32_t tri = NO_ID;
        if (o.kind == SPHERE) {  // core.rs:183-197
            if (cnt) cnt->sphere_tests++;
            const Vec3 oc = ray.origin - o.a;
            const double a = dot(ray.direction, ray.direction);
            const double b = 2.0 * dot(ray.direction, oc);
            const double c = dot(oc, oc) - o.b * o.b;
            const double disc = b * b - 4.0 * a * c;
            if (disc < 0.0) continue;
            t = (-b - std::sqrt(disc)) / (2.0 * a);
            nrm = normalize(ray.at(t) - o.a);
        } else if (o.kind == PLANE) {  // core.rs:198-210
            if (cnt) cnt->plane_tests++;
            const double dv = dot(o.a, ray.direction);
            const Vec3 center = o.a * o.b;
            if (std::fabs(dv) < EPSILON) continue;
            const double d2 = dot(center - ray.origin, o.a);
            t = d2 / dv;
            if (t < EPSILON) continue;
            nrm = o.a;
        } else {
            ModelHit mh;
            if (!ray_intersect(*o.model, *o.grid, ray, mh, faithful, cnt)) continue;
            t = mh.t; nrm = mh.normal; tri = mh.tri;
        }
        if (t <= EPSILON) continue;  // core.rs:75-77 (rejects the whole model, quirk Q12)
        if (t < closest_t) {         // core.rs:79-83 strict: earliest object wins ties
            closest_t = t;
            best_n = nrm;
            best_obj = oi;
            best_tri = tri;
            found = true;
        }
    }
    if (!found) return false;
    out.obj = best_obj; out.tri = best_tri; out.t = closest_t;
    out.hit_point = ray.at(closest_t);
    out.normal = best_n;
    return true;
}

// main.rs:57-132 (+ get_bounce_direction main.rs:50-55).  first: receives the camera ray's hit ids.
static inline Vec3 cast(const Scene& scene, const Ray& ray, const SampleRng& rng, int max_bounces, bool faithful,
                        Counters* cnt, uint32_t* first_obj, uint32_t* first_tri) {
    Vec3 color = v3(0, 0, 0);
    double attenuation = 1.0;
    Ray cur = ray;
    for (int bounce = 0; bounce < max_bounces; ++bounce) {
        if (cnt) (bounce == 0 ? cnt->rays_camera : cnt->rays_bounce)++;
        RaycastResult r;
        const bool h = hit(scene, cur, r, faithful, cnt);
        if (bounce == 0) {
            if (first_obj) *first_obj = h ? r.obj : NO_ID;
            if (first_tri) *first_tri = h ? r.tri : NO_ID;
        }
        if (!h) {
            color = color + v3(1, 1, 1) * attenuation;  // main.rs:127
            break;
        }
        Vec3 diffuse = v3(0, 0, 0);
        const Vec3 normal = dot(cur.direction, r.normal) > 0.001 ? r.normal * -1.0 : r.normal;  // main.rs:68-73
        for (const Light& light : scene.lights) {  // main.rs:78-105
            const Vec3 v = normalize(light.position - r.hit_point);
            const Ray lray{r.hit_point, v};
            const double dl2 = squared_distance(r.hit_point, light.position);
            if (cnt) cnt->rays_shadow++;
            RaycastResult occ;
            if (hit(scene, lray, occ, faithful, cnt)) {
                const double do2 = squared_distance(occ.hit_point, r.hit_point);
                if (do2 <= dl2) continue;
            }
            const double eff = dot(v, normal);
            if (eff > 0.001) {
                const double d = std::fmax(dl2 / (light.radius * light.radius), 1.0);
                const double decay = 1.0 / d;
                diffuse = diffuse + light.color * decay * eff;
            }
        }
        const Material& m = scene.objects[r.obj].material;
        const double diffuse_k = m.type == Metal ? 0.2 : 0.9;   // main.rs:108-111
        const double specular_k = m.type == Metal ? 0.8 : 0.1;  // main.rs:112-115
        const Vec3 bdir = m.type == Metal ? reflect(cur.direction, normal)                      // main.rs:52
                                          : normalize(normal + rng.random_vec((uint32_t)bounce));  // main.rs:53
        color = color + m.color * attenuation * (diffuse * diffuse_k);  // main.rs:118
        cur = Ray{r.hit_point, bdir};
        attenuation *= specular_k;
    }
    return color;
}

enum RenderFlags : uint32_t { FLAG_ZERO_JITTER = 1u };

// one pixel of main.rs:158-170; idx = x + w*y
static inline Vec3 render_pixel(const ImageData& img, uint32_t idx, uint32_t spp, uint64_t seed, uint32_t flags,
                                int max_bounces, bool faithful, Counters* cnt, uint32_t* obj0, uint32_t* tri0) {
    Vec3 px = v3(0, 0, 0);
    const uint32_t w = img.width, h = img.height;
    for (uint32_t s = 0; s < spp; ++s) {
        const uint32_t x = idx % w, y = idx / w;
        SampleRng rng{seed, idx, s};
        double xo = 0.0, yo = 0.0;
        if (!(flags & FLAG_ZERO_JITTER)) rng.jitter(xo, yo);
        const double u = ((double)x + xo - (double)w * 0.5) / (double)w;  // main.rs:165
        const double v = ((double)y + yo - (double)h * 0.5) / (double)h;  // main.rs:166
        const Ray ray = img.camera.shoot_to(u, v);
        const Vec3 c = cast(img.scene, ray, rng, max_bounces, faithful, cnt, s == 0 ? obj0 : nullptr,
                            s == 0 ? tri0 : nullptr);
        px = px + c / (double)spp;  // main.rs:169
    }
    return px;
}

// ---------------------------------------------------------------- Scene DSL parser (parser.rs:88-478)
struct ParseError {
    std::string message;
    uint32_t line = 0, column = 0;
};
class SceneParser {
   public:
    // model paths are opened relative to `base_dir` (the reference opens them relative to the cwd, core.rs:57)
    SceneParser(const std::string& content, const std::string& base_dir, bool allow_missing_normals = false)
        : s_(content), base_(base_dir), synth_(allow_missing_normals) {}
    bool parse_scene(ImageData& out, ParseError& err) {  // parser.rs:438-477
        try {
            match("size");
            const double w = parse_float(), h = parse_float();
            out.camera = parse_camera();
            while (!empty()) {
                const std::string& nt = peek();
                if (nt == "light") out.scene.lights.push_back(parse_light());
                else if (nt == "sphere") out.scene.objects.push_back(parse_sphere());
                else if (nt == "plane") out.scene.objects.push_back(parse_plane());
                else if (nt == "model") out.scene.objects.push_back(parse_model());
                else fail("unexpected token '" + nt + "'");
            }
            out.width = as_u32(w);
            out.height = as_u32(h);
            return true;
        } catch (const ParseError& e) {
            err = e;
            return false;
        }
    }

   private:
    const std::string& s_;
    std::string base_;
    bool synth_;
    std::string buf_;
    uint32_t line_ = 0, col_ = 0;
    size_t idx_ = 0;

    [[noreturn]] void fail(const std::string& m) { throw ParseError{m, line_, col_}; }
    bool empty() const { return idx_ >= s_.size(); }
    int cur() const { return empty() ? -1 : (unsigned char)s_[idx_]; }
    void advance() {  // parser.rs:105-115
        if (empty()) return;
        if (s_[idx_] == '\n') { line_++; col_ = 0; } else { col_++; }
        idx_++;
    }
    void eat_spaces() {  // parser.rs:126-139
        while (!empty()) {
            const int c = cur();
            if (c == '#') { while (!empty() && cur() != '\n') advance(); }
            else if (!std::isspace(c)) break;
            advance();
        }
    }
    int enq(std::string& r) {  // parser.rs:157-166
        if (!empty()) { r.push_back(s_[idx_]); advance(); }
        return empty() ? ' ' : cur();
    }
    std::string pop() {  // parser.rs:141-214
        if (!buf_.empty()) { std::string r = buf_; buf_.clear(); return r; }
        eat_spaces();
        std::string r;
        if (empty()) return r;
        int c = cur();
        if (c == ',' || c == '(' || c == ')' || c == ':' || c == '>') {
            advance();
            r.push_back((char)c);
        } else if (c == '"') {
            enq(r);
            bool in_string = true;
            for (;;) {
                const size_t before = idx_;
                c = enq(r);
                if (!in_string) break;
                in_string = c != '"';
                if (idx_ == before && empty()) break;  // unterminated string: stop at EOF
            }
        } else if (c == '.' || c == '+' || c == '-' || std::isdigit(c)) {
            if (c == '+' || c == '-') c = enq(r);
            while (std::isdigit(c)) c = enq(r);
            if (c == '.') {
                c = enq(r);
                while (std::isdigit(c)) c = enq(r);
            }
        } else {
            while (std::isalpha(c)) c = enq(r);
        }
        return r;
    }
    const std::string& peek() {  // parser.rs:216-222
        if (buf_.empty()) buf_ = pop();
        return buf_;
    }
    double parse_float() {  // parser.rs:228-240 (Rust str::parse::<f64>: no exponent can reach here)
        const std::string t = pop();
        bool digit = false;
        for (char ch : t) digit |= (ch >= '0' && ch <= '9');
        if (!digit) fail("error parsing file: cannot interp '" + t + "' as a float");
        char* end = nullptr;
        const double v = std::strtod(t.c_str(), &end);
        if (end == t.c_str() || *end != '\0') fail("error parsing file: cannot interp '" + t + "' as a float");
        return v;
    }
    void match(const char* lex) {  // parser.rs:242-256
        const std::string t = pop();
        if (t != lex) fail(std::string("error parsing the scene file: expected '") + lex + "', getting '" + t + "' instead");
    }
    bool maybe(const char* lex) {  // parser.rs:258-268
        if (peek() == lex) { pop(); return true; }
        return false;
    }
    Vec3 parse_vec3() {  // parser.rs:279-288
        match("(");
        const double x = parse_float();
        match(",");
        const double y = parse_float();
        match(",");
        const double z = parse_float();
        match(")");
        return v3(x, y, z);
    }
    Vec3 parse_color() {  // parser.rs:291-316
        if (maybe("red")) return v3(1, 0, 0);
        if (maybe("blue")) return v3(0, 0, 1);
        if (maybe("green")) return v3(0, 1, 0);
        if (maybe("white")) return v3(1, 1, 1);
        if (maybe("black")) return v3(0, 0, 0);
        if (maybe("cyan")) return v3(0, 1, 1);
        if (maybe("violet")) return v3(1, 0, 1);
        if (maybe("fuchsia")) return v3(0.96, 0.0, 96.0);  // sic (quirk Q15)
        if (maybe("yellow")) return v3(1, 1, 0);
        if (maybe("orange")) return v3(0.98, 0.45, 0.02);
        return parse_vec3();
    }
    Material parse_material() {  // parser.rs:318-332 -- both prefixes give Metal (quirk Q15)
        MaterialType t = Plastic;
        if (maybe("metal")) { t = Metal; match(":"); }
        else if (maybe("plastic")) { t = Metal; match(":"); }
        const Vec3 c = parse_color();
        return Material{c, t};
    }
    SceneObject parse_sphere() {  // parser.rs:334-343
        match("sphere");
        SceneObject o;
        o.kind = SPHERE;
        o.a = parse_vec3();
        o.b = parse_float();
        o.material = parse_material();
        return o;
    }
    SceneObject parse_plane() {  // parser.rs:345-354
        match("plane");
        SceneObject o;
        o.kind = PLANE;
        o.a = parse_vec3();
        o.b = parse_float();
        o.material = parse_material();
        return o;
    }
    std::string parse_string() {  // parser.rs:356-363
        std::string t = pop();
        if (t.size() < 2) fail("expected a quoted string");
        return t.substr(1, t.size() - 2);
    }
    Camera parse_camera() {  // parser.rs:365-380
        if (maybe("camera")) {
            Vec3 pos = v3(0, 0, 0);
            if (maybe("from")) pos = parse_vec3();
            const Vec3 pt = maybe("to") ? parse_vec3() : pos + v3(0, 0, 1);
            return Camera::look_at(pos, pt);
        }
        return Camera::at(v3(0, 0, 0));
    }
    Mat4 parse_trasformation() {  // parser.rs:382-408
        Mat4 t = Mat4::identity();
        while (maybe(">")) {
            const std::string nt = peek();
            Mat4 next;
            if (nt == "scale") { pop(); next = Mat4::scale(parse_float()); }
            else if (nt == "translate") { pop(); next = Mat4::translate(parse_vec3()); }
            else if (nt == "rotate") { pop(); const Vec3 ax = parse_vec3(); const double an = parse_float(); next = Mat4::rotate(ax, an); }
            else fail("unexpected token while parsing trasform");
            t = t.then(next);
        }
        return t;
    }
    SceneObject parse_model() {  // parser.rs:410-425 + core.rs:56-65
        match("model");
        const std::string path = parse_string();
        const Material mat = parse_material();
        const Mat4 tr = parse_trasformation();
        std::string full = path;
        if (!base_.empty() && !path.empty() && path[0] != '/') full = base_ + "/" + path;
        Obj obj;
        std::string e;
        if (!load_obj(full, obj, e, synth_)) fail("Cannot load model  \"" + path + "\" (" + e + ")");
        auto model = std::make_shared<Model>();
        if (!Model::make(std::move(obj), tr, *model, e)) fail(e);
        auto grid = std::make_shared<ModelGrid>(model->create_grid(32));  // core.rs:62
        model->precompute();
        SceneObject o;
        o.kind = MODEL;
        o.model = model;
        o.grid = grid;
        o.material = mat;
        return o;
    }
    Light parse_light() {  // parser.rs:427-436
        match("light");
        const Vec3 p = parse_vec3();
        const Vec3 c = parse_color();
        return Light{p, c, 2.0};
    }
};

}  // namespace orc
