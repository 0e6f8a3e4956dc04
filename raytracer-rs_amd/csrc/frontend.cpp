// frontend.cpp -- host-side scene front end of the MI355X back end (pure host code, no GPU).
//
// In the target deployment this work stays in Rust (parser.rs, obj-rs, mat4.rs, Model::new,
// create_grid) and a thin FFI crate hands the flattened scene over.  No Rust
// toolchain exists in this image, so this file is the C++ stand-in that produces the SAME
// flattened buffers from the same scene files: it is what bench.py and the tests drive the
// library with.  It is organised as a flattening pipeline (lexer -> scene records -> per-model
// vertex transform -> CSR grid by count/fill -> SoA), not as the reference's object graph.
//
// Citations are file:line under /root/reference/src/raytracing/ unless a path is given.
// Arithmetic that feeds the GPU must keep the reference's operation order:
// this translation unit is built with -ffp-contract=off.
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rtm_internal.h"

namespace {

struct D3 {
    double v[3];
};
struct M44 {
    double a[16];
};

// ----- 4x4 helpers (math/mat4.rs) --------------------------------------------------------------
M44 m_identity() { return M44{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}}; }   // mat4.rs:9-15
M44 m_scale(double f) { return M44{{f, 0, 0, 0, 0, f, 0, 0, 0, 0, f, 0, 0, 0, 0, 1}}; }  // mat4.rs:17-24
M44 m_translate(const D3& o) {                                                            // mat4.rs:26-33
    return M44{{1, 0, 0, o.v[0], 0, 1, 0, o.v[1], 0, 0, 1, o.v[2], 0, 0, 0, 1}};
}
// mat4.rs:35-60.  NOT a true rotation: the off-diagonal sine terms are kept exactly as written there.
M44 m_rotate(const D3& axis, double angle) {
    const double l = std::sqrt(axis.v[0] * axis.v[0] + axis.v[1] * axis.v[1] + axis.v[2] * axis.v[2]);
    const double ux = axis.v[0] / l, uy = axis.v[1] / l, uz = axis.v[2] / l;
    const double ct = std::cos(angle), st = std::sin(angle), omc = 1.0 - ct;
    M44 r{};
    r.a[0] = ct + (ux * ux) * omc;   r.a[1] = ux * uy * omc - uz * st;  r.a[2] = ux * uz * omc - uy * st;
    r.a[4] = ux * uy * omc - ux * st; r.a[5] = ct + (uy * uy) * omc;    r.a[6] = uz * uy * omc - ux * st;
    r.a[8] = ux * uz * omc - uy * st; r.a[9] = uz * uy * omc + ux * st; r.a[10] = ct + uz * uz * omc;
    r.a[15] = 1.0;
    return r;
}
// mat4.rs:146-222: `self.then(other)` = other * self, each entry summed left to right.
M44 m_then(const M44& self, const M44& other) {
    M44 r;
    for (int row = 0; row < 4; ++row)
        for (int col = 0; col < 4; ++col) {
            const double* o = &other.a[row * 4];
            r.a[row * 4 + col] = o[0] * self.a[col] + o[1] * self.a[4 + col] + o[2] * self.a[8 + col] + o[3] * self.a[12 + col];
        }
    return r;
}
// 3x3 minor obtained by striking row rr / column cc, expanded like mat4.rs:63-67
double m_minor(const M44& m, int rr, int cc) {
    double s[9];
    int k = 0;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            if (r != rr && c != cc) s[k++] = m.a[r * 4 + c];
    return s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) + s[2] * (s[3] * s[7] - s[4] * s[6]);
}
// mat4.rs:89-121 followed by transpose (mat4.rs:123-144): the normal transform of model.rs:93-96
bool m_inverse_transpose(const M44& m, M44& out) {
    double det = 0.0;
    for (int i = 0; i < 4; ++i) det += ((i & 1) ? -1.0 : 1.0) * m.a[i] * m_minor(m, 0, i);
    if (det == 0.0) return false;
    // inverse[c][r] = cofactor(r,c)/det  =>  transpose(inverse)[r][c] = cofactor(r,c)/det
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out.a[r * 4 + c] = ((((r + c) & 1) ? -1.0 : 1.0) * m_minor(m, r, c)) / det;
    return true;
}
// mat4.rs:224-230 (homogeneous divide kept)
D3 m_apply(const M44& m, const float p[3]) {
    const double x = p[0], y = p[1], z = p[2];  // vec3.rs:18-22 f32 -> f64
    const double rx = m.a[0] * x + m.a[1] * y + m.a[2] * z + m.a[3];
    const double ry = m.a[4] * x + m.a[5] * y + m.a[6] * z + m.a[7];
    const double rz = m.a[8] * x + m.a[9] * y + m.a[10] * z + m.a[11];
    const double rw = m.a[12] * x + m.a[13] * y + m.a[14] * z + m.a[15];
    return D3{{rx / rw, ry / rw, rz / rw}};
}

// Rust `as usize` on f64: saturating, NaN -> 0
uint64_t sat_usize(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)v;
}
uint32_t sat_u32(double v) {
    if (!(v > 0.0)) return 0;
    if (v >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)v;
}

// ----- lexer (parser.rs:97-222): lazy, one token of look-ahead, line:column of the cursor ----------
struct FrontError {
    int code;
    std::string msg;
};

class Lexer {
   public:
    explicit Lexer(const std::string& text) : t_(text) {}
    bool exhausted() const { return pos_ >= t_.size(); }  // parser.rs:101-103 (cursor only, not the look-ahead)
    const std::string& look() {                            // parser.rs:216-222
        if (ahead_.empty()) ahead_ = scan();
        return ahead_;
    }
    std::string take() {  // parser.rs:141-149
        if (!ahead_.empty()) {
            std::string r;
            r.swap(ahead_);
            return r;
        }
        return scan();
    }
    bool accept(const char* lexeme) {  // parser.rs:258-268
        if (look() == lexeme) {
            take();
            return true;
        }
        return false;
    }
    [[noreturn]] void die(const std::string& m) const {  // parser.rs:224-226 + Display :67-75
        throw FrontError{RTM_ERR_PARSE, m + " at " + std::to_string(line_) + ":" + std::to_string(col_)};
    }
    void expect(const char* lexeme) {  // parser.rs:242-256
        const std::string got = take();
        if (got != lexeme) die(std::string("error parsing the scene file: expected '") + lexeme + "', getting '" + got + "' instead");
    }
    double number() {  // parser.rs:228-240
        const std::string tok = take();
        bool any_digit = false;
        for (char c : tok) any_digit = any_digit || (c >= '0' && c <= '9');
        char* end = nullptr;
        const double v = any_digit ? std::strtod(tok.c_str(), &end) : 0.0;
        if (!any_digit || *end != '\0') die("error parsing file: cannot interp '" + tok + "' as a float");
        return v;
    }
    D3 triple() {  // parser.rs:279-288
        D3 r;
        expect("(");
        r.v[0] = number();
        expect(",");
        r.v[1] = number();
        expect(",");
        r.v[2] = number();
        expect(")");
        return r;
    }
    std::string quoted() {  // parser.rs:356-363
        std::string tok = take();
        if (tok.size() < 2) die("expected a string");
        return tok.substr(1, tok.size() - 2);
    }

   private:
    const std::string& t_;
    std::string ahead_;
    size_t pos_ = 0;
    uint32_t line_ = 0, col_ = 0;

    void step() {  // parser.rs:105-115
        if (t_[pos_] == '\n') { ++line_; col_ = 0; } else { ++col_; }
        ++pos_;
    }
    int at() const { return exhausted() ? ' ' : (unsigned char)t_[pos_]; }
    std::string scan() {
        // blanks and #-comments (parser.rs:126-139)
        while (!exhausted()) {
            if (t_[pos_] == '#') { while (!exhausted() && t_[pos_] != '\n') step(); if (!exhausted()) step(); continue; }
            if (!std::isspace((unsigned char)t_[pos_])) break;
            step();
        }
        std::string out;
        if (exhausted()) return out;
        auto shift = [&]() { out.push_back(t_[pos_]); step(); };
        const int c = at();
        if (std::strchr(",():>", c)) {  // parser.rs:170-173
            shift();
        } else if (c == '"') {  // parser.rs:174-186, no escapes, closing quote kept
            shift();
            while (!exhausted() && t_[pos_] != '"') shift();
            if (!exhausted()) shift();
        } else if (c == '.' || c == '+' || c == '-' || std::isdigit(c)) {  // parser.rs:188-205, no exponent
            if (c == '+' || c == '-') shift();
            while (std::isdigit(at()) && !exhausted()) shift();
            if (!exhausted() && t_[pos_] == '.') {
                shift();
                while (std::isdigit(at()) && !exhausted()) shift();
            }
        } else {  // parser.rs:207-211
            while (!exhausted() && std::isalpha((unsigned char)t_[pos_])) shift();
        }
        return out;
    }
};

// ----- scene records ------------------------------------------------------------------------------
struct MaterialRec {
    double rgb[3];
    uint32_t type;
};
struct ModelRec {
    double center[3], half[3];
    uint32_t cells = 0, n_tris = 0, mat = 0, obj = 0;
    std::vector<uint32_t> offsets, tri_idx;
    std::vector<double> pos[9], nrm[9];
};

const struct { const char* name; double rgb[3]; } kNamedColours[] = {  // parser.rs:293-312 (fuchsia's 96.0 is the reference's)
    {"red", {1, 0, 0}},    {"blue", {0, 0, 1}},   {"green", {0, 1, 0}},         {"white", {1, 1, 1}},  {"black", {0, 0, 0}},
    {"cyan", {0, 1, 1}},   {"violet", {1, 0, 1}}, {"fuchsia", {0.96, 0.0, 96.0}}, {"yellow", {1, 1, 0}}, {"orange", {0.98, 0.45, 0.02}},
};

}  // namespace

struct rtm_frontend {
    uint32_t width = 0, height = 0;
    rtm_camera cam{};
    uint32_t n_objects = 0;
    std::vector<double> sph[4], pl[4], li[7], mat_rgb[3];
    std::vector<uint32_t> sph_mat, sph_obj, pl_mat, pl_obj, mat_type;
    std::vector<ModelRec> models;
    std::vector<rtm_model> model_views;
    rtm_scene view{};
    std::string base_dir;
    bool synth_normals = false;

    uint32_t push_material(const MaterialRec& m) {
        for (int c = 0; c < 3; ++c) mat_rgb[c].push_back(m.rgb[c]);
        mat_type.push_back(m.type);
        return (uint32_t)mat_type.size() - 1;
    }
    void refresh_view();
};

namespace {

void set_camera_default(rtm_camera& c) {  // camera.rs:13-26
    const double p[3] = {0, 0, 0}, f[3] = {0, 0, 1}, u[3] = {0, 1, 0}, r[3] = {1, 0, 0};
    std::memcpy(c.position, p, 24); std::memcpy(c.forward, f, 24); std::memcpy(c.up, u, 24); std::memcpy(c.right, r, 24);
    c.zoom = 1.0;
}
void cross3(const double a[3], const double b[3], double o[3]) {  // vec3.rs:183-189
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
void unit3(double a[3]) {  // vec3.rs:213-215: three divisions by the length
    const double l = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    a[0] /= l; a[1] /= l; a[2] /= l;
}
void set_camera_look_at(rtm_camera& c, const D3& from, const D3& to) {  // camera.rs:30-43
    const double world_up[3] = {0, 1, 0};
    for (int i = 0; i < 3; ++i) { c.position[i] = from.v[i]; c.forward[i] = to.v[i] - from.v[i]; }
    unit3(c.forward);
    cross3(world_up, c.forward, c.right);
    unit3(c.right);
    cross3(c.forward, c.right, c.up);
    unit3(c.up);
    c.zoom = 1.0;
}

MaterialRec read_material(Lexer& lx) {  // parser.rs:318-332: `metal:` AND `plastic:` both select Metal
    MaterialRec m{};
    m.type = 0;
    if (lx.accept("metal") || lx.accept("plastic")) {
        m.type = 1;
        lx.expect(":");
    }
    for (const auto& nc : kNamedColours)  // parser.rs:291-316
        if (lx.accept(nc.name)) {
            std::memcpy(m.rgb, nc.rgb, sizeof m.rgb);
            return m;
        }
    const D3 c = lx.triple();
    std::memcpy(m.rgb, c.v, sizeof m.rgb);
    return m;
}

M44 read_transform_chain(Lexer& lx) {  // parser.rs:382-408
    M44 acc = m_identity();
    while (lx.accept(">")) {
        const std::string op = lx.look();
        M44 next;
        if (op == "scale") { lx.take(); next = m_scale(lx.number()); }
        else if (op == "translate") { lx.take(); next = m_translate(lx.triple()); }
        else if (op == "rotate") { lx.take(); const D3 ax = lx.triple(); const double an = lx.number(); next = m_rotate(ax, an); }
        else lx.die("unexpected token while parsing trasform");
        acc = m_then(acc, next);
    }
    return acc;
}

// ----- OBJ reading: stand-in for obj-rs 0.7.1 `load_obj::<Vertex, _, u16>` (core.rs:58) -------------
// positions / normals parsed as f32; triangles only; every corner needs a normal; vertices are unique
// (position index, normal index) pairs numbered by first appearance.
struct Mesh {
    std::vector<float> vpos, vnrm;  // unique vertices, xyz
    std::vector<uint32_t> corner;   // 3 per triangle, into the unique vertices
};

void read_obj(const std::string& path, bool synth_normals, Mesh& mesh) {
    std::ifstream in(path);
    if (!in) throw FrontError{RTM_ERR_IO, "cannot open " + path};
    std::vector<float> P, N;
    std::unordered_map<uint64_t, uint32_t> seen;
    std::string line;
    while (std::getline(in, line)) {
        const char* s = line.c_str();
        while (*s == ' ' || *s == '\t') ++s;
        if ((s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) || (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t'))) {
            std::vector<float>& dst = s[1] == 'n' ? N : P;
            char* e = const_cast<char*>(s + (s[1] == 'n' ? 2 : 1));
            for (int k = 0; k < 3; ++k) {
                char* e2;
                const float f = std::strtof(e, &e2);
                if (e2 == e) throw FrontError{RTM_ERR_IO, "malformed vertex in " + path};
                dst.push_back(f);
                e = e2;
            }
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            long pi[4], ni[4];
            bool hn[4];
            int nc = 0;
            const char* q = s + 1;
            for (;;) {
                while (*q == ' ' || *q == '\t' || *q == '\r') ++q;
                if (!*q) break;
                if (nc == 3) throw FrontError{RTM_ERR_IO, "Model should be triangulated first to be loaded properly"};
                char* e;
                pi[nc] = std::strtol(q, &e, 10);
                ni[nc] = 0;
                hn[nc] = false;
                q = e;
                if (*q == '/') {
                    ++q;
                    if (*q != '/') { std::strtol(q, &e, 10); q = e; }  // texture index, unused
                    if (*q == '/') {
                        ++q;
                        if (*q && *q != ' ' && *q != '\t' && *q != '\r') { ni[nc] = std::strtol(q, &e, 10); q = e; hn[nc] = true; }
                    }
                }
                ++nc;
            }
            if (nc != 3) throw FrontError{RTM_ERR_IO, "Model should be triangulated first to be loaded properly"};
            const long np = (long)P.size() / 3, nn = (long)N.size() / 3;
            long rp[3], rn[3];
            for (int k = 0; k < 3; ++k) {
                rp[k] = pi[k] > 0 ? pi[k] - 1 : np + pi[k];
                if (rp[k] < 0 || rp[k] >= np) throw FrontError{RTM_ERR_IO, "face index out of range in " + path};
                if (!hn[k] && !synth_normals) throw FrontError{RTM_ERR_IO, "Tried to extract normal data which are not contained in the model"};
                if (hn[k]) {
                    rn[k] = ni[k] > 0 ? ni[k] - 1 : nn + ni[k];
                    if (rn[k] < 0 || rn[k] >= nn) throw FrontError{RTM_ERR_IO, "face index out of range in " + path};
                }
            }
            if (synth_normals) {  // not reference behaviour: geometric face normal on three fresh vertices
                double a[3], b[3], g[3];
                for (int c = 0; c < 3; ++c) {
                    a[c] = (double)P[rp[1] * 3 + c] - (double)P[rp[0] * 3 + c];
                    b[c] = (double)P[rp[2] * 3 + c] - (double)P[rp[0] * 3 + c];
                }
                cross3(a, b, g);
                unit3(g);
                for (int k = 0; k < 3; ++k) {
                    mesh.corner.push_back((uint32_t)(mesh.vpos.size() / 3));
                    for (int c = 0; c < 3; ++c) { mesh.vpos.push_back(P[rp[k] * 3 + c]); mesh.vnrm.push_back((float)g[c]); }
                }
                continue;
            }
            for (int k = 0; k < 3; ++k) {
                const uint64_t key = ((uint64_t)rp[k] << 32) | (uint64_t)rn[k];
                auto it = seen.find(key);
                if (it == seen.end()) {
                    const uint32_t id = (uint32_t)(mesh.vpos.size() / 3);
                    for (int c = 0; c < 3; ++c) { mesh.vpos.push_back(P[rp[k] * 3 + c]); mesh.vnrm.push_back(N[rn[k] * 3 + c]); }
                    it = seen.emplace(key, id).first;
                }
                mesh.corner.push_back(it->second);
            }
        }
    }
    if (!synth_normals && mesh.vpos.size() / 3 > 65536) throw FrontError{RTM_ERR_IO, "vertex count does not fit the reference's u16 indices"};
}

// ----- flatten one model: transform once per unique vertex, bbox, CSR grid -------------------------
void flatten_model(const Mesh& mesh, const M44& xf, uint32_t cells, ModelRec& out) {
    M44 nxf;
    if (!m_inverse_transpose(xf, nxf))  // model.rs:93-96 (the reference panics here)
        throw FrontError{RTM_ERR_INVALID, "Error: the provided matrix to transform the model is not invertible"};
    const size_t nv = mesh.vpos.size() / 3;
    std::vector<D3> wp(nv), wn(nv);
    const double inf = std::numeric_limits<double>::infinity();
    double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (size_t i = 0; i < nv; ++i) {
        wp[i] = m_apply(xf, &mesh.vpos[i * 3]);  // model.rs:108-122, hoisted: a pure function of the vertex
        D3 n = m_apply(nxf, &mesh.vnrm[i * 3]);  // model.rs:125-139 (with the /w of mat4.rs:229)
        unit3(n.v);
        wn[i] = n;
        for (int c = 0; c < 3; ++c) {  // model.rs:296-323
            if (hi[c] < wp[i].v[c]) hi[c] = wp[i].v[c];
            if (lo[c] > wp[i].v[c]) lo[c] = wp[i].v[c];
        }
    }
    double bmin[3], bmax[3], cell[3];
    for (int c = 0; c < 3; ++c) {
        out.center[c] = (lo[c] + hi[c]) * 0.5;  // box3D.rs:24-29
        out.half[c] = (hi[c] - lo[c]) * 0.5;
        bmin[c] = out.center[c] - out.half[c];  // box3D.rs:55-61
        bmax[c] = out.center[c] + out.half[c];
        cell[c] = (bmax[c] - bmin[c]) / (double)cells;  // model.rs:151
    }
    const uint32_t nt = (uint32_t)(mesh.corner.size() / 3);
    out.cells = cells;
    out.n_tris = nt;
    for (int a = 0; a < 9; ++a) { out.pos[a].resize(nt); out.nrm[a].resize(nt); }
    for (uint32_t t = 0; t < nt; ++t)
        for (int k = 0; k < 3; ++k) {
            const uint32_t vi = mesh.corner[t * 3 + k];
            for (int c = 0; c < 3; ++c) { out.pos[k * 3 + c][t] = wp[vi].v[c]; out.nrm[k * 3 + c][t] = wn[vi].v[c]; }
        }
    // Triangle -> cell range (model.rs:152-167).  The per-triangle box is grown with the reference's
    // Box3::include (box3D.rs:31-45), whose centre moves by (d -/+ h')/2 with the ALREADY UPDATED half
    // extent h' -- i.e. by a quarter of what an exact AABB needs -- so the box is not the true AABB.
    // The binning must reproduce that, or the cell lists (and every hit id) differ from the reference.
    std::vector<uint32_t> range(nt * 6);
    const size_t n3 = (size_t)cells * cells * cells;
    std::vector<uint32_t> count(n3 + 1, 0);
    for (uint32_t t = 0; t < nt; ++t) {
        for (int c = 0; c < 3; ++c) {
            double ctr = out.pos[c][t], h = 0.0;  // Box3::from_single_point(v0), box3D.rs:17-22
            for (int k = 1; k < 3; ++k) {
                const double d = out.pos[k * 3 + c][t] - ctr;
                if (std::fabs(d) > h) {
                    h = (std::fabs(d) + h) / 2.0;
                    ctr += (d - h * (std::signbit(d) ? -1.0 : 1.0)) / 2.0;
                }
            }
            const uint64_t a = sat_usize(((ctr - h) - bmin[c]) / cell[c]);  // model.rs:159-162
            const uint64_t b = sat_usize(((ctr + h) - bmin[c]) / cell[c]);  // model.rs:164-167
            range[t * 6 + c] = (uint32_t)(a < cells - 1 ? a : cells - 1);
            range[t * 6 + 3 + c] = (uint32_t)(b < cells - 1 ? b : cells - 1);
        }
        for (uint32_t z = range[t * 6 + 2]; z <= range[t * 6 + 5]; ++z)
            for (uint32_t y = range[t * 6 + 1]; y <= range[t * 6 + 4]; ++y)
                for (uint32_t x = range[t * 6]; x <= range[t * 6 + 3]; ++x) count[((size_t)z * cells + y) * cells + x + 1]++;
    }
    // inclusive prefix sums = ModelGrid.offset_array (model.rs:185-193); lists ascend by triangle index
    // because triangles are visited in order (model.rs:152,179).
    for (size_t i = 0; i < n3; ++i) count[i + 1] += count[i];
    out.offsets.assign(count.begin() + 1, count.end());
    out.tri_idx.resize(count[n3]);
    std::vector<uint32_t> cursor(count.begin(), count.end() - 1);
    for (uint32_t t = 0; t < nt; ++t)
        for (uint32_t z = range[t * 6 + 2]; z <= range[t * 6 + 5]; ++z)
            for (uint32_t y = range[t * 6 + 1]; y <= range[t * 6 + 4]; ++y)
                for (uint32_t x = range[t * 6]; x <= range[t * 6 + 3]; ++x) out.tri_idx[cursor[((size_t)z * cells + y) * cells + x]++] = t;
}

void build_from_text(const std::string& text, rtm_frontend& fe) {  // parser.rs:438-477
    Lexer lx(text);
    lx.expect("size");  // parser.rs:270-276
    const double w = lx.number(), h = lx.number();
    if (lx.accept("camera")) {  // parser.rs:365-380
        D3 from{{0, 0, 0}};
        if (lx.accept("from")) from = lx.triple();
        D3 to{{from.v[0] + 0.0, from.v[1] + 0.0, from.v[2] + 1.0}};
        if (lx.accept("to")) to = lx.triple();
        set_camera_look_at(fe.cam, from, to);
    } else {
        set_camera_default(fe.cam);
    }
    while (!lx.exhausted()) {
        const std::string kw = lx.look();
        if (kw == "light") {  // parser.rs:427-436, radius fixed at 2.0
            lx.take();
            const D3 p = lx.triple();
            const MaterialRec col = [&] {
                MaterialRec m{};
                for (const auto& nc : kNamedColours)
                    if (lx.accept(nc.name)) { std::memcpy(m.rgb, nc.rgb, sizeof m.rgb); return m; }
                const D3 c = lx.triple();
                std::memcpy(m.rgb, c.v, sizeof m.rgb);
                return m;
            }();
            for (int c = 0; c < 3; ++c) { fe.li[c].push_back(p.v[c]); fe.li[3 + c].push_back(col.rgb[c]); }
            fe.li[6].push_back(2.0);
        } else if (kw == "sphere" || kw == "plane") {  // parser.rs:334-354
            lx.take();
            const D3 a = lx.triple();
            const double b = lx.number();
            const uint32_t mi = fe.push_material(read_material(lx));
            auto* arr = kw == "sphere" ? fe.sph : fe.pl;
            for (int c = 0; c < 3; ++c) arr[c].push_back(a.v[c]);
            arr[3].push_back(b);
            (kw == "sphere" ? fe.sph_mat : fe.pl_mat).push_back(mi);
            (kw == "sphere" ? fe.sph_obj : fe.pl_obj).push_back(fe.n_objects++);
        } else if (kw == "model") {  // parser.rs:410-425, core.rs:56-65
            lx.take();
            const std::string rel = lx.quoted();
            const MaterialRec mat = read_material(lx);
            const M44 xf = read_transform_chain(lx);
            std::string path = rel;
            if (!fe.base_dir.empty() && !rel.empty() && rel[0] != '/') path = fe.base_dir + "/" + rel;
            Mesh mesh;
            try {
                read_obj(path, fe.synth_normals, mesh);
            } catch (const FrontError& e) {
                if (e.code == RTM_ERR_IO) lx.die("Cannot load model  \"" + rel + "\" (" + e.msg + ")");
                throw;
            }
            fe.models.emplace_back();
            flatten_model(mesh, xf, 32, fe.models.back());  // 32: core.rs:62
            fe.models.back().mat = fe.push_material(mat);
            fe.models.back().obj = fe.n_objects++;
        } else {
            lx.die("unexpected token '" + kw + "'");
        }
    }
    fe.width = sat_u32(w);  // parser.rs:472-473
    fe.height = sat_u32(h);
}

}  // namespace

void rtm_frontend::refresh_view() {
    model_views.resize(models.size());
    for (size_t i = 0; i < models.size(); ++i) {
        const ModelRec& m = models[i];
        rtm_model& v = model_views[i];
        std::memcpy(v.bbox_center, m.center, 24);
        std::memcpy(v.bbox_half, m.half, 24);
        v.cells_per_side = m.cells;
        v.n_tris = m.n_tris;
        v.offsets = m.offsets.data();
        v.tri_idx = m.tri_idx.data();
        v.n_tri_idx = m.tri_idx.size();
        for (int a = 0; a < 9; ++a) { v.pos[a] = m.pos[a].data(); v.nrm[a] = m.nrm[a].data(); }
        v.mat_idx = m.mat;
        v.obj_idx = m.obj;
    }
    view.camera = cam;
    view.n_objects = n_objects;
    view.n_spheres = (uint32_t)sph_obj.size();
    view.sph_cx = sph[0].data(); view.sph_cy = sph[1].data(); view.sph_cz = sph[2].data(); view.sph_r = sph[3].data();
    view.sph_mat = sph_mat.data(); view.sph_obj = sph_obj.data();
    view.n_planes = (uint32_t)pl_obj.size();
    view.pl_nx = pl[0].data(); view.pl_ny = pl[1].data(); view.pl_nz = pl[2].data(); view.pl_d = pl[3].data();
    view.pl_mat = pl_mat.data(); view.pl_obj = pl_obj.data();
    view.n_models = (uint32_t)models.size();
    view.models = model_views.data();
    view.n_lights = (uint32_t)li[0].size();
    view.li_px = li[0].data(); view.li_py = li[1].data(); view.li_pz = li[2].data();
    view.li_r = li[3].data(); view.li_g = li[4].data(); view.li_b = li[5].data(); view.li_radius = li[6].data();
    view.n_materials = (uint32_t)mat_type.size();
    view.mat_r = mat_rgb[0].data(); view.mat_g = mat_rgb[1].data(); view.mat_b = mat_rgb[2].data();
    view.mat_type = mat_type.data();
}

extern "C" {

int rtm_frontend_load(const char* scene_path, const char* scene_text, const char* base_dir, uint32_t flags, rtm_frontend** out) {
    if (!out || (!scene_path && !scene_text)) return rtm_internal_fail(RTM_ERR_INVALID, "rtm_frontend_load: null argument");
    *out = nullptr;
    std::string text;
    if (scene_text) {
        text = scene_text;
    } else {
        std::ifstream in(scene_path);
        if (!in) return rtm_internal_fail(RTM_ERR_IO, std::string("cannot open scene file ") + scene_path);
        std::stringstream ss;
        ss << in.rdbuf();
        text = ss.str();
    }
    auto* fe = new rtm_frontend();
    fe->base_dir = base_dir ? base_dir : "";
    fe->synth_normals = (flags & 1u) != 0;
    try {
        build_from_text(text, *fe);
        fe->refresh_view();
    } catch (const FrontError& e) {
        delete fe;
        return rtm_internal_fail(e.code, e.msg);
    } catch (const std::exception& e) {
        delete fe;
        return rtm_internal_fail(RTM_ERR_INVALID, e.what());
    }
    *out = fe;
    return RTM_OK;
}

int rtm_frontend_add_triangles(rtm_frontend* fe, uint64_t n_tris, const float* pos9, const float* nrm9, const double* color3,
                               uint32_t mat_type, const double* transform16) {
    if (!fe || !pos9 || !nrm9 || !color3 || n_tris == 0 || n_tris > 0x7FFFFFFFull / 3)
        return rtm_internal_fail(RTM_ERR_INVALID, "rtm_frontend_add_triangles: bad argument");
    const size_t models_before = fe->models.size();
    const uint32_t objects_before = fe->n_objects;
    auto roll_back = [&]() {  // a failure half way must leave the scene and its C view as they were
        while (fe->models.size() > models_before) fe->models.pop_back();
        fe->n_objects = objects_before;
        fe->refresh_view();
    };
    try {
        Mesh mesh;
        mesh.vpos.assign(pos9, pos9 + n_tris * 9);
        mesh.vnrm.assign(nrm9, nrm9 + n_tris * 9);
        mesh.corner.resize(n_tris * 3);
        for (uint64_t i = 0; i < n_tris * 3; ++i) mesh.corner[i] = (uint32_t)i;
        M44 xf = m_identity();
        if (transform16) std::memcpy(xf.a, transform16, sizeof xf.a);
        fe->models.emplace_back();
        flatten_model(mesh, xf, 32, fe->models.back());
        MaterialRec m{};
        std::memcpy(m.rgb, color3, sizeof m.rgb);
        m.type = mat_type ? 1u : 0u;
        fe->models.back().mat = fe->push_material(m);
        fe->models.back().obj = fe->n_objects++;
        fe->refresh_view();
    } catch (const FrontError& e) {
        roll_back();
        return rtm_internal_fail(e.code, e.msg);
    } catch (const std::exception& e) {  // (e.g. std::bad_alloc)
        roll_back();
        return rtm_internal_fail(RTM_ERR_INVALID, e.what());
    }
    return RTM_OK;
}

const rtm_scene* rtm_frontend_scene(const rtm_frontend* fe) { return fe ? &fe->view : nullptr; }

int rtm_frontend_size(const rtm_frontend* fe, uint32_t* width, uint32_t* height) {
    if (!fe || !width || !height) return rtm_internal_fail(RTM_ERR_INVALID, "rtm_frontend_size: null argument");
    *width = fe->width;
    *height = fe->height;
    return RTM_OK;
}

int rtm_frontend_free(rtm_frontend* fe) {
    delete fe;
    return RTM_OK;
}

}  // extern "C"
