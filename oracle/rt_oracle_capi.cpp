// rt_oracle_capi.cpp -- C entry points of the CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
// Loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
// See rt_oracle.hpp for the parity status ("parity unpinned" by the reference) and the citations.
#include <atomic>
#include <mutex>
#include <thread>

#include "rt_oracle.hpp"

using namespace orc;

namespace {
struct Handle {
    ImageData img;
};
void set_err(char* err, int errlen, const std::string& m) {
    if (err && errlen > 0) {
        std::snprintf(err, (size_t)errlen, "%s", m.c_str());
    }
}
}  // namespace

extern "C" {

struct orc_params {
    uint32_t width, height;  // 0 = the scene file's `size`
    uint32_t spp;
    uint32_t max_bounces;    // reference: 10 (main.rs:63)
    uint64_t seed;
    uint32_t row_begin, row_stride;  // rows y = row_begin + k*row_stride are rendered, packed densely
    uint32_t flags;                  // 1 = zero jitter
    uint32_t faithful;               // 1 = per-access Mat4::apply (model.rs:108-122), 0 = pre-transformed
    uint32_t n_threads;              // 0 = hardware_concurrency
    uint32_t reserved;
};

// text != NULL: parse that string; else read scene_path.  base_dir: where model paths resolve.
void* orc_load(const char* scene_path, const char* text, const char* base_dir, int allow_missing_normals, char* err,
               int errlen) {
    std::string content;
    if (text) {
        content = text;
    } else {
        std::ifstream in(scene_path);
        if (!in) { set_err(err, errlen, std::string("cannot open ") + scene_path); return nullptr; }
        std::stringstream ss;
        ss << in.rdbuf();
        content = ss.str();
    }
    auto* h = new Handle();
    SceneParser p(content, base_dir ? base_dir : "", allow_missing_normals != 0);
    ParseError pe;
    if (!p.parse_scene(h->img, pe)) {
        set_err(err, errlen, pe.message + " at " + std::to_string(pe.line) + ":" + std::to_string(pe.column));
        delete h;
        return nullptr;
    }
    return h;
}

void orc_free(void* hp) { delete (Handle*)hp; }

void orc_info(void* hp, uint32_t* w, uint32_t* h, uint32_t* n_objects, uint32_t* n_lights, uint32_t* n_models) {
    const ImageData& img = ((Handle*)hp)->img;
    *w = img.width; *h = img.height;
    *n_objects = (uint32_t)img.scene.objects.size();
    *n_lights = (uint32_t)img.scene.lights.size();
    uint32_t nm = 0;
    for (auto& o : img.scene.objects) nm += o.kind == MODEL;
    *n_models = nm;
}

// camera as 13 doubles: position, forward, up, right, zoom
void orc_camera(void* hp, double* out13) {
    const Camera& c = ((Handle*)hp)->img.camera;
    const Vec3 v[4] = {c.position, c.forward, c.up, c.right};
    for (int i = 0; i < 4; ++i) { out13[i * 3] = v[i].x; out13[i * 3 + 1] = v[i].y; out13[i * 3 + 2] = v[i].z; }
    out13[12] = c.zoom;
}

// object oi: kind, 4 doubles (a.xyz, b), material colour + type
void orc_object(void* hp, uint32_t oi, uint32_t* kind, double* abcd, double* color, uint32_t* mat_type) {
    const SceneObject& o = ((Handle*)hp)->img.scene.objects.at(oi);
    *kind = o.kind;
    abcd[0] = o.a.x; abcd[1] = o.a.y; abcd[2] = o.a.z; abcd[3] = o.b;
    color[0] = o.material.color.x; color[1] = o.material.color.y; color[2] = o.material.color.z;
    *mat_type = o.material.type;
}
void orc_light(void* hp, uint32_t li, double* out7) {
    const Light& l = ((Handle*)hp)->img.scene.lights.at(li);
    out7[0] = l.position.x; out7[1] = l.position.y; out7[2] = l.position.z;
    out7[3] = l.color.x; out7[4] = l.color.y; out7[5] = l.color.z; out7[6] = l.radius;
}

// model-bearing object oi: sizes, then arrays (faithful get_triangle / get_triangle_normals outputs)
int orc_model_info(void* hp, uint32_t oi, uint64_t* n_tris, uint64_t* n_idx, uint32_t* cps, double* bbox_center_half6) {
    const SceneObject& o = ((Handle*)hp)->img.scene.objects.at(oi);
    if (o.kind != MODEL) return -1;
    *n_tris = o.model->n_tris();
    *n_idx = o.grid->triangle_indices.size();
    *cps = o.grid->n;
    const Box3& b = o.model->bounding_box;
    const double v[6] = {b.center.x, b.center.y, b.center.z, b.half.x, b.half.y, b.half.z};
    std::memcpy(bbox_center_half6, v, sizeof v);
    return 0;
}
int orc_model_export(void* hp, uint32_t oi, double* pos9, double* nrm9, uint64_t* offsets, uint64_t* tri_idx) {
    const SceneObject& o = ((Handle*)hp)->img.scene.objects.at(oi);
    if (o.kind != MODEL) return -1;
    const size_t nt = o.model->n_tris();
    for (size_t i = 0; i < nt; ++i) {
        const Triangle t = o.model->get_triangle(i);
        Vec3 n[3];
        o.model->get_triangle_normals(i, n);
        const Vec3 p[3] = {t.v0, t.v1, t.v2};
        for (int k = 0; k < 3; ++k) {
            if (pos9) { pos9[i * 9 + k * 3] = p[k].x; pos9[i * 9 + k * 3 + 1] = p[k].y; pos9[i * 9 + k * 3 + 2] = p[k].z; }
            if (nrm9) { nrm9[i * 9 + k * 3] = n[k].x; nrm9[i * 9 + k * 3 + 1] = n[k].y; nrm9[i * 9 + k * 3 + 2] = n[k].z; }
        }
    }
    if (offsets) std::memcpy(offsets, o.grid->offset_array.data(), o.grid->offset_array.size() * 8);
    if (tri_idx) std::memcpy(tri_idx, o.grid->triangle_indices.data(), o.grid->triangle_indices.size() * 8);
    return 0;
}

// Append a model made of raw f32 triangles (the synthetic soup of BASELINE config 5): one vertex per
// corner, no de-duplication, u32 indices; transform16 row-major (NULL = identity); grid = create_grid(32).
int orc_add_triangles(void* hp, uint64_t n_tris, const float* pos9, const float* nrm9, const double* color3,
                      uint32_t mat_type, const double* transform16, char* err, int errlen) {
    Obj obj;
    obj.vertices.resize(n_tris * 3);
    obj.indices.resize(n_tris * 3);
    for (uint64_t i = 0; i < n_tris * 3; ++i) {
        for (int j = 0; j < 3; ++j) { obj.vertices[i].position[j] = pos9[i * 3 + j]; obj.vertices[i].normal[j] = nrm9[i * 3 + j]; }
        obj.indices[i] = (uint32_t)i;
    }
    Mat4 t = Mat4::identity();
    if (transform16) std::memcpy(t.m, transform16, sizeof t.m);
    auto model = std::make_shared<Model>();
    std::string e;
    if (!Model::make(std::move(obj), t, *model, e)) { set_err(err, errlen, e); return -1; }
    auto grid = std::make_shared<ModelGrid>(model->create_grid(32));
    model->precompute();
    SceneObject o;
    o.kind = MODEL;
    o.model = model;
    o.grid = grid;
    o.material = Material{v3(color3[0], color3[1], color3[2]), mat_type ? Metal : Plastic};
    ((Handle*)hp)->img.scene.objects.push_back(std::move(o));
    return 0;
}

// counters9: rays_camera, rays_bounce, rays_shadow, sphere_tests, plane_tests, bbox_tests, cells_visited,
//            tri_tests, model_hits
int orc_render(void* hp, const orc_params* p, double* rgb, uint32_t* obj_id, uint32_t* tri_id, uint64_t* counters9) {
    Handle* h = (Handle*)hp;
    ImageData img = h->img;  // shallow copy (models are shared_ptr)
    if (p->width) img.width = p->width;
    if (p->height) img.height = p->height;
    const uint32_t W = img.width, H = img.height;
    const uint32_t stride = p->row_stride ? p->row_stride : 1;
    if (p->row_begin >= H && H > 0) return -2;
    const uint32_t n_rows = H == 0 ? 0 : (H - p->row_begin + stride - 1) / stride;
    const uint64_t n_px = (uint64_t)n_rows * W;
    unsigned nt = p->n_threads ? p->n_threads : std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    std::atomic<uint64_t> next{0};
    const uint64_t chunk = 256;  // dynamic chunking, stand-in for rayon work stealing (main.rs:158)
    std::vector<Counters> per_thread(nt);
    const bool want_cnt = counters9 != nullptr;
    auto worker = [&](unsigned tid) {
        Counters* cnt = want_cnt ? &per_thread[tid] : nullptr;
        for (;;) {
            const uint64_t b = next.fetch_add(chunk);
            if (b >= n_px) break;
            const uint64_t e = std::min(n_px, b + chunk);
            for (uint64_t li = b; li < e; ++li) {
                const uint32_t lrow = (uint32_t)(li / W), x = (uint32_t)(li % W);
                const uint32_t y = p->row_begin + lrow * stride;
                const uint32_t idx = x + W * y;
                uint32_t o0 = NO_ID, t0 = NO_ID;
                const Vec3 c = render_pixel(img, idx, p->spp, p->seed, p->flags, (int)p->max_bounces, p->faithful != 0,
                                            cnt, &o0, &t0);
                rgb[li * 3] = c.x; rgb[li * 3 + 1] = c.y; rgb[li * 3 + 2] = c.z;
                if (obj_id) obj_id[li] = o0;
                if (tri_id) tri_id[li] = t0;
            }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto& t : th) t.join();
    if (want_cnt) {
        Counters tot;
        for (auto& c : per_thread) tot.add(c);
        const uint64_t v[9] = {tot.rays_camera, tot.rays_bounce, tot.rays_shadow, tot.sphere_tests, tot.plane_tests,
                               tot.bbox_tests,  tot.cells_visited, tot.tri_tests, tot.model_hits};
        std::memcpy(counters9, v, sizeof v);
    }
    return 0;
}

// closest-hit of arbitrary rays (differential fuzzing): rays6 = origin xyz, direction xyz
int orc_trace(void* hp, uint64_t n, const double* rays6, uint32_t* obj_id, uint32_t* tri_id, double* t_out,
              double* normal3, int faithful) {
    const Scene& sc = ((Handle*)hp)->img.scene;
    for (uint64_t i = 0; i < n; ++i) {
        const Ray r{v3(rays6[i * 6], rays6[i * 6 + 1], rays6[i * 6 + 2]), v3(rays6[i * 6 + 3], rays6[i * 6 + 4], rays6[i * 6 + 5])};
        RaycastResult res;
        const bool h = hit(sc, r, res, faithful != 0, nullptr);
        obj_id[i] = h ? res.obj : NO_ID;
        tri_id[i] = h ? res.tri : NO_ID;
        t_out[i] = h ? res.t : std::numeric_limits<double>::infinity();
        normal3[i * 3] = h ? res.normal.x : 0.0;
        normal3[i * 3 + 1] = h ? res.normal.y : 0.0;
        normal3[i * 3 + 2] = h ? res.normal.z : 0.0;
    }
    return 0;
}

// one ray through hit (core.rs:68-92) with the work counters of that query alone (known-answer tests of the DDA)
int orc_trace_counters(void* hp, const double* ray6, uint64_t* counters9) {
    const Scene& sc = ((Handle*)hp)->img.scene;
    const Ray r{v3(ray6[0], ray6[1], ray6[2]), v3(ray6[3], ray6[4], ray6[5])};
    RaycastResult res;
    Counters c;
    const bool h = hit(sc, r, res, false, &c);
    const uint64_t v[9] = {c.rays_camera, c.rays_bounce, c.rays_shadow, c.sphere_tests, c.plane_tests, c.bbox_tests, c.cells_visited,
                           c.tri_tests, c.model_hits};
    for (int i = 0; i < 9; ++i) counters9[i] = v[i];
    return h ? 1 : 0;
}

// camera rays for given (u,v) pairs (camera.rs:46-52)
void orc_shoot(void* hp, uint64_t n, const double* uv2, double* rays6) {
    const Camera& c = ((Handle*)hp)->img.camera;
    for (uint64_t i = 0; i < n; ++i) {
        const Ray r = c.shoot_to(uv2[i * 2], uv2[i * 2 + 1]);
        rays6[i * 6] = r.origin.x; rays6[i * 6 + 1] = r.origin.y; rays6[i * 6 + 2] = r.origin.z;
        rays6[i * 6 + 3] = r.direction.x; rays6[i * 6 + 4] = r.direction.y; rays6[i * 6 + 5] = r.direction.z;
    }
}

// Philox + float mapping, exposed for known-answer tests
void orc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t* out4) {
    Philox::gen(seed, c0, c1, c2, c3, out4);
}
void orc_sample_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, double* jitter2, double* rvec3) {
    SampleRng r{seed, pixel, sample};
    r.jitter(jitter2[0], jitter2[1]);
    const Vec3 v = r.random_vec(bounce);
    rvec3[0] = v.x; rvec3[1] = v.y; rvec3[2] = v.z;
}

// u8 conversion of main.rs:33-48,176-185 (quirk Q17)
void orc_to_rgb8(uint64_t n_px, const double* rgb, int gamma, uint8_t* out) {
    for (uint64_t i = 0; i < n_px * 3; ++i) {
        double v = rgb[i];
        if (gamma) v = v > 0.0 ? std::pow(v, 1.0 / 2.2) : 0.0;
        out[i] = as_u8(v * 255.0);
    }
}

}  // extern "C"
