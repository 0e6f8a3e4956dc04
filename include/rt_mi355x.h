/* rt_mi355x.h -- C ABI of the MI355X rendering back end for colobrodo/raytracer-rs.
 *
 * The library replaces the reference's per-pixel render loop and everything it calls
 * (src/main.rs:153-171 -> cast main.rs:57-132 -> hit core.rs:68-92 -> Solid::hit core.rs:180-213
 * -> ray_intersect core.rs:94-178 -> Triangle::intersect model.rs:43-74 ...).  The caller (the
 * reference's main(), through the thin FFI crate shown in INTEGRATION.md) keeps the scene parser,
 * OBJ loader, transforms, grid construction and BMP writer; it hands over a FLATTENED scene
 * (plain pointers and sizes, caller-owned) and receives the f64 framebuffer `pixels`
 * (main.rs:153, read back at main.rs:176-178).
 *
 * All entry points return 0 on success and a negative rtm_status on error; the message is
 * available from rtm_last_error().  Nothing unwinds across this boundary.  Single caller
 * thread per process (the reference renders once from main(), main.rs:158); not re-entrant: a second thread
 * entering a render entry point while another is inside gets RTM_ERR_STATE.  One frame is in flight at a time:
 * the library keeps one work space and one set of internal streams per process, so a frame queued on any stream
 * starts on the device only after the previous frame's last kernel (an event orders them; frames on different
 * caller streams do not overlap, they do not corrupt each other either).
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTM_NO_ID 0xFFFFFFFFu /* miss (obj_id) / not a triangle (tri_id) */

typedef enum rtm_status {
    RTM_OK = 0,
    RTM_ERR_INVALID = -1,   /* bad argument / inconsistent scene buffers */
    RTM_ERR_DEVICE = -2,    /* HIP runtime error (no GPU, out of memory, launch failure) */
    RTM_ERR_PARSE = -3,     /* front end: scene text error, carries "msg at line:col" (parser.rs:43-75) */
    RTM_ERR_IO = -4,        /* front end: file could not be read ("Cannot load model", parser.rs:415-420) */
    RTM_ERR_STATE = -5      /* rtm_init not called / handle already freed / another thread is inside a render call */
} rtm_status;

/* Camera fields of camera.rs:4-10 (private in the reference: the FFI crate adds accessors). */
typedef struct rtm_camera {
    double position[3];
    double forward[3];
    double up[3];
    double right[3];
    double zoom;
} rtm_camera;

/* One Solid::Model (core.rs:52) = Model (model.rs:9-14) + ModelGrid (model.rs:205-210), flattened.
 * Triangle i, corner k, component c lives in pos[k*3+c][i]: nine SoA arrays of n_tris doubles, holding
 * Model::get_triangle(i) (model.rs:108-122) evaluated once (a pure function of the vertex);
 * nrm[...] likewise holds get_triangle_normals(i) (model.rs:125-139), i.e. already transformed by
 * normal_trasform INCLUDING the /w and normalised. */
typedef struct rtm_model {
    double bbox_center[3];        /* Box3.center          (box3D.rs:5)  */
    double bbox_half[3];          /* Box3.half_extension  (box3D.rs:6)  */
    uint32_t cells_per_side;      /* ModelGrid.cells_per_side (32, core.rs:62) */
    uint32_t n_tris;
    const uint32_t* offsets;      /* ModelGrid.offset_array: INCLUSIVE prefix sums, cells_per_side^3 entries */
    const uint32_t* tri_idx;      /* ModelGrid.triangle_indices */
    uint64_t n_tri_idx;           /* == offsets[cells^3-1] */
    const double* pos[9];         /* v0x v0y v0z v1x v1y v1z v2x v2y v2z */
    const double* nrm[9];         /* n0x ... n2z */
    uint32_t mat_idx;
    uint32_t obj_idx;             /* index in Scene.objects (file order); ties go to the lower one (core.rs:79) */
} rtm_model;

typedef struct rtm_scene {
    rtm_camera camera;
    uint32_t n_objects;           /* Scene.objects.len() (core.rs:13) */
    /* Solid::Sphere (core.rs:50), SoA, any order; obj idx keeps file order */
    uint32_t n_spheres;
    const double *sph_cx, *sph_cy, *sph_cz, *sph_r;
    const uint32_t *sph_mat, *sph_obj;
    /* Solid::Plane (core.rs:51); normal as written in the scene file, NOT normalised */
    uint32_t n_planes;
    const double *pl_nx, *pl_ny, *pl_nz, *pl_d;
    const uint32_t *pl_mat, *pl_obj;
    uint32_t n_models;
    const rtm_model* models;
    /* Light (core.rs:17-21) */
    uint32_t n_lights;
    const double *li_px, *li_py, *li_pz, *li_r, *li_g, *li_b, *li_radius;
    /* Material (core.rs:30-33); type 0 = Plastic, 1 = Metal (core.rs:24-27) */
    uint32_t n_materials;
    const double *mat_r, *mat_g, *mat_b;
    const uint32_t* mat_type;
} rtm_scene;

#define RTM_FLAG_ZERO_JITTER 1u /* x_offset = y_offset = 0 (known-answer tests); reference always jitters, main.rs:162-163 */

/* Two device formulations of the same arithmetic exist; both give bit-identical images.  Default: the ray-front
 * pipeline (rt_front.h).  The flag selects the older single-kernel form, kept as the cross-check the parity tests hold against
 * the oracle and as the fallback for scenes beyond the ray-front path's packed indices (rtm_front_scene_fits).  (Flag value 2
 * selected a straight per-lane transliteration until round 3; it is retired and now rejected as an unknown flag -- the same
 * per-lane code still answers rtm_trace_rays.) */
#define RTM_FLAG_KERNEL_WAVE 4u /* the wave-scheduled single kernel (rt_wave.h) */

typedef struct rtm_params {
    uint32_t width, height;       /* ImageData.width/height (parser.rs:80-81) */
    uint32_t spp;                 /* args.sample_rate (main.rs:27), >= 1 */
    uint32_t max_bounces;         /* K in cast (main.rs:63); reference value 10; 1..64 */
    uint64_t seed;                /* Philox key; replaces the OS-seeded thread_rng (main.rs:162) */
    uint32_t row_begin;           /* rows y = row_begin + k*row_stride are rendered and packed densely:   */
    uint32_t row_stride;          /*   out row k <-> image row y (static row-interleave across GPUs); >= 1 */
    uint32_t flags;
    uint32_t reserved;
} rtm_params;

/* Work counters, exact (not sampled).  rays_* count hit() queries (core.rs:68). */
typedef struct rtm_stats {
    uint64_t rays_camera, rays_bounce, rays_shadow;
    uint64_t sphere_tests, plane_tests, bbox_tests, cells_visited, tri_tests, model_hits;
} rtm_stats;

typedef struct rtm_scene_dev rtm_scene_dev; /* opaque: a scene resident in HBM */

/* Number of rows a params block renders: ceil((height-row_begin)/row_stride). */
uint32_t rtm_local_rows(const rtm_params* p);

/* Whether a scene with `list_entries` grid-list entries and `cells` grid cells (padded to 32 per model) over all its models
 * fits the packed indices of the ray-front pipeline (1) or is rendered by the single-kernel formulation (0).  Host arithmetic
 * only, no device needed; no counterpart in the reference.  Limits: 429 496 729 entries (a list position times ten is formed
 * in 32 bits), 2^27 cells. */
int rtm_front_scene_fits(uint64_t list_entries, uint64_t cells);

/* Select the HIP device of this process (one process per GPU).  device < 0: use LOCAL_RANK or 0.
 * Calling it again with another device releases what the library held on the old one (work space, streams, events);
 * free the scenes uploaded there first.  No counterpart in the reference (it has no device). */
int rtm_init(int device);
int rtm_shutdown(void);
const char* rtm_last_error(void);
/* Library version string and the gfx target it was compiled for. */
const char* rtm_version(void);
/* Scheduling knobs of the device path (no effect on results; no counterpart in the reference).  Keys:
 *   "front_max_paths"  sample paths in flight at once over the chunks of a frame (work space: ~0.5 KB each); 0 (default) = by frame
 *                      size: chunks of 4 Mi to 64 Mi paths, two rounds of "front_streams" chunks where the frame allows,
 *   "front_cull"       1 (default): shadow rays whose light is behind the surface (main.rs:96 fails) are not traced,
 *   "front_refill"     idle lanes per wave at which the trace kernel fetches new rays (default 24),
 *   "front_streams"    chunks of a frame in flight at once on internal streams (1..4, default 2; 0 = by frame size: 3 from
 *                      12 Mi sample paths per frame, else 2),
 *   "front_claim"      ray items a trace wave claims from a sub-queue at a time (>= 16, default 128),
 *   "front_steal"      sub-queues of other waves a trace wave may try when its own are empty (default 8),
 *   "front_timed_events"  1: event pairs around the trace launches of ordinary frames (rtm_timed_events_collect),
 *   "front_coop"       1 (default): when at most four rays of a wave are testing triangles, its other lanes test
 *                      their cells' entries for them,
 *   "front_rehit"      1 (default): in scenes with at most four spheres and planes the analytic hit of a path query is
 *                      computed again when its answer is taken instead of travelling with the path (12 bytes per path and round),
 *                      and for up to 16 bounces the attenuation is multiplied up again from the vertices' metal flags (8 bytes),
 *   "front_inlane"     a path whose bounce and shadow queries all missed the models' boxes advances to its next vertex inside
 *                      the shade kernel's lane when at least this many lanes of the wave do (1..64, default 24; 0 = never: every
 *                      vertex waits a round; -1 = default).  Unless set by hand it is off in scenes with more than 16 spheres and planes,
 *   "front_inlane_lights"  1 (default): in scenes where "front_inlane" is off by itself (many analytic primitives) and with more than
 *                      two lights, the further light batches of a vertex are still taken in the lane (every path of the wave is in
 *                      that state at once, so nothing is lost to divergence),
 *   "front_sph_screen" 1 (default): in scenes with more than four spheres a screen in packed f32 (two spheres per instruction) runs in
 *                      front of the f64 sphere test of core.rs:183-195, which then sees only the few spheres a ray can meet; the
 *                      screen's margin is a proven bound on what f32 loses, results are bit-identical,
 *   "front_pool"       1: the trace kernel parks a pool of 96 rays per wave in LDS and runs every walk / test step on up to 64
 *                      rays that are ready for it (rt_pool.h); 0 (default): one ray per lane, in registers.  Measured slower
 *                      (DESIGN.md 4.2), kept as a cross-check; "pool_refill", "pool_walk_more", "pool_test_frac" tune it.
 * Returns RTM_ERR_INVALID for an unknown key or a value out of range.  The same knobs are read from the
 * environment (RTM_FRONT_MAX_PATHS, ...) by rtm_init. */
int rtm_set_tuning(const char* key, long long value);

/* Replaces main.rs:153-171 verbatim: fills out_rgb[(k*width + x)*3 + c] (f64, linear, unclamped; for
 * row_begin=0,row_stride=1 that is pixels[x + width*y], main.rs:177).  Optional outputs (may be NULL):
 * out_obj_id / out_tri_id = Scene.objects index and triangle index (core.rs:141,169) of the FIRST sample's
 * camera-ray hit per pixel; out_stats = exact work counters.  Host pointers; uploads the scene, renders,
 * copies back, frees. */
int rtm_render(const rtm_scene* scene, const rtm_params* params, double* out_rgb, uint32_t* out_obj_id,
               uint32_t* out_tri_id, rtm_stats* out_stats);

/* Same path with the scene kept resident in HBM between frames (no reference counterpart: the reference
 * holds its Scene in host memory for the life of main()). */
int rtm_scene_upload(const rtm_scene* scene, rtm_scene_dev** out);
int rtm_scene_free(rtm_scene_dev* scene);
/* d_rgb / d_obj_id / d_tri_id are DEVICE pointers (local_rows*width*3 doubles / local_rows*width u32);
 * stream is a hipStream_t (NULL = default stream).  Asynchronous unless out_stats != NULL. */
int rtm_render_device(const rtm_scene_dev* scene, const rtm_params* params, double* d_rgb, uint32_t* d_obj_id,
                      uint32_t* d_tri_id, rtm_stats* out_stats, void* stream);
/* rtm_render_device for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  It has no counterpart in the reference.  Same kernels and same bits as rtm_render_device's
 * default (ray-front) path, but their work space -- path state, query results, ray items, sub-queue counters -- is a
 * static device array of rtm_noalloc_scratch_bytes() = 256 MiB that every GPU the library runs on holds until the
 * process ends, instead of hipMalloc'ed memory; the call makes no hipMalloc/hipFree and creates no stream or event:
 * it only launches kernels on `stream`.  The frame is rendered in groups of 8-row bands that fit the array; a frame
 * whose single band does not fit (about width x spp > 32 Ki sample paths per row: e.g. 1200 wide up to 27 spp, 3840 wide
 * up to 8 spp) is rejected with RTM_ERR_INVALID and a message that states the need.  No out_stats, no
 * RTM_FLAG_KERNEL_* (RTM_ERR_INVALID); scenes beyond the ray-front pipeline's limits are rejected, not rerouted.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (different streams, or a captured
 * graph replayed while another call runs) RACE and corrupt each other's frames -- serialise them; the array keeps the
 * last call's contents and the kernels rewrite every word before reading it, so nothing needs clearing. */
int rtm_render_device_noalloc(const rtm_scene_dev* scene, const rtm_params* params, double* d_rgb, uint32_t* d_obj_id,
                              uint32_t* d_tri_id, void* stream);
uint64_t rtm_noalloc_scratch_bytes(void);

/* Average device time (ms, HIP events on `stream`) of `iters` back-to-back rtm_render_device launches. */
int rtm_time_render_device(const rtm_scene_dev* scene, const rtm_params* params, double* d_rgb, int iters,
                           void* stream, double* out_ms_avg);

/* Where a frame's device time goes, for bench.py's roofline line (no counterpart in the reference).  The frame is
 * rendered twice on `stream`: once with a HIP event pair around every kernel launch, once with the counting
 * instantiation under the production settings (rtm_render_device's out_stats turns shadow-ray culling off so that
 * its counters equal the reference's work; these are the counters of the work actually done).  Synchronous. */
typedef struct rtm_profile {
    double total_ms;                        /* sum of the kernels below */
    double shade_ms, trace_ms, resolve_ms;  /* rt_front.h kernels; a single-kernel formulation reports trace_ms only */
    uint32_t shade_launches, trace_launches;
    uint64_t traced_rays;                   /* rays that entered a model's box and went through the trace kernel */
    uint64_t cells_visited, tri_tests;      /* work of the trace kernel */
    uint64_t rays_total;                    /* camera + bounce + shadow rays issued (culled shadow rays not counted) */
    uint64_t sphere_tests, plane_tests, bbox_tests;  /* work of the shade kernel's queries (culled shadow rays make none) */
} rtm_profile;
int rtm_profile_render_device(const rtm_scene_dev* scene, const rtm_params* params, double* d_rgb, void* stream,
                              rtm_profile* out);

/* The dominant kernel timed in the regime of ordinary frames (no counterpart in the reference): after
 * rtm_set_tuning("front_timed_events", 1) every rtm_render_device call records a HIP event pair around each trace-kernel
 * launch, on the stream the kernel is launched on; this call waits for the device, returns the summed duration and the
 * number of launches since the last call, and forgets them. */
int rtm_timed_events_collect(double* out_trace_ms, uint32_t* out_trace_launches);

/* Closest-hit of caller-supplied rays (hit, core.rs:68-92), for differential tests: rays6 = origin xyz,
 * direction xyz (host, n*6); outputs host arrays (obj n, tri n, t n, normal n*3). */
int rtm_trace_rays(const rtm_scene_dev* scene, uint64_t n, const double* rays6, uint32_t* out_obj_id,
                   uint32_t* out_tri_id, double* out_t, double* out_normal3);

/* Framebuffer -> 8-bit RGB with the reference's rule (main.rs:33-48,176-185): optional gamma
 * c>0 ? c^(1/2.2) : 0, then (c*255.0) saturating-truncating cast.  d_rgb, d_rgb8: device pointers.
 * Without gamma the bytes are exactly the reference's.  With gamma the power is the DEVICE's pow(), the reference's is
 * the host libm's (f64::powf): a value that lands within an ulp of a multiple of 1/255 can come out one code value apart, so
 * this entry point's gamma path is at most +-1 per byte.  rtm_render_rgb8 and rtm_cli_main --gamma-correction do not use it:
 * they apply the gamma on the host and give the reference's bytes. */
int rtm_tonemap_device(const double* d_rgb, uint64_t n_pixels, int gamma, uint8_t* d_rgb8, void* stream);

/* main.rs:153-185 in one call for callers that only want the 8-bit image: renders like
 * rtm_render_device, then (gamma == 0) converts on the device with rtm_tonemap_device's rule and copies local_rows*width*3
 * bytes to the HOST buffer out_rgb8 -- an eighth of the f64 framebuffer over PCIe -- or (gamma != 0) copies the f64
 * framebuffer to the host and applies main.rs:42-48 there with the host libm's pow, as the reference does (main.rs:179-183):
 * in both cases the bytes are the reference's.  out_seconds (may be NULL): wall time of render + conversion + copy, the
 * bracket of main.rs:156-187.  Synchronous. */
int rtm_render_rgb8(const rtm_scene_dev* scene, const rtm_params* params, int gamma, uint8_t* out_rgb8,
                    double* out_seconds);

/* De-interleave gathered per-rank row blocks into the row-major framebuffer (multi-GPU assemble):
 * d_blocks holds n_ranks blocks, block g = rows y = g + k*n_ranks packed densely; all device pointers. */
int rtm_deinterleave_device(const double* d_blocks, uint32_t width, uint32_t height, uint32_t n_ranks,
                            double* d_rgb, void* stream);

/* Test hook (no reference counterpart): evaluates on the GPU, per element, the f64 primitives the bit-exact
 * path relies on -- out[0][i] = a/b, out[1][i] = sqrt(|a|), out[2][i] = fmin(a,b), out[3][i] = fmax(a,b),
 * out[4][i] = a*b + a (must round twice: no FMA contraction), out[5][i] = (double)sat_u32 cell cast of a.
 * a, b, out6 (6*n doubles) are HOST pointers. */
int rtm_f64_ops_selftest(uint64_t n, const double* a, const double* b, double* out6);

/* ---- Host-side front end (C++ stand-in for the parts that stay in Rust: parser.rs, obj-rs load_obj,
 * mat4.rs, Model::new model.rs:91-103, create_grid model.rs:147-201, Camera::look_at camera.rs:30-43).
 * Produces the flattened rtm_scene the FFI crate would build.  Pure host code, no GPU needed. */
typedef struct rtm_frontend rtm_frontend;
/* scene_text != NULL: parse it; else read scene_path.  Model paths resolve against base_dir (the reference
 * resolves against the cwd, core.rs:57).  flags bit 0: synthesise geometric normals for OBJ faces without
 * `vn` (NOT reference behaviour; teapot.obj only). */
int rtm_frontend_load(const char* scene_path, const char* scene_text, const char* base_dir, uint32_t flags,
                      rtm_frontend** out);
/* Append a Solid::Model built from raw f32 triangles (synthetic soup): pos9/nrm9 = n_tris*9 floats
 * (corner-major xyz), transform16 row-major or NULL, grid = create_grid(32). */
int rtm_frontend_add_triangles(rtm_frontend* fe, uint64_t n_tris, const float* pos9, const float* nrm9,
                               const double* color3, uint32_t mat_type, const double* transform16);
const rtm_scene* rtm_frontend_scene(const rtm_frontend* fe);
int rtm_frontend_size(const rtm_frontend* fe, uint32_t* width, uint32_t* height);
int rtm_frontend_free(rtm_frontend* fe);

/* ---- Model::create_grid (model.rs:147-201) on the device: bins n_tris triangles (pos[k*3+c][t], as in
 * rtm_model) into cells_per_side^3 cells of the box (bbox_center, bbox_half) exactly as the reference does -- triangle box
 * grown with Box3::include (box3D.rs:31-45, which is not the true bounding box), corner cells by saturating casts
 * (model.rs:159-167), lists in ascending triangle order -- and writes offsets (ModelGrid.offset_array, INCLUSIVE prefix sums,
 * cells^3 entries) and tri_idx (ModelGrid.triangle_indices).  The number of list entries comes back in *out_n_tri_idx; when
 * tri_idx_capacity is smaller (or the tri_idx pointer NULL) the offsets are still written and the call returns
 * RTM_ERR_INVALID: allocate that many and call again.  cells_per_side 1..128 (the reference uses 32, core.rs:62).
 * rtm_grid_build: host pointers in and out.  rtm_grid_build_device: d_pos[9], d_offsets, d_tri_idx are device pointers
 * (the array d_pos itself is a host array of nine device pointers); it makes three temporary device allocations (cell ranges
 * per triangle, counts and positions per (cell, segment): freed before it returns) and synchronises `stream` twice (the list
 * total comes back to the host before the lists are written).  Needs rtm_init (RTM_ERR_STATE otherwise). */
int rtm_grid_build(const double* const pos[9], uint32_t n_tris, const double bbox_center[3], const double bbox_half[3],
                   uint32_t cells_per_side, uint32_t* out_offsets, uint32_t* out_tri_idx, uint64_t tri_idx_capacity,
                   uint64_t* out_n_tri_idx);
int rtm_grid_build_device(const double* const d_pos[9], uint32_t n_tris, const double bbox_center[3], const double bbox_half[3],
                          uint32_t cells_per_side, uint32_t* d_offsets, uint32_t* d_tri_idx, uint64_t tri_idx_capacity,
                          uint64_t* out_n_tri_idx, void* stream);

/* ---- The reference's whole program, for machines without a Rust toolchain.
 * rtm_write_bmp: what buffer.save(&args.output) writes for an ImageBuffer<Rgb<u8>> (main.rs:173,190; `image` crate
 * 0.25.1 BMP encoder: 54-byte header, 24 bpp, bottom-up, BGR, rows padded to 4 bytes); rgb8 = height*width*3, row-major,
 * top row first.  rtm_cli_main: main() (main.rs:134-192) with clap's arguments (main.rs:17-31): <scene> [-o output.bmp]
 * [-s 20] [--gamma-correction]; prints "Rendered <output> in <time>" (main.rs:188); returns the process exit status
 * (0 ok, 1 error, 2 usage).  The binary raytracer-rs_amd/lib/raytracer_mi355x only calls it. */
int rtm_write_bmp(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb8);
int rtm_cli_main(int argc, const char* const* argv);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
