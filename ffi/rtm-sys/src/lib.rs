//! `#[repr(C)]` mirrors and `extern "C"` declarations of `include/rt_mi355x.h`, one to one and in the header's order.
//! Field order, types and function signatures are checked against the header by `tests/test_ffi_crate.py`
//! (this image has no Rust toolchain: the crate is source that a maintainer of raytracer-rs adds as a path dependency).
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_longlong, c_void};

pub const RTM_NO_ID: u32 = 0xFFFF_FFFF;

pub const RTM_OK: c_int = 0;
pub const RTM_ERR_INVALID: c_int = -1;
pub const RTM_ERR_DEVICE: c_int = -2;
pub const RTM_ERR_PARSE: c_int = -3;
pub const RTM_ERR_IO: c_int = -4;
pub const RTM_ERR_STATE: c_int = -5;

pub const RTM_FLAG_ZERO_JITTER: u32 = 1;
pub const RTM_FLAG_KERNEL_WAVE: u32 = 4;

/// Camera fields (camera.rs:4-10).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtm_camera {
    pub position: [f64; 3],
    pub forward: [f64; 3],
    pub up: [f64; 3],
    pub right: [f64; 3],
    pub zoom: f64,
}

/// One `Solid::Model` (core.rs:52), flattened: Model (model.rs:9-14) + ModelGrid (model.rs:205-210).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rtm_model {
    pub bbox_center: [f64; 3],
    pub bbox_half: [f64; 3],
    pub cells_per_side: u32,
    pub n_tris: u32,
    pub offsets: *const u32,
    pub tri_idx: *const u32,
    pub n_tri_idx: u64,
    pub pos: [*const f64; 9],
    pub nrm: [*const f64; 9],
    pub mat_idx: u32,
    pub obj_idx: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rtm_scene {
    pub camera: rtm_camera,
    pub n_objects: u32,
    pub n_spheres: u32,
    pub sph_cx: *const f64,
    pub sph_cy: *const f64,
    pub sph_cz: *const f64,
    pub sph_r: *const f64,
    pub sph_mat: *const u32,
    pub sph_obj: *const u32,
    pub n_planes: u32,
    pub pl_nx: *const f64,
    pub pl_ny: *const f64,
    pub pl_nz: *const f64,
    pub pl_d: *const f64,
    pub pl_mat: *const u32,
    pub pl_obj: *const u32,
    pub n_models: u32,
    pub models: *const rtm_model,
    pub n_lights: u32,
    pub li_px: *const f64,
    pub li_py: *const f64,
    pub li_pz: *const f64,
    pub li_r: *const f64,
    pub li_g: *const f64,
    pub li_b: *const f64,
    pub li_radius: *const f64,
    pub n_materials: u32,
    pub mat_r: *const f64,
    pub mat_g: *const f64,
    pub mat_b: *const f64,
    pub mat_type: *const u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtm_params {
    pub width: u32,
    pub height: u32,
    pub spp: u32,
    pub max_bounces: u32,
    pub seed: u64,
    pub row_begin: u32,
    pub row_stride: u32,
    pub flags: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtm_stats {
    pub rays_camera: u64,
    pub rays_bounce: u64,
    pub rays_shadow: u64,
    pub sphere_tests: u64,
    pub plane_tests: u64,
    pub bbox_tests: u64,
    pub cells_visited: u64,
    pub tri_tests: u64,
    pub model_hits: u64,
}

/// Opaque: a scene resident in HBM.
#[repr(C)]
pub struct rtm_scene_dev {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtm_profile {
    pub total_ms: f64,
    pub shade_ms: f64,
    pub trace_ms: f64,
    pub resolve_ms: f64,
    pub shade_launches: u32,
    pub trace_launches: u32,
    pub traced_rays: u64,
    pub cells_visited: u64,
    pub tri_tests: u64,
    pub rays_total: u64,
    pub sphere_tests: u64,
    pub plane_tests: u64,
    pub bbox_tests: u64,
}

/// Opaque: the C++ stand-in front end (not needed from Rust, which has its own parser; declared for completeness).
#[repr(C)]
pub struct rtm_frontend {
    _private: [u8; 0],
}

extern "C" {
    pub fn rtm_local_rows(p: *const rtm_params) -> u32;
    pub fn rtm_init(device: c_int) -> c_int;
    pub fn rtm_shutdown() -> c_int;
    pub fn rtm_last_error() -> *const c_char;
    pub fn rtm_version() -> *const c_char;
    pub fn rtm_set_tuning(key: *const c_char, value: c_longlong) -> c_int;
    pub fn rtm_render(scene: *const rtm_scene, params: *const rtm_params, out_rgb: *mut f64, out_obj_id: *mut u32,
                      out_tri_id: *mut u32, out_stats: *mut rtm_stats) -> c_int;
    pub fn rtm_scene_upload(scene: *const rtm_scene, out: *mut *mut rtm_scene_dev) -> c_int;
    pub fn rtm_scene_free(scene: *mut rtm_scene_dev) -> c_int;
    pub fn rtm_render_device(scene: *const rtm_scene_dev, params: *const rtm_params, d_rgb: *mut f64, d_obj_id: *mut u32,
                             d_tri_id: *mut u32, out_stats: *mut rtm_stats, stream: *mut c_void) -> c_int;
    pub fn rtm_render_device_noalloc(scene: *const rtm_scene_dev, params: *const rtm_params, d_rgb: *mut f64,
                                     d_obj_id: *mut u32, d_tri_id: *mut u32, stream: *mut c_void) -> c_int;
    pub fn rtm_noalloc_scratch_bytes() -> u64;
    pub fn rtm_front_scene_fits(list_entries: u64, cells: u64) -> c_int;
    pub fn rtm_time_render_device(scene: *const rtm_scene_dev, params: *const rtm_params, d_rgb: *mut f64, iters: c_int,
                                  stream: *mut c_void, out_ms_avg: *mut f64) -> c_int;
    pub fn rtm_profile_render_device(scene: *const rtm_scene_dev, params: *const rtm_params, d_rgb: *mut f64,
                                     stream: *mut c_void, out: *mut rtm_profile) -> c_int;
    pub fn rtm_timed_events_collect(out_trace_ms: *mut f64, out_trace_launches: *mut u32) -> c_int;
    pub fn rtm_trace_rays(scene: *const rtm_scene_dev, n: u64, rays6: *const f64, out_obj_id: *mut u32, out_tri_id: *mut u32,
                          out_t: *mut f64, out_normal3: *mut f64) -> c_int;
    pub fn rtm_tonemap_device(d_rgb: *const f64, n_pixels: u64, gamma: c_int, d_rgb8: *mut u8, stream: *mut c_void) -> c_int;
    pub fn rtm_render_rgb8(scene: *const rtm_scene_dev, params: *const rtm_params, gamma: c_int, out_rgb8: *mut u8,
                           out_seconds: *mut f64) -> c_int;
    pub fn rtm_deinterleave_device(d_blocks: *const f64, width: u32, height: u32, n_ranks: u32, d_rgb: *mut f64,
                                   stream: *mut c_void) -> c_int;
    pub fn rtm_f64_ops_selftest(n: u64, a: *const f64, b: *const f64, out6: *mut f64) -> c_int;
    pub fn rtm_frontend_load(scene_path: *const c_char, scene_text: *const c_char, base_dir: *const c_char, flags: u32,
                             out: *mut *mut rtm_frontend) -> c_int;
    pub fn rtm_frontend_add_triangles(fe: *mut rtm_frontend, n_tris: u64, pos9: *const f32, nrm9: *const f32,
                                      color3: *const f64, mat_type: u32, transform16: *const f64) -> c_int;
    pub fn rtm_frontend_scene(fe: *const rtm_frontend) -> *const rtm_scene;
    pub fn rtm_frontend_size(fe: *const rtm_frontend, width: *mut u32, height: *mut u32) -> c_int;
    pub fn rtm_frontend_free(fe: *mut rtm_frontend) -> c_int;
    pub fn rtm_grid_build(pos: *const *const f64, n_tris: u32, bbox_center: *const f64, bbox_half: *const f64,
                          cells_per_side: u32, out_offsets: *mut u32, out_tri_idx: *mut u32, tri_idx_capacity: u64,
                          out_n_tri_idx: *mut u64) -> c_int;
    pub fn rtm_grid_build_device(d_pos: *const *const f64, n_tris: u32, bbox_center: *const f64, bbox_half: *const f64,
                                 cells_per_side: u32, d_offsets: *mut u32, d_tri_idx: *mut u32, tri_idx_capacity: u64,
                                 out_n_tri_idx: *mut u64, stream: *mut c_void) -> c_int;
    pub fn rtm_write_bmp(path: *const c_char, width: u32, height: u32, rgb8: *const u8) -> c_int;
    pub fn rtm_cli_main(argc: c_int, argv: *const *const c_char) -> c_int;
}
