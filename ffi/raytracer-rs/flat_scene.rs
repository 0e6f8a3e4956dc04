//! src/flat_scene.rs -- hands the parsed scene to the MI355X back end (rtm-sys) and gets `pixels` back: the replacement
//! for the rayon loop of main.rs:153-171.  Add `mod flat_scene;` to main.rs, `rtm-sys = { path = "<repo>/ffi/rtm-sys" }`
//! to Cargo.toml, and the accessors of camera_accessors.rs / model_accessors.rs to the two modules they name.
//!
//! Everything is COPIED into C-layout buffers owned by `FlatScene`: the reference's types are not `#[repr(C)]`.
use crate::raytracing::camera::Camera;
use crate::raytracing::core::{Material, MaterialType, Scene, Solid};
use crate::raytracing::parser::ImageData;
use crate::raytracing::Vec3;
use rtm_sys::*;
use std::ffi::CStr;

fn v3(v: Vec3) -> [f64; 3] {
    [v.x, v.y, v.z]
}

/// One `Solid::Model` as nine position arrays, nine normal arrays and the grid, all owned.
struct FlatModel {
    bbox_center: [f64; 3],
    bbox_half: [f64; 3],
    cells_per_side: u32,
    offsets: Vec<u32>,
    tri_idx: Vec<u32>,
    pos: [Vec<f64>; 9],
    nrm: [Vec<f64>; 9],
    mat_idx: u32,
    obj_idx: u32,
}

/// The flattened scene: structure-of-arrays buffers in the layout of `rtm_scene` (include/rt_mi355x.h).
pub struct FlatScene {
    camera: rtm_camera,
    n_objects: u32,
    sph: [Vec<f64>; 4], // cx cy cz r
    sph_mat: Vec<u32>,
    sph_obj: Vec<u32>,
    pl: [Vec<f64>; 4], // nx ny nz d
    pl_mat: Vec<u32>,
    pl_obj: Vec<u32>,
    models: Vec<FlatModel>,
    c_models: Vec<rtm_model>, // pointers into `models`; rebuilt by as_rtm()
    li: [Vec<f64>; 7],        // px py pz r g b radius
    mat: [Vec<f64>; 3],
    mat_type: Vec<u32>,
}

impl FlatScene {
    /// Walks `scene.objects` in file order; the position in that Vec is the object index the back end uses to break
    /// ties exactly as `hit` does (core.rs:79: strictly smaller t, so the earlier object keeps an equal one).
    pub fn from(camera: &Camera, scene: &Scene) -> FlatScene {
        let mut f = FlatScene {
            camera: rtm_camera {
                position: v3(camera.position()),
                forward: v3(camera.forward()),
                up: v3(camera.up()),
                right: v3(camera.right()),
                zoom: camera.zoom(),
            },
            n_objects: scene.objects.len() as u32,
            sph: Default::default(),
            sph_mat: Vec::new(),
            sph_obj: Vec::new(),
            pl: Default::default(),
            pl_mat: Vec::new(),
            pl_obj: Vec::new(),
            models: Vec::new(),
            c_models: Vec::new(),
            li: Default::default(),
            mat: Default::default(),
            mat_type: Vec::new(),
        };
        for (obj_idx, object) in scene.objects.iter().enumerate() {
            let mat_idx = f.push_material(&object.material);
            match &object.solid {
                Solid::Sphere { center, radius } => {
                    // core.rs:50
                    f.sph[0].push(center.x);
                    f.sph[1].push(center.y);
                    f.sph[2].push(center.z);
                    f.sph[3].push(*radius);
                    f.sph_mat.push(mat_idx);
                    f.sph_obj.push(obj_idx as u32);
                }
                Solid::Plane { normal, distance } => {
                    // core.rs:51; the normal goes over as written in the scene file, NOT normalised (core.rs:209)
                    f.pl[0].push(normal.x);
                    f.pl[1].push(normal.y);
                    f.pl[2].push(normal.z);
                    f.pl[3].push(*distance);
                    f.pl_mat.push(mat_idx);
                    f.pl_obj.push(obj_idx as u32);
                }
                Solid::Model { model, grid } => {
                    // core.rs:52.  get_triangle / get_triangle_normals (model.rs:108-139) are pure functions of the
                    // triangle index: evaluated once here instead of on every intersection test, same f64 bits.
                    let n = model.n_triangles();
                    let mut pos: [Vec<f64>; 9] = Default::default();
                    let mut nrm: [Vec<f64>; 9] = Default::default();
                    for a in 0..9 {
                        pos[a].reserve(n);
                        nrm[a].reserve(n);
                    }
                    for t in 0..n {
                        let (v0, v1, v2) = model.get_triangle(t).vertices();
                        let (n0, n1, n2) = model.get_triangle_normals(t);
                        for (k, (v, nn)) in [(v0, n0), (v1, n1), (v2, n2)].iter().enumerate() {
                            pos[k * 3].push(v.x);
                            pos[k * 3 + 1].push(v.y);
                            pos[k * 3 + 2].push(v.z);
                            nrm[k * 3].push(nn.x);
                            nrm[k * 3 + 1].push(nn.y);
                            nrm[k * 3 + 2].push(nn.z);
                        }
                    }
                    f.models.push(FlatModel {
                        // Box3 is stored as centre + half extension (box3D.rs:5-6); min() / max() are recomputed from
                        // them on every use (box3D.rs:55-61), and so does the back end
                        bbox_center: v3(model.bounding_box.center),
                        bbox_half: v3(model.bounding_box.half_extension),
                        cells_per_side: grid.cells_per_side(),
                        offsets: grid.offset_array().iter().map(|&o| o as u32).collect(), // inclusive prefix sums (model.rs:185-193)
                        tri_idx: grid.triangle_indices().iter().map(|&i| i as u32).collect(),
                        pos,
                        nrm,
                        mat_idx,
                        obj_idx: obj_idx as u32,
                    });
                }
            }
        }
        for light in &scene.lights {
            // core.rs:17-21
            f.li[0].push(light.position.x);
            f.li[1].push(light.position.y);
            f.li[2].push(light.position.z);
            f.li[3].push(light.color.x);
            f.li[4].push(light.color.y);
            f.li[5].push(light.color.z);
            f.li[6].push(light.radius);
        }
        f
    }

    /// One material entry per object (core.rs:30-33); type 0 = Plastic, 1 = Metal (core.rs:24-27).
    fn push_material(&mut self, m: &Material) -> u32 {
        self.mat[0].push(m.color.x);
        self.mat[1].push(m.color.y);
        self.mat[2].push(m.color.z);
        self.mat_type.push(match m.type_ {
            MaterialType::Plastic => 0,
            MaterialType::Metal => 1,
        });
        (self.mat_type.len() - 1) as u32
    }

    /// The C view; valid while `self` is neither moved nor changed.
    pub fn as_rtm(&mut self) -> rtm_scene {
        self.c_models = self
            .models
            .iter()
            .map(|m| {
                let mut pos = [std::ptr::null::<f64>(); 9];
                let mut nrm = [std::ptr::null::<f64>(); 9];
                for a in 0..9 {
                    pos[a] = m.pos[a].as_ptr();
                    nrm[a] = m.nrm[a].as_ptr();
                }
                rtm_model {
                    bbox_center: m.bbox_center,
                    bbox_half: m.bbox_half,
                    cells_per_side: m.cells_per_side,
                    n_tris: m.pos[0].len() as u32,
                    offsets: m.offsets.as_ptr(),
                    tri_idx: m.tri_idx.as_ptr(),
                    n_tri_idx: m.tri_idx.len() as u64,
                    pos,
                    nrm,
                    mat_idx: m.mat_idx,
                    obj_idx: m.obj_idx,
                }
            })
            .collect();
        rtm_scene {
            camera: self.camera,
            n_objects: self.n_objects,
            n_spheres: self.sph_mat.len() as u32,
            sph_cx: self.sph[0].as_ptr(),
            sph_cy: self.sph[1].as_ptr(),
            sph_cz: self.sph[2].as_ptr(),
            sph_r: self.sph[3].as_ptr(),
            sph_mat: self.sph_mat.as_ptr(),
            sph_obj: self.sph_obj.as_ptr(),
            n_planes: self.pl_mat.len() as u32,
            pl_nx: self.pl[0].as_ptr(),
            pl_ny: self.pl[1].as_ptr(),
            pl_nz: self.pl[2].as_ptr(),
            pl_d: self.pl[3].as_ptr(),
            pl_mat: self.pl_mat.as_ptr(),
            pl_obj: self.pl_obj.as_ptr(),
            n_models: self.c_models.len() as u32,
            models: self.c_models.as_ptr(),
            n_lights: self.li[0].len() as u32,
            li_px: self.li[0].as_ptr(),
            li_py: self.li[1].as_ptr(),
            li_pz: self.li[2].as_ptr(),
            li_r: self.li[3].as_ptr(),
            li_g: self.li[4].as_ptr(),
            li_b: self.li[5].as_ptr(),
            li_radius: self.li[6].as_ptr(),
            n_materials: self.mat_type.len() as u32,
            mat_r: self.mat[0].as_ptr(),
            mat_g: self.mat[1].as_ptr(),
            mat_b: self.mat[2].as_ptr(),
            mat_type: self.mat_type.as_ptr(),
        }
    }
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(rtm_last_error()).to_string_lossy().into_owned() }
}

/// What main.rs:153-171 computes: `pixels[x + width * y]`, f64, linear, unclamped.  `Vec3` is three f64 in declaration
/// order (vec3.rs:5-9); add `#[repr(C)]` to it so that a `Vec<Vec3>` is the `width * height * 3` doubles the back end fills.
pub fn render(image: &ImageData, sample_rate: u32, seed: u64) -> Result<Vec<Vec3>, String> {
    let mut flat = FlatScene::from(&image.camera, &image.scene);
    let scene = flat.as_rtm();
    let params = rtm_params {
        width: image.width,
        height: image.height,
        spp: sample_rate,  // args.sample_rate (main.rs:27)
        max_bounces: 10,   // main.rs:63
        seed,              // replaces the OS-seeded thread_rng (main.rs:162): frames are reproducible
        row_begin: 0,
        row_stride: 1,
        flags: 0,
        reserved: 0,
    };
    let mut pixels = vec![Vec3::zero(); (image.width * image.height) as usize]; // main.rs:153
    unsafe {
        if rtm_init(-1) != RTM_OK {
            return Err(last_error());
        }
        let rc = rtm_render(&scene, &params, pixels.as_mut_ptr() as *mut f64, std::ptr::null_mut(), std::ptr::null_mut(),
                            std::ptr::null_mut());
        if rc != RTM_OK {
            return Err(last_error());
        }
    }
    Ok(pixels)
}
