//! gpu.rs — `extern "C"` binding of include/rtc_hip.h (ABI version 2) for the reference crate, plus the flattener
//! and the two calls it replaces: `Camera::render` (camera.rs:67-80) and `World::color_at` (world.rs:80-82).
//!
//! Drop this file into the crate's `src/`, add `mod gpu;` to main.rs and the build.rs next to it.  The crate needs
//! three small accessors for fields that are private today (they are written out at the bottom of this file).
//!
//! Every `#[repr(C)]` struct below mirrors the C struct of the same name field for field; tests/test_rust_binding.py
//! parses this file and compares names, order, types and sizes with include/rtc_hip.h and the ctypes mirror.
//! Not compiled in the build image (no Rust toolchain there).
#![allow(dead_code)]

use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const RTC_ABI_VERSION: c_int = 3;

pub const RTC_OK: c_int = 0;
pub const RTC_ERR_INVALID_ARG: c_int = -1;
pub const RTC_ERR_NO_DEVICE: c_int = -2;
pub const RTC_ERR_HIP: c_int = -3;
pub const RTC_ERR_NON_AFFINE: c_int = -4;
pub const RTC_ERR_DEPTH_UNDERFLOW: c_int = -5;
pub const RTC_ERR_UNSUPPORTED: c_int = -6;
pub const RTC_ERR_TOO_MANY_GROUPS: c_int = -7;
pub const RTC_ERR_COLLECTIVE: c_int = -8;
pub const RTC_ERR_RAY_POOL: c_int = -9;

pub const RTC_SPHERE: u8 = 0;
pub const RTC_PLANE: u8 = 1;
pub const RTC_CUBE: u8 = 2;
pub const RTC_CYLINDER: u8 = 3;
pub const RTC_CONE: u8 = 4;
pub const RTC_TRIANGLE: u8 = 5;

pub const RTC_PATTERN_STRIPE: i32 = 1;
pub const RTC_PATTERN_GRADIENT: i32 = 2;
pub const RTC_PATTERN_RING: i32 = 3;
pub const RTC_PATTERN_CHECKERS: i32 = 4;
pub const RTC_PATTERN_TEST: i32 = 5;

pub const RTC_FLAG_TIMING: u32 = 1;
pub const RTC_FLAG_COUNT_RAYS: u32 = 2;
pub const RTC_FLAG_EXACT_SWEEP: u32 = 4;
pub const RTC_FLAG_GATHER_P2P: u32 = 8;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtcPattern {
    pub kind: i32,
    pub _pad: i32,
    pub a: [f64; 3],
    pub b: [f64; 3],
    pub transform_inverse: [f64; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtcMaterial {
    pub color: [f64; 3],
    pub ambient: f64,
    pub diffuse: f64,
    pub specular: f64,
    pub shininess: f64,
    pub reflective: f64,
    pub transparency: f64,
    pub refractive_index: f64,
    pub pattern: i32,
    pub _pad: i32,
}

#[repr(C)]
pub struct RtcSceneDesc {
    pub light_position: [f64; 3],
    pub light_intensity: [f64; 3],
    pub n_transforms: u32,
    pub transform_inverse: *const f64,
    pub n_materials: u32,
    pub materials: *const RtcMaterial,
    pub n_patterns: u32,
    pub patterns: *const RtcPattern,
    pub n_groups: u32,
    pub group_parent: *const i32,
    pub group_bounds_min: *const f64,
    pub group_bounds_max: *const f64,
    pub n_leaves: u32,
    pub leaf_kind: *const u8,
    pub leaf_transform: *const u32,
    pub leaf_material: *const u32,
    pub leaf_group: *const i32,
    pub leaf_class: *const u32,
    pub leaf_params: *const f64,
    pub leaf_triangle: *const u32,
    pub n_triangles: u32,
    pub tri_p1: *const f64,
    pub tri_e1: *const f64,
    pub tri_e2: *const f64,
    pub tri_normal: *const f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtcCameraDesc {
    pub hsize: u32,
    pub vsize: u32,
    pub half_width: f64,
    pub half_height: f64,
    pub pixel_size: f64,
    pub transform_inverse: [f64; 16],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtcRenderOpts {
    pub recursion_limit: i32,
    pub block_rows: u32,
    pub block_first: u32,
    pub block_step: u32,
    pub flags: u32,
    pub block_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct RtcStats {
    pub radiance_rays: u64,
    pub shadow_rays: u64,
    pub triangle_tests: u64,
    pub primitive_tests: u64,
    pub filter_tests: u64,
    pub box_tests: u64,
    pub culled_rays: u64,
    pub unswept_shadow_rays: u64,
    pub kernel_ms: f64,
    pub upload_ms: f64,
    pub download_ms: f64,
}

pub enum RtcScene {}
pub enum RtcMulti {}

extern "C" {
    pub fn rtc_abi_version() -> c_int;
    pub fn rtc_device_count() -> c_int;
    pub fn rtc_last_error_string() -> *const c_char;
    pub fn rtc_local_rows(vsize: u32, opts: *const RtcRenderOpts) -> u32;
    pub fn rtc_scene_create(desc: *const RtcSceneDesc, device: c_int, out: *mut *mut RtcScene) -> c_int;
    pub fn rtc_scene_destroy(scene: *mut RtcScene);
    pub fn rtc_render_device(scene: *mut RtcScene, camera: *const RtcCameraDesc, opts: *const RtcRenderOpts,
                             d_rgb: *mut f32, d_rgb64: *mut f64, hip_stream: *mut c_void) -> c_int;
    pub fn rtc_scene_stats(scene: *mut RtcScene, stats: *mut RtcStats) -> c_int;
    pub fn rtc_scene_reset_counters(scene: *mut RtcScene) -> c_int;
    pub fn rtc_render(desc: *const RtcSceneDesc, camera: *const RtcCameraDesc, opts: *const RtcRenderOpts,
                      device: c_int, h_rgb: *mut f32, h_rgb64: *mut f64, stats: *mut RtcStats) -> c_int;
    pub fn rtc_multi_create(desc: *const RtcSceneDesc, n_devices: c_int, devices: *const c_int,
                            out: *mut *mut RtcMulti) -> c_int;
    pub fn rtc_multi_destroy(multi: *mut RtcMulti);
    pub fn rtc_multi_render(multi: *mut RtcMulti, camera: *const RtcCameraDesc, opts: *const RtcRenderOpts,
                            h_rgb: *mut f32, h_rgb64: *mut f64, stats: *mut RtcStats) -> c_int;
    pub fn rtc_render_multi(desc: *const RtcSceneDesc, camera: *const RtcCameraDesc, opts: *const RtcRenderOpts,
                            n_devices: c_int, devices: *const c_int, h_rgb: *mut f32, h_rgb64: *mut f64,
                            stats: *mut RtcStats) -> c_int;
    pub fn rtc_release_cached();
    pub fn rtc_color_at(scene: *mut RtcScene, n: u64, origins: *const f64, directions: *const f64,
                        opts: *const RtcRenderOpts, h_rgb: *mut f32, h_rgb64: *mut f64) -> c_int;
    pub fn rtc_selftest_math(device: c_int, op: c_int, n: u64, a: *const f64, b: *const f64, out: *mut f64) -> c_int;
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(rtc_last_error_string()) }.to_string_lossy().into_owned()
}

// ------------------------------------------------------------------------------------------------------
// Flattener: World -> the SoA tables RtcSceneDesc points into (what host/flatten.cpp does in the C++ mirror).
// `Shape::set_transform` already pushed group transforms down to the leaves (shape.rs:207-218), so every leaf
// carries its full `transform_inverse()`; groups survive as (parent, Bounds::new(group)).
// ------------------------------------------------------------------------------------------------------
use crate::bounds::Bounds;
use crate::camera::Camera;
use crate::canvas::Canvas;
use crate::color::Color;
use crate::material::Material;
use crate::matrix::Matrix;
use crate::ray::Ray;
use crate::shape::{Shape, ShapeKind};
use crate::world::{World, RECURSION_LIMIT};

#[derive(Default)]
pub struct FlatScene {
    pub light_position: [f64; 3],
    pub light_intensity: [f64; 3],
    pub transform_inverse: Vec<f64>,
    pub materials: Vec<RtcMaterial>,
    pub patterns: Vec<RtcPattern>,
    pub group_parent: Vec<i32>,
    pub group_bounds_min: Vec<f64>,
    pub group_bounds_max: Vec<f64>,
    pub leaf_kind: Vec<u8>,
    pub leaf_transform: Vec<u32>,
    pub leaf_material: Vec<u32>,
    pub leaf_group: Vec<i32>,
    pub leaf_class: Vec<u32>,
    pub leaf_params: Vec<f64>,
    pub leaf_triangle: Vec<u32>,
    pub tri_p1: Vec<f64>,
    pub tri_e1: Vec<f64>,
    pub tri_e2: Vec<f64>,
    pub tri_normal: Vec<f64>,
}

fn matrix_rows(m: &Matrix<4>) -> [f64; 16] {
    let mut out = [0.0; 16];
    for r in 0..4 {
        for c in 0..4 {
            out[4 * r + c] = m[r][c];
        }
    }
    out
}

fn bits(v: &[f64]) -> Vec<u64> {
    v.iter().map(|x| x.to_bits()).collect()
}

struct Flattener<'a> {
    out: FlatScene,
    leaves: Vec<&'a Shape>,
    xf_ids: HashMap<Vec<u64>, u32>,
    mat_ids: HashMap<Vec<u64>, u32>,
}

impl<'a> Flattener<'a> {
    fn transform_id(&mut self, inv: &Matrix<4>) -> u32 {
        let rows = matrix_rows(inv);
        let key = bits(&rows);
        if let Some(id) = self.xf_ids.get(&key) {
            return *id;
        }
        let id = (self.out.transform_inverse.len() / 16) as u32;
        self.out.transform_inverse.extend_from_slice(&rows);
        self.xf_ids.insert(key, id);
        id
    }

    fn material_id(&mut self, m: &Material) -> u32 {
        let mut flat = vec![m.color.red, m.color.green, m.color.blue, m.ambient, m.diffuse, m.specular, m.shininess,
                            m.reflective, m.transparency, m.refractive_index];
        let pat = m.pattern.as_ref().map(|p| p.to_ffi()); // accessor to add in pattern.rs, see the end of this file
        if let Some(p) = &pat {
            flat.push(p.kind as f64);
            flat.extend_from_slice(&p.a);
            flat.extend_from_slice(&p.b);
            flat.extend_from_slice(&p.transform_inverse);
        }
        let key = bits(&flat);
        if let Some(id) = self.mat_ids.get(&key) {
            return *id;
        }
        let mut r = RtcMaterial {
            color: [m.color.red, m.color.green, m.color.blue],
            ambient: m.ambient, diffuse: m.diffuse, specular: m.specular, shininess: m.shininess,
            reflective: m.reflective, transparency: m.transparency, refractive_index: m.refractive_index,
            pattern: -1, _pad: 0,
        };
        if let Some(p) = pat {
            r.pattern = self.out.patterns.len() as i32;
            self.out.patterns.push(p);
        }
        let id = self.out.materials.len() as u32;
        self.out.materials.push(r);
        self.mat_ids.insert(key, id);
        id
    }

    fn visit(&mut self, s: &'a Shape, parent: i32) {
        if let ShapeKind::Group { shapes } = &s.kind {
            let gid = self.out.group_parent.len() as i32;
            self.out.group_parent.push(parent);
            let b = Bounds::new(s); // bounds.rs:11-140, once per group instead of once per ray (shape.rs:401)
            self.out.group_bounds_min.extend_from_slice(&[b.min.x, b.min.y, b.min.z]);
            self.out.group_bounds_max.extend_from_slice(&[b.max.x, b.max.y, b.max.z]);
            for c in shapes {
                self.visit(c, gid);
            }
            return;
        }
        let (kind, params) = match &s.kind {
            ShapeKind::Sphere => (RTC_SPHERE, [0.0, 0.0, 0.0]),
            ShapeKind::Plane => (RTC_PLANE, [0.0, 0.0, 0.0]),
            ShapeKind::Cube => (RTC_CUBE, [0.0, 0.0, 0.0]),
            ShapeKind::Cylinder { minimum, maximum, capped } => (RTC_CYLINDER, [*minimum, *maximum, if *capped { 1.0 } else { 0.0 }]),
            ShapeKind::Cone { minimum, maximum, capped } => (RTC_CONE, [*minimum, *maximum, if *capped { 1.0 } else { 0.0 }]),
            ShapeKind::Triangle { .. } => (RTC_TRIANGLE, [0.0, 0.0, 0.0]),
            ShapeKind::Group { .. } => unreachable!(),
        };
        self.leaves.push(s);
        self.out.leaf_kind.push(kind);
        let xf = self.transform_id(&s.transform_inverse());
        self.out.leaf_transform.push(xf);
        let mat = self.material_id(s.get_material());
        self.out.leaf_material.push(mat);
        self.out.leaf_group.push(parent);
        self.out.leaf_params.extend_from_slice(&params);
        if let ShapeKind::Triangle { p1, e1, e2, normal, .. } = &s.kind {
            self.out.leaf_triangle.push((self.out.tri_p1.len() / 3) as u32);
            self.out.tri_p1.extend_from_slice(&[p1.x, p1.y, p1.z]);
            self.out.tri_e1.extend_from_slice(&[e1.x, e1.y, e1.z]);
            self.out.tri_e2.extend_from_slice(&[e2.x, e2.y, e2.z]);
            self.out.tri_normal.extend_from_slice(&[normal.x, normal.y, normal.z]);
        } else {
            self.out.leaf_triangle.push(0);
        }
    }

    /// leaf_class[i] = first leaf j <= i with leaf_j == leaf_i under Shape's value PartialEq (shape.rs:638-646):
    /// prepare_computations compares container shapes that way (intersection.rs:42).  Only consumed when some
    /// material is transparent; quadratic in the number of non-triangle leaves, near-linear for meshes.
    fn classes(&mut self) {
        let n = self.leaves.len();
        self.out.leaf_class = (0..n as u32).collect();
        let mut tri_by_x: Vec<(f64, usize)> = Vec::new();
        let mut prims: Vec<usize> = Vec::new();
        for (i, s) in self.leaves.iter().enumerate() {
            match &s.kind {
                ShapeKind::Triangle { p1, .. } => tri_by_x.push((p1.x, i)),
                _ => prims.push(i),
            }
        }
        tri_by_x.sort_by(|a, b| a.partial_cmp(b).unwrap_or(std::cmp::Ordering::Equal));
        for i in 0..n {
            let s = self.leaves[i];
            let mut best = i;
            if let ShapeKind::Triangle { p1, .. } = &s.kind {
                let lo = tri_by_x.partition_point(|(x, _)| *x < p1.x - 1e-5);
                for (x, j) in &tri_by_x[lo..] {
                    if *x >= p1.x + 1e-5 {
                        break;
                    }
                    if *j < best && self.leaves[*j] == s {
                        best = *j;
                    }
                }
            } else {
                for j in &prims {
                    if *j >= best {
                        break;
                    }
                    if self.leaves[*j] == s {
                        best = *j;
                        break;
                    }
                }
            }
            self.out.leaf_class[i] = best as u32;
        }
    }
}

impl World {
    pub fn flatten(&self) -> FlatScene {
        let mut f = Flattener { out: FlatScene::default(), leaves: Vec::new(), xf_ids: HashMap::new(), mat_ids: HashMap::new() };
        let (p, c) = (&self.light.position, &self.light.intensity);
        f.out.light_position = [p.x, p.y, p.z];
        f.out.light_intensity = [c.red, c.green, c.blue];
        for s in &self.objects {
            f.visit(s, -1);
        }
        if f.out.materials.iter().any(|m| m.transparency != 0.0) {
            f.classes();
        }
        f.out
    }

    /// World::color_at (world.rs:80-82) on the GPU, for a batch of rays.
    pub fn color_at_gpu(&self, rays: &[Ray], device: i32) -> Vec<Color> {
        let flat = self.flatten();
        let desc = flat.desc();
        let mut scene: *mut RtcScene = std::ptr::null_mut();
        let rc = unsafe { rtc_scene_create(&desc, device, &mut scene) };
        if rc != RTC_OK {
            panic!("rtc_scene_create: {}", last_error());
        }
        let mut o = Vec::with_capacity(rays.len() * 3);
        let mut d = Vec::with_capacity(rays.len() * 3);
        for r in rays {
            o.extend_from_slice(&[r.origin.x, r.origin.y, r.origin.z]);
            d.extend_from_slice(&[r.direction.x, r.direction.y, r.direction.z]);
        }
        let opts = RtcRenderOpts { recursion_limit: RECURSION_LIMIT as i32, ..Default::default() };
        let mut out = vec![0f64; rays.len() * 3];
        let rc = unsafe { rtc_color_at(scene, rays.len() as u64, o.as_ptr(), d.as_ptr(), &opts, std::ptr::null_mut(), out.as_mut_ptr()) };
        unsafe { rtc_scene_destroy(scene) };
        if rc != RTC_OK {
            panic!("rtc_color_at: {}", last_error());
        }
        out.chunks(3).map(|c| Color::new(c[0], c[1], c[2])).collect()
    }
}

impl FlatScene {
    /// Borrows self's Vecs for the duration of one call, like `&World`.
    pub fn desc(&self) -> RtcSceneDesc {
        RtcSceneDesc {
            light_position: self.light_position,
            light_intensity: self.light_intensity,
            n_transforms: (self.transform_inverse.len() / 16) as u32,
            transform_inverse: self.transform_inverse.as_ptr(),
            n_materials: self.materials.len() as u32,
            materials: self.materials.as_ptr(),
            n_patterns: self.patterns.len() as u32,
            patterns: self.patterns.as_ptr(),
            n_groups: self.group_parent.len() as u32,
            group_parent: self.group_parent.as_ptr(),
            group_bounds_min: self.group_bounds_min.as_ptr(),
            group_bounds_max: self.group_bounds_max.as_ptr(),
            n_leaves: self.leaf_kind.len() as u32,
            leaf_kind: self.leaf_kind.as_ptr(),
            leaf_transform: self.leaf_transform.as_ptr(),
            leaf_material: self.leaf_material.as_ptr(),
            leaf_group: self.leaf_group.as_ptr(),
            leaf_class: if self.leaf_class.is_empty() { std::ptr::null() } else { self.leaf_class.as_ptr() },
            leaf_params: self.leaf_params.as_ptr(),
            leaf_triangle: self.leaf_triangle.as_ptr(),
            n_triangles: (self.tri_p1.len() / 3) as u32,
            tri_p1: self.tri_p1.as_ptr(),
            tri_e1: self.tri_e1.as_ptr(),
            tri_e2: self.tri_e2.as_ptr(),
            tri_normal: self.tri_normal.as_ptr(),
        }
    }
}

impl Camera {
    /// Camera::render (camera.rs:67-80) on one MI355X, or, with more than one entry in `devices`, with the frame's
    /// rows sharded over those devices and gathered to devices[0] over xGMI (rtc_render_multi).
    /// The device copy of the scene stays resident between calls with an identical world (include/rtc_hip.h).
    pub fn render_gpu(&self, world: &World, devices: &[i32]) -> Canvas {
        let flat = world.flatten();
        let desc = flat.desc();
        let cam = self.to_ffi(); // accessor to add in camera.rs, see the end of this file
        let opts = RtcRenderOpts { recursion_limit: RECURSION_LIMIT as i32, ..Default::default() };
        let n = (cam.hsize as usize) * (cam.vsize as usize) * 3;
        let mut rgb64 = vec![0f64; n];
        let rc = unsafe {
            if devices.len() > 1 {
                rtc_render_multi(&desc, &cam, &opts, devices.len() as c_int, devices.as_ptr(), std::ptr::null_mut(),
                                 rgb64.as_mut_ptr(), std::ptr::null_mut())
            } else {
                rtc_render(&desc, &cam, &opts, devices.first().copied().unwrap_or(0), std::ptr::null_mut(),
                           rgb64.as_mut_ptr(), std::ptr::null_mut())
            }
        };
        if rc != RTC_OK {
            panic!("rtc_render: {}", last_error()); // the reference panics where the C ABI returns a status
        }
        let (w, h) = (cam.hsize as usize, cam.vsize as usize);
        let mut image = Canvas::new(w, h);
        for y in 0..h {
            for x in 0..w {
                let i = 3 * (x + y * w); // Canvas layout: x + y*width, top row first (canvas.rs:20-26)
                image.set_pixel(x, y, Color::new(rgb64[i], rgb64[i + 1], rgb64[i + 2]));
            }
        }
        image
    }
}

// ------------------------------------------------------------------------------------------------------
// Accessors the crate has to add (the fields are private today):
//
// pattern.rs (PatternKind and Pattern's fields are private, pattern.rs:4-19):
//     impl Pattern {
//         pub(crate) fn to_ffi(&self) -> crate::gpu::RtcPattern {
//             let (kind, a, b) = match self.kind {
//                 PatternKind::Stripe(a, b) => (crate::gpu::RTC_PATTERN_STRIPE, a, b),
//                 PatternKind::Gradient(a, b) => (crate::gpu::RTC_PATTERN_GRADIENT, a, b),
//                 PatternKind::Ring(a, b) => (crate::gpu::RTC_PATTERN_RING, a, b),
//                 PatternKind::Checkers(a, b) => (crate::gpu::RTC_PATTERN_CHECKERS, a, b),
//                 PatternKind::Test => (crate::gpu::RTC_PATTERN_TEST, BLACK, BLACK),
//             };
//             let mut m = [0.0; 16];
//             for r in 0..4 { for c in 0..4 { m[4 * r + c] = self.transform_inverse[r][c]; } }
//             crate::gpu::RtcPattern { kind, _pad: 0, a: [a.red, a.green, a.blue], b: [b.red, b.green, b.blue], transform_inverse: m }
//         }
//     }
//
// camera.rs (all eight fields are private, camera.rs:4-13):
//     impl Camera {
//         pub(crate) fn to_ffi(&self) -> crate::gpu::RtcCameraDesc {
//             let mut m = [0.0; 16];
//             for r in 0..4 { for c in 0..4 { m[4 * r + c] = self.transform_inverse[r][c]; } }
//             crate::gpu::RtcCameraDesc { hsize: self.hsize as u32, vsize: self.vsize as u32, half_width: self.half_width,
//                                         half_height: self.half_height, pixel_size: self.pixel_size, transform_inverse: m }
//         }
//     }
//
// world.rs: `const RECURSION_LIMIT` (world.rs:11) becomes `pub(crate) const`.
// camera.rs:67: `pub fn render(&self, world: &World) -> Canvas { self.render_gpu(world, &[0]) }`.
