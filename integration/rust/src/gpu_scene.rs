//! gpu_scene.rs -- the reference crate's `Scene` on the GPU.
//!
//! `GpuScene` owns an `rtx_context` holding the flattened scene and its tree. It implements
//! `IntersectableScene` (src/scene/mod.rs:148-154), so `RayTracer::set_scene` (src/lib.rs:64-67) takes it like
//! `Scene` or `BvhScene`; `render` is the whole-frame replacement for the pixel loop of `RayTracer::trace_rays`
//! (src/lib.rs:257-277). Written against the crate's early-2015 types; never compiled (no Rust toolchain here).
use std::ffi::{CStr, CString};
use std::ptr;

use bmp::{Image, Pixel};
use ffi::*;
use ray::Ray;
use scene::intersection::Intersection;
use scene::shapes::Primitive;
use scene::material::{Color, Material};
use scene::{Camera, IntersectableScene, Light, Scene, SceneIntersection};
use vec::Vec3;

pub struct GpuScene {
    ctx: *mut rtx_context,
    scene: Scene, // keeps camera, lights and primitives for get_camera / get_lights / the returned Intersection
    pub build: rtx_build_stats,
}

fn check(ctx: *const rtx_context, status: i32, what: &str) {
    if status != RTX_OK {
        let msg = unsafe { CStr::from_ptr(rtx_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("{}: {}", what, msg); // the reference's error convention (src/lib.rs:72,275)
    }
}

// The crate targets an early-2015 nightly: only `Vec::push` / `iter` / `len` / `as_ptr` are used on vectors below
// (no `extend_from_slice`, which arrived later).
fn push3(out: &mut Vec<f32>, vec: &Vec3) {
    out.push(vec.x);
    out.push(vec.y);
    out.push(vec.z);
}

fn rgb(color: &Color) -> [f32; 3] {
    [color.r_val(), color.g_val(), color.b_val()] // the fields r, g, b are private (src/scene/material.rs:6-10)
}

fn flat_material(mat: &Material) -> rtx_material {
    rtx_material {
        diffuse: rgb(&mat.diffuse),
        ambient: rgb(&mat.ambient),
        specular: rgb(&mat.specular),
        emissive: rgb(&mat.emissive),
        shininess: mat.shininess,
        transparency: mat.transparency,
    }
}

impl GpuScene {
    /// Flattening rules: primitive id = index in `scene.primitives` (the parser has already reversed every poly_set,
    /// src/scene/parser/mod.rs:333-338); a poly's `materials` is its de-duplicated list (parser/mod.rs:266-291).
    /// Field names are the reference's: `Vertex.position` / `.normal` (src/scene/shapes/poly.rs:10-15),
    /// `Sphere.origin` / `.radius` (sphere.rs:8-11), `PointLight.pos` (src/scene/mod.rs:63-66), `AreaLight.min` / `.max`
    /// (mod.rs:78-82), `DirectionalLight.dir` (mod.rs:108-111), `Camera.*` (mod.rs:123-129).
    pub fn new(scene: Scene, device: i32, force_bvh: bool) -> GpuScene {
        let mut tri_pos: Vec<f32> = Vec::new();
        let mut tri_nrm: Vec<f32> = Vec::new();
        let mut tri_mat: Vec<u32> = Vec::new();
        let mut tri_flags: Vec<u8> = Vec::new();
        let mut tri_index: Vec<u32> = Vec::new();
        let mut sph: Vec<f32> = Vec::new();
        let mut sph_mat: Vec<u32> = Vec::new();
        let mut sph_index: Vec<u32> = Vec::new();
        let mut materials: Vec<rtx_material> = Vec::new();
        for (id, prim) in scene.primitives.iter().enumerate() {
            match *prim {
                Primitive::Poly(ref poly) => {
                    for vert in poly.vertices.iter() {
                        push3(&mut tri_pos, &vert.position);
                        push3(&mut tri_nrm, &vert.normal);
                    }
                    for k in 0..3 {
                        if k < poly.materials.len() {
                            materials.push(flat_material(&poly.materials[k]));
                            tri_mat.push((materials.len() - 1) as u32);
                        } else {
                            tri_mat.push(RTX_MATERIAL_NONE);
                        }
                    }
                    tri_flags.push((if poly.vertex_normal { RTX_TRI_VERTEX_NORMAL } else { 0 })
                        | (if poly.vertex_material { RTX_TRI_VERTEX_MATERIAL } else { 0 }));
                    tri_index.push(id as u32);
                }
                Primitive::Sphere(ref sphere) => {
                    push3(&mut sph, &sphere.origin);
                    sph.push(sphere.radius);
                    materials.push(flat_material(&sphere.materials[0]));
                    sph_mat.push((materials.len() - 1) as u32);
                    sph_index.push(id as u32);
                }
            }
        }
        let mut lights: Vec<rtx_light> = Vec::new();
        for light in scene.lights.iter() {
            lights.push(match *light {
                Light::Point(ref point) => rtx_light { type_: RTX_LIGHT_POINT, a: [point.pos.x, point.pos.y, point.pos.z], b: [0.0; 3],
                                                       color: rgb(&point.intensity) },
                Light::Area(ref area) => rtx_light { type_: RTX_LIGHT_AREA, a: [area.min.x, area.min.y, area.min.z],
                                                     b: [area.max.x, area.max.y, area.max.z], color: rgb(&area.intensity) },
                Light::Directional(ref directional) => rtx_light { type_: RTX_LIGHT_DIRECTIONAL,
                                                                   a: [directional.dir.x, directional.dir.y, directional.dir.z],
                                                                   b: [0.0; 3], color: rgb(&directional.intensity) },
            });
        }
        let cam = &scene.camera;
        let desc = rtx_scene_desc {
            n_tris: tri_index.len() as u32,
            tri_pos: tri_pos.as_ptr(),
            tri_normals: tri_nrm.as_ptr(),
            tri_mat: tri_mat.as_ptr(),
            tri_flags: tri_flags.as_ptr(),
            tri_prim_index: tri_index.as_ptr(),
            n_spheres: sph_index.len() as u32,
            sph_center_radius: sph.as_ptr(),
            sph_mat: sph_mat.as_ptr(),
            sph_prim_index: sph_index.as_ptr(),
            n_materials: materials.len() as u32,
            materials: materials.as_ptr(),
            n_lights: lights.len() as u32,
            lights: lights.as_ptr(),
            camera: rtx_camera {
                pos: [cam.pos.x, cam.pos.y, cam.pos.z],
                view_dir: [cam.view_dir.x, cam.view_dir.y, cam.view_dir.z],
                ortho_up: [cam.ortho_up.x, cam.ortho_up.y, cam.ortho_up.z],
                focal_dist: cam.focal_dist,
                vertical_fov: cam.vertical_fov,
            },
        };
        let mut ctx: *mut rtx_context = ptr::null_mut();
        check(ptr::null(), unsafe { rtx_create(device, &mut ctx) }, "rtx_create");
        if force_bvh {
            let key = CString::new("force_mode").unwrap();
            check(ctx, unsafe { rtx_set_option(ctx, key.as_ptr(), 2) }, "rtx_set_option");
        }
        check(ctx, unsafe { rtx_scene_set(ctx, &desc) }, "rtx_scene_set");
        let mut build = rtx_build_stats::default();
        check(ctx, unsafe { rtx_build(ctx, &mut build) }, "rtx_build");
        GpuScene { ctx: ctx, scene: scene, build: build }
    }

    /// Batched `intersects`: (primitive index or -1, t) per ray.
    pub fn intersect_batch(&self, rays: &[Ray]) -> Vec<(i32, f32)> {
        let mut ori = Vec::with_capacity(rays.len() * 3);
        let mut dir = Vec::with_capacity(rays.len() * 3);
        for ray in rays.iter() {
            push3(&mut ori, &ray.ori);
            push3(&mut dir, &ray.dir);
        }
        let mut prim: Vec<i32> = Vec::with_capacity(rays.len());
        let mut t: Vec<f32> = Vec::with_capacity(rays.len());
        for _ in 0..rays.len() {
            prim.push(-1);
            t.push(0.0);
        }
        check(self.ctx, unsafe { rtx_intersect_batch(self.ctx, rays.len() as u64, ori.as_ptr(), dir.as_ptr(),
                                                      prim.as_mut_ptr(), t.as_mut_ptr()) }, "rtx_intersect_batch");
        prim.into_iter().zip(t.into_iter()).collect()
    }

    /// The whole frame: what the loop of `RayTracer::trace_rays` (src/lib.rs:262-271) produces, row 0 = top row.
    pub fn render(&self, width: u32, height: u32, depth: usize, num_samples: usize) -> (Image, rtx_render_stats) {
        let n_bytes = (width as usize) * (height as usize) * 3;
        let mut rgb: Vec<u8> = Vec::with_capacity(n_bytes);
        for _ in 0..n_bytes {
            rgb.push(0);
        }
        let mut stats = rtx_render_stats::default();
        check(self.ctx, unsafe { rtx_render(self.ctx, width, height, depth as u32, num_samples as u32, rgb.as_mut_ptr(), &mut stats) },
              "rtx_render");
        let mut img = Image::new(width, height);
        for (x, y) in img.coordinates() {
            let i = ((y * width + x) * 3) as usize;
            img.set_pixel(x, y, Pixel { r: rgb[i], g: rgb[i + 1], b: rgb[i + 2] }); // where src/lib.rs:268 sets it
        }
        (img, stats)
    }
}

impl Drop for GpuScene {
    fn drop(&mut self) {
        unsafe { rtx_destroy(self.ctx) }
    }
}

impl<'a> IntersectableScene<'a> for GpuScene {
    fn get_camera(&self) -> &Camera {
        &self.scene.camera
    }
    fn get_lights(&self) -> &[Light] {
        &self.scene.lights
    }
    /// One-ray form of the batched call (a device round trip per ray: for parity checks, not for rendering).
    fn intersects(&'a self, ray: &Ray) -> SceneIntersection<'a> {
        let (prim, t) = self.intersect_batch(&[ray.clone()])[0];
        if prim < 0 {
            SceneIntersection::Missed
        } else {
            SceneIntersection::Intersected(Intersection::new(t, ray.clone(), &self.scene.primitives[prim as usize]))
        }
    }
}
