//! ffi.rs -- `#[repr(C)]` mirrors of include/rtx.h and the `extern "C"` block.
//! Field order and types follow the header one for one (checked by tests/test_rust_shim.py).
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

pub const RTX_ABI_VERSION: u32 = 2;
pub const RTX_OK: c_int = 0;
pub const RTX_ERR_INVALID_ARG: c_int = 1;
pub const RTX_ERR_NO_DEVICE: c_int = 2;
pub const RTX_ERR_HIP: c_int = 3;
pub const RTX_ERR_NO_SCENE: c_int = 4;
pub const RTX_ERR_NOT_BUILT: c_int = 5;
pub const RTX_ERR_UNSUPPORTED: c_int = 6;
pub const RTX_ERR_OUT_OF_MEMORY: c_int = 7;
pub const RTX_MAX_DEPTH: u32 = 32;
pub const RTX_MATERIAL_NONE: u32 = 0xFFFF_FFFF;
pub const RTX_LIGHT_POINT: u32 = 0;
pub const RTX_LIGHT_AREA: u32 = 1;
pub const RTX_LIGHT_DIRECTIONAL: u32 = 2;
pub const RTX_TRI_VERTEX_NORMAL: u8 = 1;
pub const RTX_TRI_VERTEX_MATERIAL: u8 = 2;
pub const RTX_NOALLOC_MAX_RAYS: u64 = 1 << 26;
pub const RTX_COMM_ID_BYTES: usize = 128;
pub const RTX_MAX_BATCH_FRAMES: u32 = 64;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rtx_material {
    pub diffuse: [f32; 3],
    pub ambient: [f32; 3],
    pub specular: [f32; 3],
    pub emissive: [f32; 3],
    pub shininess: f32,
    pub transparency: f32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rtx_light {
    pub type_: u32,
    pub a: [f32; 3],
    pub b: [f32; 3],
    pub color: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rtx_camera {
    pub pos: [f32; 3],
    pub view_dir: [f32; 3],
    pub ortho_up: [f32; 3],
    pub focal_dist: f32,
    pub vertical_fov: f32,
}

#[repr(C)]
pub struct rtx_scene_desc {
    pub n_tris: u32,
    pub tri_pos: *const f32,
    pub tri_normals: *const f32,
    pub tri_mat: *const u32,
    pub tri_flags: *const u8,
    pub tri_prim_index: *const u32,
    pub n_spheres: u32,
    pub sph_center_radius: *const f32,
    pub sph_mat: *const u32,
    pub sph_prim_index: *const u32,
    pub n_materials: u32,
    pub materials: *const rtx_material,
    pub n_lights: u32,
    pub lights: *const rtx_light,
    pub camera: rtx_camera,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rtx_build_stats {
    pub n_prims: u32,
    pub n_nodes: u32,
    pub n_leaves: u32,
    pub max_depth: u32,
    pub uses_bvh: u32,
    pub key_bits: u32,
    pub build_ms: f32,
    pub upload_ms: f32,
    pub n_always: u32,
    pub reserved: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rtx_render_stats {
    pub primary_rays: u64,
    pub shadow_rays: u64,
    pub reflect_rays: u64,
    pub refract_rays: u64,
    pub skipped_rays: u64,
    pub shaded_nodes: u64,
    pub panic_pixels: u64,
    pub render_ms: f32,
    pub closest_ms: f32,
    pub shadow_ms: f32,
    pub closest_launches: u32,
    pub shadow_launches: u32,
    pub batches: u32,
    pub reserved: u32,
}

/// Opaque: lives in the library.
pub enum rtx_context {}

extern "C" {
    pub fn rtx_abi_version() -> c_int;
    pub fn rtx_source_hash() -> *const c_char;
    pub fn rtx_create(device: c_int, out_ctx: *mut *mut rtx_context) -> c_int;
    pub fn rtx_destroy(ctx: *mut rtx_context);
    pub fn rtx_last_error(ctx: *const rtx_context) -> *const c_char;
    pub fn rtx_set_option(ctx: *mut rtx_context, key: *const c_char, value: i64) -> c_int;
    pub fn rtx_scene_set(ctx: *mut rtx_context, scene: *const rtx_scene_desc) -> c_int;
    pub fn rtx_scene_set_device(ctx: *mut rtx_context, scene: *const rtx_scene_desc) -> c_int;
    pub fn rtx_clone(src: *mut rtx_context, out_ctx: *mut *mut rtx_context) -> c_int;
    pub fn rtx_build(ctx: *mut rtx_context, stats: *mut rtx_build_stats) -> c_int;
    pub fn rtx_intersect_batch(ctx: *mut rtx_context, n: u64, ori3: *const f32, dir3: *const f32,
                               prim_id: *mut i32, t: *mut f32) -> c_int;
    pub fn rtx_intersect_batch_device(ctx: *mut rtx_context, n: u64, d_ori3: *const f32, d_dir3: *const f32,
                                      d_prim_id: *mut i32, d_t: *mut f32) -> c_int;
    pub fn rtx_intersect_batch_device_noalloc(ctx: *mut rtx_context, n: u64, d_ori3: *const f32, d_dir3: *const f32,
                                              d_prim_id: *mut i32, d_t: *mut f32) -> c_int;
    pub fn rtx_intersect_batch_exhaustive(ctx: *mut rtx_context, n: u64, ori3: *const f32, dir3: *const f32,
                                          prim_id: *mut i32, t: *mut f32) -> c_int;
    pub fn rtx_primary_hits(ctx: *mut rtx_context, width: u32, height: u32, prim_id: *mut i32, t: *mut f32,
                            ray_dir: *mut f32) -> c_int;
    pub fn rtx_render(ctx: *mut rtx_context, width: u32, height: u32, depth: u32, samples: u32, rgb: *mut u8,
                      stats: *mut rtx_render_stats) -> c_int;
    pub fn rtx_render_device(ctx: *mut rtx_context, width: u32, height: u32, depth: u32, samples: u32, d_rgb: *mut u8,
                             stats: *mut rtx_render_stats) -> c_int;
    pub fn rtx_render_device_async(ctx: *mut rtx_context, width: u32, height: u32, depth: u32, samples: u32,
                                   d_rgb: *mut u8) -> c_int;
    pub fn rtx_render_tiles_async(ctx: *mut rtx_context, width: u32, height: u32, depth: u32, samples: u32, rank: u32,
                                  world: u32, d_rgb_tiles: *mut u8) -> c_int;
    pub fn rtx_render_wait(ctx: *mut rtx_context, stats: *mut rtx_render_stats) -> c_int;
    pub fn rtx_render_tiles(ctx: *mut rtx_context, width: u32, height: u32, depth: u32, samples: u32, rank: u32,
                            world: u32, d_rgb_tiles: *mut u8, stats: *mut rtx_render_stats) -> c_int;
    pub fn rtx_tiles_per_rank(width: u32, height: u32, world: u32) -> u64;
    pub fn rtx_assemble_tiles(ctx: *mut rtx_context, width: u32, height: u32, world: u32, d_slabs: *const u8,
                              d_rgb: *mut u8) -> c_int;
    pub fn rtx_assemble_tiles_async(ctx: *mut rtx_context, width: u32, height: u32, world: u32, d_slabs: *const u8,
                                    d_rgb: *mut u8) -> c_int;
    pub fn rtx_render_frames(ctx: *mut rtx_context, n_frames: u32, cameras: *const rtx_camera, width: u32, height: u32,
                             depth: u32, samples: u32, rank: u32, world: u32, d_out: *mut u8, row_major: c_int,
                             stats: *mut rtx_render_stats) -> c_int;
    pub fn rtx_render_frames_async(ctx: *mut rtx_context, n_frames: u32, cameras: *const rtx_camera, width: u32,
                                   height: u32, depth: u32, samples: u32, rank: u32, world: u32, d_out: *mut u8,
                                   row_major: c_int) -> c_int;
    pub fn rtx_assemble_frames_async(ctx: *mut rtx_context, n_frames: u32, width: u32, height: u32, world: u32,
                                     d_slabs: *const u8, d_rgb: *mut u8) -> c_int;
    pub fn rtx_comm_unique_id(id128: *mut u8) -> c_int;
    pub fn rtx_comm_init(ctx: *mut rtx_context, rank: u32, world: u32, id128: *const u8) -> c_int;
    pub fn rtx_comm_destroy(ctx: *mut rtx_context) -> c_int;
    pub fn rtx_scene_broadcast(ctx: *mut rtx_context, scene: *const rtx_scene_desc, root: u32) -> c_int;
    pub fn rtx_gather_frame(ctx: *mut rtx_context, width: u32, height: u32, d_rgb_tiles: *const u8, d_rgb: *mut u8,
                            root: u32) -> c_int;
    pub fn rtx_gather_frames(ctx: *mut rtx_context, n_frames: u32, width: u32, height: u32, d_rgb_tiles: *const u8,
                             d_rgb: *mut u8, root: u32) -> c_int;
    pub fn rtx_debug_stamps(ctx: *mut rtx_context, kind: c_int, out: *mut u64, capacity: u64) -> u64;
    pub fn rtx_stream(ctx: *mut rtx_context) -> *mut c_void;
}
