//! `#[repr(C)]` mirrors of every struct of include/rt_abi.h and include/rt_checkpoint.h, and the `extern "C"` block.
//! Field order, types and sizes are checked against the C headers by tests/test_rust_binding.py (no rustc needed:
//! the test computes the repr(C) layout of these definitions and compares it with `offsetof` from gcc).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_BAD_ARG: c_int = -1;
pub const RT_ERR_HIP: c_int = -2;
pub const RT_ERR_OOM: c_int = -3;
pub const RT_ERR_UNSUPPORTED: c_int = -4;
pub const RT_ERR_BAD_BLOB: c_int = -5;
pub const RT_NONE: u32 = 0xFFFF_FFFF;

pub const RT_INT_MIS: i32 = 0; // MultipleImportanceSampleIntegrator   integrator.rs:29-171
pub const RT_INT_LIGHT_IS: i32 = 1; // ImportanceSampleLightIntegrator      integrator.rs:172-260
pub const RT_INT_BRDF: i32 = 2; // BRDFSampledPathIntegrator            integrator.rs:262-327
pub const RT_INT_UNIFORM: i32 = 3; // UniformSampledPathIntegrator         integrator.rs:329-397
pub const RT_BG_BLACK: u32 = 0; // no_light  desktop/src/main.rs:46
pub const RT_BG_SKY: u32 = 1; // skybox    desktop/src/main.rs:37

pub const RT_BLOB_MAGIC: u32 = 0x4253_5452;
pub const RT_BLOB_VERSION: u32 = 2;
pub const RT_NODE_NO_BOX: u32 = 1;
pub const RT_NODE_DUP_RIGHT: u32 = 2;
pub const RT_PRIM_TRIANGLE: u32 = 0;
pub const RT_PRIM_SPHERE: u32 = 1;
pub const RT_PRIM_MOVING_SPHERE: u32 = 2;
pub const RT_PRIM_AARECT: u32 = 3;
pub const RT_PRIM_INSTANCE: u32 = 4;
pub const RT_MAT_LAMBERTIAN: u32 = 0;
pub const RT_MAT_METAL: u32 = 1;
pub const RT_MAT_DIELECTRIC: u32 = 2;
pub const RT_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const RT_TEX_SOLID: u32 = 0;
pub const RT_TEX_CHECKER: u32 = 1;
pub const RT_TEX_IMAGE: u32 = 2;
pub const RT_RENDER_COUNT_TESTS: u32 = 1;
pub const RT_RENDER_NO_SORT: u32 = 2;
pub const RT_RENDER_ONE_LANE: u32 = 4;
pub const RT_RENDER_LANE_WALK: u32 = 8;
pub const RT_RENDER_MATERIAL_BINS: u32 = 16;
pub const RT_RENDER_COUNT_STEPS: u32 = 32;
pub const RT_RENDER_SERIAL_SHADOW: u32 = 64;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_node {
    pub bmin: [f64; 3],
    pub bmax: [f64; 3],
    pub skip: u32,
    pub flags: u32,
    pub prim0: u32,
    pub prim1: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_prim {
    pub kind: u32,
    pub index: u32,
    pub xform: u32,
    pub material: u32,
    pub object_id: u32,
    pub inst_id: u32,
    pub sub_end: u32,
    pub pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_geom {
    pub d: [f64; 9],
    pub kind: u32,
    pub xform: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_tri_attrs {
    pub n0: [f64; 3],
    pub n1: [f64; 3],
    pub n2: [f64; 3],
    pub uv0: [f64; 2],
    pub uv1: [f64; 2],
    pub uv2: [f64; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_xform {
    pub t_inv: [f64; 16],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_material {
    pub kind: u32,
    pub texture: u32,
    pub albedo: [f64; 3],
    pub fuzz: f64,
    pub ior: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_texture {
    pub kind: u32,
    pub a: u32,
    pub b: u32,
    pub c: u32,
    pub colour: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_light {
    pub center: [f64; 3],
    pub radius: f64,
    pub material: u32,
    pub object_id: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_camera {
    pub origin: [f64; 3],
    pub lower_left_corner: [f64; 3],
    pub horizontal: [f64; 3],
    pub vertical: [f64; 3],
    pub u: [f64; 3],
    pub v: [f64; 3],
    pub lens_radius: f64,
    pub time0: f64,
    pub time1: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_blob_header {
    pub magic: u32,
    pub version: u32,
    pub total_bytes: u64,
    pub n_nodes: u32,
    pub n_prims: u32,
    pub n_tris: u32,
    pub n_spheres: u32,
    pub n_moving_spheres: u32,
    pub n_rects: u32,
    pub n_xforms: u32,
    pub n_materials: u32,
    pub n_textures: u32,
    pub n_lights: u32,
    pub n_texel_bytes: u32,
    pub root_node: u32,
    pub root_end: u32,
    pub background: u32,
    pub off_nodes: u64,
    pub off_prims: u64,
    pub off_geom: u64,
    pub off_tri_attrs: u64,
    pub reserved0: u64,
    pub reserved1: u64,
    pub reserved2: u64,
    pub off_xforms: u64,
    pub off_materials: u64,
    pub off_textures: u64,
    pub off_texels: u64,
    pub off_lights: u64,
    pub camera: rt_camera,
    pub bvh_seed: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_render_params {
    pub width: u32,
    pub height: u32,
    pub spp_begin: u32,
    pub spp_end: u32,
    pub max_depth: i32,
    pub integrator: i32,
    pub seed: u64,
    pub samples_per_pass: u32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct rt_stats {
    pub rays: u64,
    pub paths: u64,
    pub box_tests: u64,
    pub tri_tests: u64,
    pub sphere_tests: u64,
    pub launches: u64,
    pub ms_total: f64,
    pub ms_trace: f64,
    pub ms_shade: f64,
    pub ms_other: f64,
    pub trace_launches: u64,
    pub trace_wave_steps: u64,
    pub filter_violations: u64,
    pub ms_trace_shadow: f64,
    pub trace_shadow_launches: u64,
    pub shadow_rays: u64,
    pub samples_per_pass: u32,
    pub lanes: u32,
    pub walk_run_steps: u64,
    pub walk_tri_steps: u64,
    pub walk_slow_steps: u64,
    pub walk_iterations: u64,
    pub ms_tail: f64,
    pub tail_bounces: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_pdf_desc {
    pub kind: u32,
    pub cannot_refract: u32,
    pub a: [f64; 3],
    pub b: [f64; 3],
    pub s: f64,
}

/// include/rt_checkpoint.h
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_checkpoint_header {
    pub magic: [u8; 8],
    pub width: u32,
    pub height: u32,
    pub spp_begin: u32,
    pub spp_done: u32,
    pub spp_end: u32,
    pub max_depth: i32,
    pub integrator: i32,
    pub rank: u32,
    pub world: u32,
    pub chunk: u32,
    pub seed: u64,
    pub scene_id: u64,
}

pub enum rt_context {}
pub enum rt_scene {}
pub enum rt_job {}

extern "C" {
    pub fn rt_device_count() -> c_int;
    pub fn rt_context_create(device: c_int, stream: *mut c_void, out: *mut *mut rt_context) -> c_int;
    pub fn rt_context_destroy(ctx: *mut rt_context) -> c_int;
    pub fn rt_context_synchronize(ctx: *mut rt_context) -> c_int;
    pub fn rt_scene_upload(ctx: *mut rt_context, blob: *const c_void, nbytes: u64, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut rt_scene) -> c_int;
    pub fn rt_stream_build_host(blob: *const c_void, nbytes: u64, stream_out: *mut c_void, capacity: u64, stream_bytes: *mut u64, root_off: *mut u32, node_off: *mut u32) -> c_int;
    pub fn rt_stream_build_prod_host(blob: *const c_void, nbytes: u64, stream_out: *mut c_void, capacity: u64, stream_bytes: *mut u64, root_off: *mut u32, node_off: *mut u32) -> c_int;
    pub fn rt_stream_nodes_prod_host(blob: *const c_void, nbytes: u64, nodes_out: *mut rt_node, capacity_nodes: u32, n_nodes_out: *mut u32, range_begin: *mut u32, range_end: *mut u32, elided_out: *mut u8) -> c_int;
    pub fn rt_prune_info_host(blob: *const c_void, nbytes: u64, info: *mut f64, n_rays: u32, rays: *const f64, certified: *mut u8) -> c_int;
    pub fn rt_trace_closest(scene: *mut rt_scene, n: u32, rays: *const f64, t_min: f64, t_max: f64, prim_id: *mut u32, inst_id: *mut u32, t: *mut f64, counters: *mut u64) -> c_int;
    pub fn rt_trace_closest_device(scene: *mut rt_scene, n: u32, rays_dev: *const f64, t_min: f64, t_max: f64, prim_id_dev: *mut u32, t_dev: *mut f64) -> c_int;
    pub fn rt_trace_closest_device_noalloc(scene: *mut rt_scene, n: u32, rays_dev: *const f64, t_min: f64, t_max: f64, prim_id_dev: *mut u32, t_dev: *mut f64) -> c_int;
    pub fn rt_primary_rays(scene: *mut rt_scene, width: u32, height: u32, sample: u32, seed: u64, rays: *mut f64) -> c_int;
    pub fn rt_render(scene: *mut rt_scene, params: *const rt_render_params, accum_dev: *mut f32, stats: *mut rt_stats) -> c_int;
    pub fn rt_render_begin(scene: *mut rt_scene, params: *const rt_render_params, accum_dev: *mut f32, job: *mut *mut rt_job) -> c_int;
    pub fn rt_render_end(job: *mut rt_job, stats: *mut rt_stats) -> c_int;
    pub fn rt_capture_rays(scene: *mut rt_scene, params: *const rt_render_params, bounce: u32, shadow: c_int, stride: u32, max_n: u32, n_out: *mut u32, queue_len: *mut u32,
                           rays: *mut f64, prim_id: *mut u32, inst_id: *mut u32, t: *mut f64, t_light: *mut f64) -> c_int;
    pub fn rt_accum_zero(ctx: *mut rt_context, accum_dev: *mut f32, width: u32, height: u32) -> c_int;
    pub fn rt_scene_set_pass_budget(scene: *mut rt_scene, bytes_per_lane: u64) -> c_int;
    pub fn rt_render_host(scene: *mut rt_scene, params: *const rt_render_params, accum_host: *mut f32, stats: *mut rt_stats) -> c_int;
    pub fn rt_resolve(ctx: *mut rt_context, accum_dev: *const f32, width: u32, height: u32, spp: u32, tonemap: c_int, out_dev: *mut f32) -> c_int;
    pub fn rt_resolve_host(accum_host: *const f32, width: u32, height: u32, spp: u32, tonemap: c_int, out_host: *mut f32) -> c_int;
    pub fn rt_sample_radiance(scene: *mut rt_scene, width: u32, height: u32, n: u32, pixel_index: *const u32, sample_index: *const u32, max_depth: i32, integrator: i32,
                              seed: u64, out_rgb: *mut f64) -> c_int;
    pub fn rt_sample_radiance_sig(scene: *mut rt_scene, width: u32, height: u32, n: u32, pixel_index: *const u32, sample_index: *const u32, max_depth: i32, integrator: i32,
                                  seed: u64, out_rgb: *mut f64, out_sig: *mut u64) -> c_int;
    pub fn rt_pdf_eval(ctx: *mut rt_context, n: u32, n_comp: u32, pdfs: *const rt_pdf_desc, dirs: *const f64, seed: u64, value: *mut f64, generated: *mut f64) -> c_int;
    pub fn rt_ieee_div_sqrt(ctx: *mut rt_context, n: u32, a: *const f64, b: *const f64, q: *mut f64, s: *mut f64) -> c_int;
    pub fn rt_last_error() -> *const c_char;
}
