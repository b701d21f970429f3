//! GPU path for CampbellOwen/RayTracing's `renderer` crate on MI355X: the `extern "C"` bindings of librt_hip.so
//! (include/rt_abi.h), a flattener from the crate's scene graph to the RTSB blob, and `GpuRenderer`, the drop-in for the
//! tile / pixel / spp loop of desktop/src/main.rs:978-1019.
//!
//! In `desktop/src/main.rs` the loop over tiles becomes
//! ```ignore
//! let gpu = renderer_gpu::GpuRenderer::new(0, &scene)?;                       // upload once
//! let img = gpu.render(renderer_gpu::abi::RT_INT_MIS, image_width, image_height, samples_per_pixel, max_depth, 24301)?;
//! write_image(&img, "output.ppm");                                             // exporters/ppm.rs, unchanged
//! ```
pub mod abi;
pub mod bvh_axis;
pub mod flatten;

use abi::*;
use glam::DVec3;
use std::ffi::CStr;
use std::os::raw::c_void;

#[derive(Debug)]
pub enum GpuError {
    Flatten(flatten::FlattenError),
    /// negative class of include/rt_abi.h and the library's message
    Rt(i32, String),
}
fn check(rc: i32) -> Result<(), GpuError> {
    if rc == RT_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned();
    Err(GpuError::Rt(rc, msg))
}

/// One context (one GPU, one host thread) and one uploaded scene.  Not `Sync`: a context is not thread-safe; use one per GPU.
pub struct GpuRenderer {
    ctx: *mut rt_context,
    scene: *mut rt_scene,
}

impl GpuRenderer {
    pub fn new(device: i32, scene: &renderer::Scene) -> Result<Self, GpuError> {
        let blob = flatten::flatten(scene).map_err(GpuError::Flatten)?;
        Self::from_blob(device, &blob)
    }
    pub fn from_blob(device: i32, blob: &[u8]) -> Result<Self, GpuError> {
        let mut ctx = std::ptr::null_mut();
        let mut gs = std::ptr::null_mut();
        unsafe {
            check(rt_context_create(device, std::ptr::null_mut(), &mut ctx))?;
            if let Err(e) = check(rt_scene_upload(ctx, blob.as_ptr() as *const c_void, blob.len() as u64, &mut gs)) {
                rt_context_destroy(ctx);
                return Err(e);
            }
        }
        Ok(GpuRenderer { ctx, scene: gs })
    }

    /// Bytes of path state per render lane that `render` may use (default 80 GiB per lane, two lanes).  On a GPU that other work
    /// occupies in part the library steps the budget down by itself when an allocation fails; a caller that wants resumed or
    /// sharded renders to cut the same passes sets it once, before the first render.
    pub fn set_pass_budget(&self, bytes_per_lane: u64) -> Result<(), GpuError> {
        unsafe { check(rt_scene_set_pass_budget(self.scene, bytes_per_lane)) }
    }

    /// Radiance SUMS of the absolute sample indices [spp_begin, spp_end) for every pixel (row 0 = bottom row), fp32.
    /// Ranks of a multi-GPU render call this with disjoint ranges and add the results before `resolve`.
    pub fn render_sums(&self, integrator: i32, width: u32, height: u32, spp_begin: u32, spp_end: u32, max_depth: i32, seed: u64) -> Result<(Vec<f32>, rt_stats), GpuError> {
        let p = rt_render_params { width, height, spp_begin, spp_end, max_depth, integrator, seed, samples_per_pass: 0, flags: 0 };
        let mut acc = vec![0f32; (width as usize) * (height as usize) * 3];
        let mut st = rt_stats::default();
        unsafe { check(rt_render_host(self.scene, &p, acc.as_mut_ptr(), &mut st))? };
        Ok((acc, st))
    }

    /// mean -> aces_tonemapping (main.rs:913-921) -> NaN scrub (image.rs:48-56)
    pub fn resolve(sums: &[f32], width: u32, height: u32, spp: u32, tonemap: bool) -> Result<Vec<f32>, GpuError> {
        let mut px = vec![0f32; sums.len()];
        unsafe { check(rt_resolve_host(sums.as_ptr(), width, height, spp, tonemap as i32, px.as_mut_ptr()))? };
        Ok(px)
    }

    /// Drop-in for desktop/src/main.rs:978-1019: the finished `Image`, ready for `write_image`.
    pub fn render(&self, integrator: i32, width: u32, height: u32, spp: u32, max_depth: i32, seed: u64) -> Result<renderer::Image, GpuError> {
        let (sums, _) = self.render_sums(integrator, width, height, 0, spp, max_depth, seed)?;
        let px = Self::resolve(&sums, width, height, spp, true)?;
        let mut img = renderer::Image::new((width, height));
        for (i, c) in px.chunks(3).enumerate() {
            img.data[i] = DVec3::new(c[0] as f64, c[1] as f64, c[2] as f64); // row 0 = bottom row, as write_image expects (ppm.rs:15)
        }
        Ok(img)
    }

    /// `scene.hit(&ray, t_min, t_max)` (scene.rs:31-38) for many rays: (object id or RT_NONE, t) per ray.
    pub fn trace_closest(&self, rays: &[[f64; 7]], t_min: f64, t_max: f64) -> Result<(Vec<u32>, Vec<f64>), GpuError> {
        let n = rays.len();
        let (mut prim, mut t) = (vec![RT_NONE; n], vec![0f64; n]);
        unsafe {
            check(rt_trace_closest(self.scene, n as u32, rays.as_ptr() as *const f64, t_min, t_max, prim.as_mut_ptr(), std::ptr::null_mut(), t.as_mut_ptr(), std::ptr::null_mut()))?
        };
        Ok((prim, t))
    }
}

impl Drop for GpuRenderer {
    fn drop(&mut self) {
        unsafe {
            rt_scene_destroy(self.scene); // scenes first: rt_context_destroy refuses while scenes are alive
            rt_context_destroy(self.ctx);
        }
    }
}
