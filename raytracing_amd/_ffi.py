"""ctypes bindings for librt_host.so (include/rt_host.h) and librt_hip.so (include/rt_abi.h).

Plumbing only: every computation happens in the two shared libraries.  librt_hip.so is loaded
lazily and loudly -- there is no CPU fallback for the traced/shaded path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(_HERE, "host", "librt_host.so")
HIP_LIB = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "csrc", "librt_hip.so")  # RT_HIP_LIB: diagnostic builds only
ASSETS = os.path.join(os.path.dirname(_HERE), "assets")

u32, u64, i32, f64, f32 = C.c_uint32, C.c_uint64, C.c_int32, C.c_double, C.c_float
P = C.POINTER


class RenderParams(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("spp_begin", u32), ("spp_end", u32), ("max_depth", i32),
                ("integrator", i32), ("seed", u64), ("samples_per_pass", u32), ("flags", u32)]


class Stats(C.Structure):
    _fields_ = [("rays", u64), ("paths", u64), ("box_tests", u64), ("tri_tests", u64), ("sphere_tests", u64),
                ("launches", u64), ("ms_total", f64), ("ms_trace", f64), ("ms_shade", f64), ("ms_other", f64),
                ("trace_launches", u64), ("trace_wave_steps", u64), ("filter_violations", u64),
                ("ms_trace_shadow", f64), ("trace_shadow_launches", u64), ("shadow_rays", u64), ("samples_per_pass", u32), ("lanes", u32),
                ("walk_run_steps", u64), ("walk_tri_steps", u64), ("walk_slow_steps", u64), ("walk_iterations", u64),
                ("ms_tail", f64), ("tail_bounces", u64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Node(C.Structure):
    _fields_ = [("bmin", f64 * 3), ("bmax", f64 * 3), ("skip", u32), ("flags", u32), ("prim0", u32), ("prim1", u32)]


class Prim(C.Structure):
    _fields_ = [("kind", u32), ("index", u32), ("xform", u32), ("material", u32), ("object_id", u32), ("inst_id", u32),
                ("sub_end", u32), ("pad", u32)]


class Camera(C.Structure):
    _fields_ = [("origin", f64 * 3), ("lower_left_corner", f64 * 3), ("horizontal", f64 * 3), ("vertical", f64 * 3),
                ("u", f64 * 3), ("v", f64 * 3), ("lens_radius", f64), ("time0", f64), ("time1", f64)]


class BlobHeader(C.Structure):
    _fields_ = [("magic", u32), ("version", u32), ("total_bytes", u64),
                ("n_nodes", u32), ("n_prims", u32), ("n_tris", u32), ("n_spheres", u32), ("n_moving_spheres", u32),
                ("n_rects", u32), ("n_xforms", u32), ("n_materials", u32), ("n_textures", u32), ("n_lights", u32),
                ("n_texel_bytes", u32), ("root_node", u32), ("root_end", u32), ("background", u32),
                ("off_nodes", u64), ("off_prims", u64), ("off_geom", u64), ("off_tri_attrs", u64),
                ("reserved0", u64), ("reserved1", u64), ("reserved2", u64), ("off_xforms", u64),
                ("off_materials", u64), ("off_textures", u64), ("off_texels", u64), ("off_lights", u64),
                ("camera", Camera), ("bvh_seed", u64)]


_host = None
_hip = None


def host():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(HOST_LIB)
        L.rth_last_error.restype = C.c_char_p
        L.rth_set_bvh_axis_seed.argtypes = [u64]
        L.rth_set_bvh_axis_mode.argtypes = [u32]; L.rth_set_bvh_axis_mode.restype = C.c_int
        L.rth_scene_builtin.restype = C.c_void_p
        L.rth_scene_builtin.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.rth_scene_from_description.restype = C.c_void_p
        L.rth_scene_from_description.argtypes = [C.c_char_p, u64]
        L.rth_scene_free.argtypes = [C.c_void_p]
        L.rth_scene_describe.restype = C.c_void_p
        L.rth_scene_describe.argtypes = [C.c_void_p, P(u64)]
        L.rth_scene_flatten.restype = C.c_void_p
        L.rth_scene_flatten.argtypes = [C.c_void_p, P(u64)]
        L.rth_scene_object_count.restype = u32
        L.rth_scene_object_count.argtypes = [C.c_void_p]
        L.rth_write_ppm.argtypes = [C.c_char_p, C.c_void_p, u32, u32]
        L.rth_write_pfm.argtypes = [C.c_char_p, C.c_void_p, u32, u32]
        L.rth_write_png.argtypes = [C.c_char_p, C.c_void_p, u32, u32]
        L.rth_image_get_tile.argtypes = [u32] * 6 + [P(u32)]
        L.rth_tile_get_xy.argtypes = [u32] * 5 + [P(u32)]
        L.rth_image_merge_tile.argtypes = [C.c_void_p, u32, u32, u32, u32, u32, u32, C.c_void_p]
        L.rth_mat4_inverse.argtypes = [C.c_void_p, C.c_void_p]
        L.rth_sphere_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rth_pcg64_stream.argtypes = [u64, u32, C.c_void_p]
        _host = L
    return _host


def hip():
    """librt_hip.so -- raises if the HIP extension is absent (no fallback path exists)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(f"{HIP_LIB} missing: the GPU path has no CPU fallback; build it with __graft_entry__.build()")
        L = C.CDLL(HIP_LIB)
        L.rt_last_error.restype = C.c_char_p
        L.rt_device_count.restype = C.c_int
        L.rt_context_create.argtypes = [C.c_int, C.c_void_p, P(C.c_void_p)]
        L.rt_context_destroy.argtypes = [C.c_void_p]
        L.rt_context_synchronize.argtypes = [C.c_void_p]
        L.rt_scene_upload.argtypes = [C.c_void_p, C.c_void_p, u64, P(C.c_void_p)]
        L.rt_scene_destroy.argtypes = [C.c_void_p]
        L.rt_stream_build_host.argtypes = [C.c_void_p, u64, C.c_void_p, u64, P(u64), P(u32), C.c_void_p]
        L.rt_stream_build_prod_host.argtypes = [C.c_void_p, u64, C.c_void_p, u64, P(u64), P(u32), C.c_void_p]
        L.rt_stream_nodes_prod_host.argtypes = [C.c_void_p, u64, C.c_void_p, u32, P(u32), P(u32), P(u32), C.c_void_p]
        L.rt_prune_info_host.argtypes = [C.c_void_p, u64, C.c_void_p, u32, C.c_void_p, C.c_void_p]
        L.rt_trace_closest.argtypes = [C.c_void_p, u32, C.c_void_p, f64, f64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_trace_closest_device.argtypes = [C.c_void_p, u32, C.c_void_p, f64, f64, C.c_void_p, C.c_void_p]
        L.rt_trace_closest_device_noalloc.argtypes = [C.c_void_p, u32, C.c_void_p, f64, f64, C.c_void_p, C.c_void_p]
        L.rt_primary_rays.argtypes = [C.c_void_p, u32, u32, u32, u64, C.c_void_p]
        L.rt_render.argtypes = [C.c_void_p, P(RenderParams), C.c_void_p, P(Stats)]
        L.rt_render_begin.argtypes = [C.c_void_p, P(RenderParams), C.c_void_p, P(C.c_void_p)]
        L.rt_render_end.argtypes = [C.c_void_p, P(Stats)]
        L.rt_capture_rays.argtypes = [C.c_void_p, P(RenderParams), u32, C.c_int, u32, u32, P(u32), P(u32), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_accum_zero.argtypes = [C.c_void_p, C.c_void_p, u32, u32]
        L.rt_scene_set_pass_budget.argtypes = [C.c_void_p, u64]
        L.rt_render_host.argtypes = [C.c_void_p, P(RenderParams), C.c_void_p, P(Stats)]
        L.rt_resolve.argtypes = [C.c_void_p, C.c_void_p, u32, u32, u32, C.c_int, C.c_void_p]
        L.rt_resolve_host.argtypes = [C.c_void_p, u32, u32, u32, C.c_int, C.c_void_p]
        L.rt_sample_radiance.argtypes = [C.c_void_p, u32, u32, u32, C.c_void_p, C.c_void_p, i32, i32, u64, C.c_void_p]
        L.rt_sample_radiance_sig.argtypes = [C.c_void_p, u32, u32, u32, C.c_void_p, C.c_void_p, i32, i32, u64, C.c_void_p, C.c_void_p]
        L.rt_pdf_eval.argtypes = [C.c_void_p, u32, u32, C.c_void_p, C.c_void_p, u64, C.c_void_p, C.c_void_p]
        L.rt_ieee_div_sqrt.argtypes = [C.c_void_p, u32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _hip = L
    return _hip


ABI_SYMBOLS = ["rt_device_count", "rt_context_create", "rt_context_destroy", "rt_context_synchronize", "rt_scene_upload",
               "rt_scene_destroy", "rt_stream_build_host", "rt_stream_build_prod_host", "rt_stream_nodes_prod_host", "rt_prune_info_host", "rt_trace_closest", "rt_trace_closest_device", "rt_trace_closest_device_noalloc", "rt_primary_rays", "rt_render", "rt_render_begin", "rt_render_end", "rt_capture_rays",
               "rt_accum_zero", "rt_scene_set_pass_budget", "rt_render_host", "rt_resolve", "rt_resolve_host", "rt_sample_radiance", "rt_sample_radiance_sig",
               "rt_pdf_eval", "rt_ieee_div_sqrt", "rt_last_error"]
HOST_SYMBOLS = ["rth_set_bvh_axis_seed", "rth_set_bvh_axis_mode", "rth_scene_builtin", "rth_scene_from_description", "rth_scene_free",
                "rth_scene_describe", "rth_scene_flatten", "rth_scene_object_count", "rth_write_ppm", "rth_write_pfm", "rth_write_png",
                "rth_image_get_tile", "rth_tile_get_xy", "rth_image_merge_tile", "rth_mat4_inverse",
                "rth_sphere_transform", "rth_pcg64_stream", "rth_last_error"]
