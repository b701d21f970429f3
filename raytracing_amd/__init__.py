"""raytracing_amd -- MI355X-native `ray_colour` path.

Thin Python face over the two C libraries:
  librt_host.so  host-side scene construction, BVH build (bvh.rs:36-87) and flattening
  librt_hip.so   gfx950 kernels behind the C ABI of include/rt_abi.h
Nothing here computes: numpy only carries buffers.  Missing librt_hip.so raises.
"""
import ctypes as C
import weakref

import numpy as np

from . import _ffi
from ._ffi import ASSETS, RenderParams, Stats

INT_MIS, INT_LIGHT_IS, INT_BRDF, INT_UNIFORM = 0, 1, 2, 3
INTEGRATORS = {"mis": INT_MIS, "light_is": INT_LIGHT_IS, "brdf": INT_BRDF, "uniform": INT_UNIFORM}
NONE = 0xFFFFFFFF
RENDER_COUNT_TESTS, RENDER_NO_SORT, RENDER_ONE_LANE, RENDER_LANE_WALK, RENDER_MATERIAL_BINS, RENDER_COUNT_STEPS, RENDER_SERIAL_SHADOW = 1, 2, 4, 8, 16, 32, 64
BVH_AXIS_MODES = {"seeded": 0, "min_area": 1}   # include/rt_host.h RTH_BVH_AXIS_*
NOALLOC_MAX_RAYS = 262144
PDF_COSINE, PDF_UNIFORM_HEMI, PDF_UNIFORM_SPHERE, PDF_FUZZY, PDF_DIELECTRIC, PDF_CONE = range(6)
PDF_DESC = np.dtype([("kind", "<u4"), ("cannot_refract", "<u4"), ("a", "<f8", 3), ("b", "<f8", 3), ("s", "<f8")])   # rt_pdf_desc, 64 B


class RtError(RuntimeError):
    pass


def _host_err():
    return RtError("librt_host: " + _ffi.host().rth_last_error().decode())


def _hip_check(rc):
    if rc != 0:
        raise RtError(f"librt_hip rc={rc}: " + _ffi.hip().rt_last_error().decode())


class HostScene:
    """A scene graph held by librt_host.so (renderer::Scene)."""

    def __init__(self, handle):
        if not handle:
            raise _host_err()
        self._h = handle

    @classmethod
    def builtin(cls, name, bvh_seed=0, legacy_bvh=False, assets=ASSETS, bvh_axis="seeded"):
        """bvh_axis: "seeded" -- split axes from the splitmix64 stream of `bvh_seed` (stands in for bvh.rs:53's unseeded thread_rng);
        "min_area" -- at every node the axis, of the three the reference can draw, whose median split has the least area x count
        (include/rt_host.h): one particular tree of the reference's distribution, 25 % faster to walk on the pagoda."""
        L = _ffi.host()
        L.rth_set_bvh_axis_seed(bvh_seed)
        if L.rth_set_bvh_axis_mode(BVH_AXIS_MODES[bvh_axis]) != 0:
            raise ValueError("bvh_axis")
        try:
            return cls(L.rth_scene_builtin(name.encode(), assets.encode(), 1 if legacy_bvh else 0))
        finally:
            L.rth_set_bvh_axis_mode(0)

    @classmethod
    def from_description(cls, text):
        b = text.encode() if isinstance(text, str) else text
        return cls(_ffi.host().rth_scene_from_description(b, len(b)))

    def describe(self):
        n = _ffi.u64()
        p = _ffi.host().rth_scene_describe(self._h, C.byref(n))
        if not p:
            raise _host_err()
        return C.string_at(p, n.value)

    def blob(self):
        n = _ffi.u64()
        p = _ffi.host().rth_scene_flatten(self._h, C.byref(n))
        if not p:
            raise _host_err()
        return C.string_at(p, n.value)

    def header(self):
        return _ffi.BlobHeader.from_buffer_copy(self.blob()[:C.sizeof(_ffi.BlobHeader)])

    def nodes(self):
        b, h = self.blob(), self.header()
        return np.frombuffer(b, dtype=np.dtype([("bmin", "<f8", 3), ("bmax", "<f8", 3), ("skip", "<u4"), ("flags", "<u4"),
                                                ("prim0", "<u4"), ("prim1", "<u4")]), count=h.n_nodes, offset=h.off_nodes)

    def prims(self):
        b, h = self.blob(), self.header()
        return np.frombuffer(b, dtype=np.dtype([("kind", "<u4"), ("index", "<u4"), ("xform", "<u4"), ("material", "<u4"),
                                                ("object_id", "<u4"), ("inst_id", "<u4"), ("sub_end", "<u4"), ("pad", "<u4")]),
                             count=h.n_prims, offset=h.off_prims)

    def geoms(self):
        b, h = self.blob(), self.header()
        return np.frombuffer(b, dtype=np.dtype([("d", "<f8", 9), ("kind", "<u4"), ("xform", "<u4")]), count=h.n_prims, offset=h.off_geom)

    def trace_stream(self, production=False):
        """The device trace stream rt_scene_upload derives from this scene, built on the host (no GPU needed):
        (stream bytes as uint8 array incl. padding, offset of the root record, record offset per node index).
        production=True: the form the pruning kernel walks (always-entered inner nodes have no record)."""
        b, h = self.blob(), self.header()
        fn = _ffi.hip().rt_stream_build_prod_host if production else _ffi.hip().rt_stream_build_host
        n, root = _ffi.u64(), _ffi.u32()
        _hip_check(fn(b, len(b), None, 0, C.byref(n), C.byref(root), None))
        out = np.zeros(n.value, np.uint8); off = np.zeros(h.n_nodes + 1, np.uint32)
        _hip_check(fn(b, len(b), out.ctypes.data, out.nbytes, C.byref(n), C.byref(root), off.ctypes.data))
        return out, int(root.value), off

    def production_nodes(self):
        """The node array the production stream's records index (rt_stream_nodes_prod_host): (nodes, range_begin, range_end, elided) -- the blob's
        nodes followed, when the scene-level hierarchy was rebuilt, by the new scene-level range; elided[i] = 1: node i has no record."""
        b = self.blob()
        n, r0, r1 = _ffi.u32(), _ffi.u32(), _ffi.u32()
        _hip_check(_ffi.hip().rt_stream_nodes_prod_host(b, len(b), None, 0, C.byref(n), C.byref(r0), C.byref(r1), None))
        nodes = np.zeros(n.value, self.nodes().dtype); gone = np.zeros(n.value, np.uint8)
        _hip_check(_ffi.hip().rt_stream_nodes_prod_host(b, len(b), nodes.ctypes.data, n.value, C.byref(n), C.byref(r0), C.byref(r1), gone.ctypes.data))
        return nodes, int(r0.value), int(r1.value), gone

    def prune_info(self, rays=None):
        """Strict-pruning set-up of this scene as rt_scene_upload derives it (host only).  rays: n x 6 (origin, direction) in the
        pruned space -> also returns the device's verdict per ray: 1 certified, 2 certified with protected suspect leaves, 0 walked without pruning."""
        b = self.blob()
        info = np.zeros(12)
        n = 0 if rays is None else len(rays)
        r = np.ascontiguousarray(rays, np.float64) if n else np.zeros((1, 6))
        cert = np.zeros(max(n, 1), np.uint8)
        _hip_check(_ffi.hip().rt_prune_info_host(b, len(b), info.ctypes.data, n, r.ctypes.data, cert.ctypes.data))
        keys = ("on", "xform", "triangles", "prunable_nodes", "normals", "grid_entries", "shape_min", "tau_coef", "delta_coef", "radius", "box_abs_max", "elided_nodes")
        d = dict(zip(keys, info))
        return (d, cert[:n].copy()) if rays is not None else d

    def __del__(self):
        try:
            if self._h:
                _ffi.host().rth_scene_free(self._h)
                self._h = None
        except Exception:
            pass


class Context:
    """rt_context: one per GPU / process."""

    def __init__(self, device=0, stream=None):
        L = _ffi.hip()
        h = C.c_void_p()
        _hip_check(L.rt_context_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self.device = device
        self._scenes = []   # weak references to live GpuScene objects: closed before the context

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, scene):
        blob = scene.blob() if isinstance(scene, HostScene) else scene
        return GpuScene(self, blob)

    def synchronize(self):
        _hip_check(_ffi.hip().rt_context_synchronize(self._h))

    def accum_zero(self, accum_ptr, width, height):
        _hip_check(_ffi.hip().rt_accum_zero(self._h, C.c_void_p(accum_ptr), width, height))

    def resolve(self, accum_ptr, width, height, spp, tonemap, out_ptr):
        _hip_check(_ffi.hip().rt_resolve(self._h, C.c_void_p(accum_ptr), width, height, spp, int(tonemap), C.c_void_p(out_ptr)))

    def pdf_eval(self, descs, dirs, seed=0x5EED):
        """PDF::value / PDF::generate for descs (n x n_comp structured array of PDF_DESC; n_comp > 1 = MixturePDF Uniform) and dirs (n x 3)."""
        descs = np.ascontiguousarray(descs, PDF_DESC)
        descs = descs.reshape(len(descs), -1)
        n, n_comp = descs.shape
        dirs = np.ascontiguousarray(dirs, np.float64).reshape(n, 3)
        value = np.empty(n); gen = np.empty((n, 3))
        _hip_check(_ffi.hip().rt_pdf_eval(self._h, n, n_comp, descs.ctypes.data, dirs.ctypes.data, seed, value.ctypes.data, gen.ctypes.data))
        return value, gen

    def ieee_div_sqrt(self, a, b):
        a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
        q = np.empty_like(a); s = np.empty_like(a)
        _hip_check(_ffi.hip().rt_ieee_div_sqrt(self._h, a.size, a.ctypes.data, b.ctypes.data, q.ctypes.data, s.ctypes.data))
        return q, s

    def close(self):
        if self._h:
            for ref in list(self._scenes):   # a scene keeps a pointer to its context: release the scenes first
                sc = ref()
                if sc is not None:
                    sc.close()
            self._scenes.clear()
            _hip_check(_ffi.hip().rt_context_destroy(self._h))
            self._h = None


class GpuScene:
    """rt_scene: the flattened blob resident in HBM."""

    def __init__(self, ctx, blob):
        h = C.c_void_p()
        _hip_check(_ffi.hip().rt_scene_upload(ctx._h, blob, len(blob), C.byref(h)))
        self._h = h
        self.ctx = ctx   # keeps the context alive as long as the scene
        ctx._scenes.append(weakref.ref(self))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):   # a dropped scene would otherwise hold its path buffers (up to two thirds of HBM)
        try:
            self.close()
        except Exception:
            pass

    def trace_closest(self, rays, t_min, t_max, counters=False):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 7)
        n = rays.shape[0]
        prim = np.empty(n, np.uint32); inst = np.empty(n, np.uint32); t = np.empty(n, np.float64)
        cnt = np.zeros(8, np.uint64)  # box, triangle, sphere tests; filter violations (must be 0)
        _hip_check(_ffi.hip().rt_trace_closest(self._h, n, rays.ctypes.data, t_min, t_max, prim.ctypes.data, inst.ctypes.data,
                                               t.ctypes.data, cnt.ctypes.data if counters else None))
        return (prim, inst, t, cnt) if counters else (prim, inst, t)

    def trace_closest_device(self, n, rays_ptr, t_min, t_max, prim_ptr, t_ptr):
        _hip_check(_ffi.hip().rt_trace_closest_device(self._h, n, C.c_void_p(rays_ptr), t_min, t_max, C.c_void_p(prim_ptr), C.c_void_p(t_ptr)))

    def trace_closest_device_noalloc(self, n, rays_ptr, t_min, t_max, prim_ptr, t_ptr):
        """rt_trace_closest_device without any allocation (fixed scratch inside the library, n <= NOALLOC_MAX_RAYS)."""
        _hip_check(_ffi.hip().rt_trace_closest_device_noalloc(self._h, n, C.c_void_p(rays_ptr), t_min, t_max, C.c_void_p(prim_ptr), C.c_void_p(t_ptr)))

    def primary_rays(self, width, height, sample, seed):
        rays = np.empty((width * height, 7), np.float64)
        _hip_check(_ffi.hip().rt_primary_rays(self._h, width, height, sample, seed, rays.ctypes.data))
        return rays

    def _params(self, width, height, spp_begin, spp_end, max_depth, integrator, seed, samples_per_pass, flags):
        return RenderParams(width, height, spp_begin, spp_end, max_depth, integrator, seed, samples_per_pass, flags)

    def render(self, accum_ptr, width, height, spp_begin, spp_end, max_depth, integrator, seed=0x5EED, samples_per_pass=0, flags=0):
        """Adds radiance sums of samples [spp_begin, spp_end) into the DEVICE fp32 accumulator at accum_ptr."""
        p = self._params(width, height, spp_begin, spp_end, max_depth, integrator, seed, samples_per_pass, flags)
        st = Stats()
        _hip_check(_ffi.hip().rt_render(self._h, C.byref(p), C.c_void_p(accum_ptr), C.byref(st)))
        return st

    def set_pass_budget(self, bytes_per_lane):
        """Bytes of path state per lane that samples_per_pass=0 may use (0 = default, 80 GiB)."""
        _hip_check(_ffi.hip().rt_scene_set_pass_budget(self._h, int(bytes_per_lane)))

    def render_begin(self, accum_ptr, width, height, spp_begin, spp_end, max_depth, integrator, seed=0x5EED, samples_per_pass=0, flags=0):
        """rt_render_begin: queues the call (at most two in flight per scene) and returns a job handle for render_end."""
        p = self._params(width, height, spp_begin, spp_end, max_depth, integrator, seed, samples_per_pass, flags)
        job = C.c_void_p()
        _hip_check(_ffi.hip().rt_render_begin(self._h, C.byref(p), C.c_void_p(accum_ptr), C.byref(job)))
        return job

    def render_end(self, job):
        st = Stats()
        _hip_check(_ffi.hip().rt_render_end(job, C.byref(st)))
        return st

    def capture_rays(self, width, height, sample, max_depth, integrator, bounce, shadow, stride, max_n, seed=0x5EED, flags=0):
        """Every stride-th ray of the render loop's queue at `bounce` (shadow rays if shadow) for one sample of every pixel, with the
        loop's own traversal result: (rays n x 7, prim, inst, t, t_light, queue length)."""
        p = self._params(width, height, sample, sample + 1, max_depth, integrator, seed, 0, flags)
        rays = np.empty((max_n, 7)); prim = np.empty(max_n, np.uint32); inst = np.empty(max_n, np.uint32); t = np.empty(max_n); tl = np.empty(max_n)
        n, qlen = _ffi.u32(), _ffi.u32()
        _hip_check(_ffi.hip().rt_capture_rays(self._h, C.byref(p), bounce, int(bool(shadow)), stride, max_n, C.byref(n), C.byref(qlen), rays.ctypes.data,
                                              prim.ctypes.data, inst.ctypes.data, t.ctypes.data, tl.ctypes.data))
        k = n.value
        return rays[:k].copy(), prim[:k].copy(), inst[:k].copy(), t[:k].copy(), tl[:k].copy(), qlen.value

    def render_host(self, width, height, spp_begin, spp_end, max_depth, integrator, seed=0x5EED, samples_per_pass=0, flags=0):
        p = self._params(width, height, spp_begin, spp_end, max_depth, integrator, seed, samples_per_pass, flags)
        st = Stats()
        acc = np.zeros((height, width, 3), np.float32)
        _hip_check(_ffi.hip().rt_render_host(self._h, C.byref(p), acc.ctypes.data, C.byref(st)))
        return acc, st

    def sample_radiance(self, width, height, pixel_index, sample_index, max_depth, integrator, seed=0x5EED):
        px = np.ascontiguousarray(pixel_index, np.uint32); sm = np.ascontiguousarray(sample_index, np.uint32)
        out = np.empty((px.size, 3), np.float64)
        _hip_check(_ffi.hip().rt_sample_radiance(self._h, width, height, px.size, px.ctypes.data, sm.ctypes.data, max_depth, integrator,
                                                 seed, out.ctypes.data))
        return out

    def sample_radiance_sig(self, width, height, pixel_index, sample_index, max_depth, integrator, seed=0x5EED):
        """sample_radiance plus each path's signature (u64): the hash of its discrete decisions (include/rt_abi.h)."""
        px = np.ascontiguousarray(pixel_index, np.uint32); sm = np.ascontiguousarray(sample_index, np.uint32)
        out = np.empty((px.size, 3), np.float64); sig = np.empty(px.size, np.uint64)
        _hip_check(_ffi.hip().rt_sample_radiance_sig(self._h, width, height, px.size, px.ctypes.data, sm.ctypes.data, max_depth, integrator,
                                                     seed, out.ctypes.data, sig.ctypes.data))
        return out, sig

    def close(self):
        if self._h:
            _ffi.hip().rt_scene_destroy(self._h)
            self._h = None


def resolve_host(accum, spp, tonemap=True):
    accum = np.ascontiguousarray(accum, np.float32)
    h, w = accum.shape[:2]
    out = np.empty_like(accum)
    _hip_check(_ffi.hip().rt_resolve_host(accum.ctypes.data, w, h, spp, int(tonemap), out.ctypes.data))
    return out


def write_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.float64)
    h, w = rgb.shape[:2]
    if _ffi.host().rth_write_ppm(path.encode(), rgb.ctypes.data, w, h) != 0:
        raise _host_err()


def write_pfm(path, rgb):
    """Linear float32 dump of a resolved frame (rows bottom-up, little endian): for diffing; no counterpart in the reference."""
    rgb = np.ascontiguousarray(rgb, np.float64)
    h, w = rgb.shape[:2]
    if _ffi.host().rth_write_pfm(path.encode(), rgb.ctypes.data, w, h) != 0:
        raise _host_err()


def write_png(path, rgb):
    """8-bit RGB PNG holding exactly the values write_ppm prints (sqrt gamma, clamp .999, *256 truncation)."""
    rgb = np.ascontiguousarray(rgb, np.float64)
    h, w = rgb.shape[:2]
    if _ffi.host().rth_write_png(path.encode(), rgb.ctypes.data, w, h) != 0:
        raise _host_err()
