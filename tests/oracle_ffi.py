"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE.  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by raytracing_amd/."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")
u32, u64, f64 = C.c_uint32, C.c_uint64, C.c_double
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} missing: make -C oracle")
        L = C.CDLL(LIB)
        L.orc_last_error.restype = C.c_char_p
        L.orc_scene_load.restype = C.c_void_p
        L.orc_scene_load.argtypes = [C.c_char_p, u64]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_trace_closest.argtypes = [C.c_void_p, u32, C.c_void_p, f64, f64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hit_record.argtypes = [C.c_void_p, C.c_void_p, f64, f64, C.c_void_p]
        L.orc_primary_rays.argtypes = [C.c_void_p, u32, u32, u32, u64, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, u32, u32, u32, u32, C.c_int, C.c_int, u64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_sample_radiance.argtypes = [C.c_void_p, u32, u32, u32, u32, u32, C.c_int, C.c_int, u64, C.c_void_p]
        L.orc_sample_radiance_sig.argtypes = [C.c_void_p, u32, u32, u32, u32, u32, C.c_int, C.c_int, u64, C.c_void_p, C.c_void_p]
        L.orc_resolve.argtypes = [C.c_void_p, u32, u32, u32, C.c_int, C.c_void_p]
        L.orc_quantize_ppm.argtypes = [C.c_void_p, u32, u32, C.c_void_p]
        L.orc_bvh_dump.restype = C.c_int64
        L.orc_bvh_dump.argtypes = [C.c_void_p, u32, C.c_void_p, C.c_void_p]
        L.orc_object_bbox.argtypes = [C.c_void_p, u32, f64, f64, C.c_void_p]
        L.orc_list_bbox.argtypes = [C.c_void_p, u32, C.c_void_p, f64, f64, C.c_void_p]
        L.orc_object_hit.argtypes = [C.c_void_p, u32, C.c_void_p, f64, f64, C.c_void_p]
        L.orc_aabb_hit.argtypes = [C.c_void_p, C.c_void_p, f64, f64]
        L.orc_philox.argtypes = [C.c_void_p, u32, u32]
        L.orc_rng_stream.argtypes = [u64, u32, u32, u32, u32, C.c_void_p]
        L.orc_get_tile.argtypes = [u32] * 6 + [C.c_void_p]
        L.orc_tile_xy.argtypes = [u32] * 5 + [C.c_void_p]
        L.orc_mat4_inverse.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_pdf_eval.argtypes = [u32, u32, C.c_void_p, C.c_void_p, u64, u32, C.c_void_p, C.c_void_p]
        L.orc_div_sqrt.argtypes = [u32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


class OracleScene:
    def __init__(self, desc):
        b = desc.encode() if isinstance(desc, str) else desc
        self._h = lib().orc_scene_load(b, len(b))
        if not self._h:
            raise OracleError(lib().orc_last_error().decode())

    def _chk(self, rc):
        if rc < 0:
            raise OracleError(lib().orc_last_error().decode())
        return rc

    def trace_closest(self, rays, t_min, t_max):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 7)
        n = rays.shape[0]
        prim = np.empty(n, np.uint32); inst = np.empty(n, np.uint32); t = np.empty(n, np.float64); cnt = np.zeros(4, np.uint64)
        self._chk(lib().orc_trace_closest(self._h, n, rays.ctypes.data, t_min, t_max, prim.ctypes.data, inst.ctypes.data, t.ctypes.data, cnt.ctypes.data))
        return prim, inst, t, cnt

    def hit_record(self, ray, t_min, t_max):
        r = np.ascontiguousarray(ray, np.float64); out = np.zeros(12, np.float64)
        return (out if self._chk(lib().orc_hit_record(self._h, r.ctypes.data, t_min, t_max, out.ctypes.data)) else None)

    def object_hit(self, obj_id, ray, t_min, t_max):
        r = np.ascontiguousarray(ray, np.float64); out = np.zeros(12, np.float64)
        return (out if self._chk(lib().orc_object_hit(self._h, obj_id, r.ctypes.data, t_min, t_max, out.ctypes.data)) else None)

    def object_bbox(self, obj_id, t0=0.0, t1=0.0):
        out = np.zeros(6, np.float64)
        return out if self._chk(lib().orc_object_bbox(self._h, obj_id, t0, t1, out.ctypes.data)) else None

    def list_bbox(self, ids, t0=0.0, t1=0.0):
        ids = np.ascontiguousarray(ids, np.uint32); out = np.zeros(6, np.float64)
        return out if self._chk(lib().orc_list_bbox(self._h, ids.size, ids.ctypes.data, t0, t1, out.ctypes.data)) else None

    def primary_rays(self, w, h, sample, seed):
        rays = np.empty((w * h, 7), np.float64)
        self._chk(lib().orc_primary_rays(self._h, w, h, sample, seed, rays.ctypes.data))
        return rays

    def render(self, w, h, spp_begin, spp_end, max_depth, integrator, seed=0x5EED, nthreads=1):
        out = np.zeros((h, w, 3), np.float64); stats = np.zeros(5, np.uint64); sec = f64(0)
        self._chk(lib().orc_render(self._h, w, h, spp_begin, spp_end, max_depth, integrator, seed, nthreads, out.ctypes.data, stats.ctypes.data, C.byref(sec)))
        return out, dict(rays=int(stats[0]), paths=int(stats[1]), box=int(stats[2]), tri=int(stats[3]), sph=int(stats[4]), seconds=sec.value)

    def sample_radiance(self, w, h, x, y, s, max_depth, integrator, seed=0x5EED):
        out = np.zeros(3, np.float64)
        self._chk(lib().orc_sample_radiance(self._h, w, h, x, y, s, max_depth, integrator, seed, out.ctypes.data))
        return out

    def sample_radiance_sig(self, w, h, x, y, s, max_depth, integrator, seed=0x5EED):
        out = np.zeros(3); sig = u64(0)
        self._chk(lib().orc_sample_radiance_sig(self._h, w, h, x, y, s, max_depth, integrator, seed, out.ctypes.data, C.byref(sig)))
        return out, int(sig.value)

    def bvh_dump(self, obj_id=0xFFFFFFFF):
        n = lib().orc_bvh_dump(self._h, obj_id, None, None)
        if n < 0:
            raise OracleError(lib().orc_last_error().decode())
        boxes = np.zeros((n, 6), np.float64); kids = np.zeros((n, 2), np.uint32)
        lib().orc_bvh_dump(self._h, obj_id, boxes.ctypes.data, kids.ctypes.data)
        return boxes, kids

    def __del__(self):
        try:
            if self._h:
                lib().orc_scene_free(self._h); self._h = None
        except Exception:
            pass


def resolve(sum_rgb, spp, tonemap=True):
    s = np.ascontiguousarray(sum_rgb, np.float64); h, w = s.shape[:2]; out = np.empty_like(s)
    lib().orc_resolve(s.ctypes.data, w, h, spp, int(tonemap), out.ctypes.data)
    return out


def quantize_ppm(rgb):
    s = np.ascontiguousarray(rgb, np.float64); h, w = s.shape[:2]; out = np.empty((h, w, 3), np.uint32)
    lib().orc_quantize_ppm(s.ctypes.data, w, h, out.ctypes.data)
    return out


def aabb_hit(box6, ray7, t_min, t_max):
    b = np.ascontiguousarray(box6, np.float64); r = np.ascontiguousarray(ray7, np.float64)
    return bool(lib().orc_aabb_hit(b.ctypes.data, r.ctypes.data, t_min, t_max))


def philox(ctr, k0, k1):
    c = np.array(ctr, np.uint32)
    lib().orc_philox(c.ctypes.data, k0, k1)
    return c


def rng_stream(seed, pixel, sample, level, n):
    out = np.empty(n, np.float64)
    lib().orc_rng_stream(seed, pixel, sample, level, n, out.ctypes.data)
    return out


def mat4_inverse(m):
    m = np.ascontiguousarray(m, np.float64); out = np.empty(16, np.float64)
    lib().orc_mat4_inverse(m.ctypes.data, out.ctypes.data)
    return out


def div_sqrt(a, b):
    a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64)
    q = np.empty_like(a); s = np.empty_like(a)
    lib().orc_div_sqrt(a.size, a.ctypes.data, b.ctypes.data, q.ctypes.data, s.ctypes.data)
    return q, s


def pdf_eval(descs, dirs, seed=0x5EED, skip=0):
    """PDF::value / PDF::generate of the oracle for descs (n x n_comp, dtype raytracing_amd.PDF_DESC; n_comp > 1 = MixturePdf)."""
    descs = np.ascontiguousarray(descs)
    descs = descs.reshape(len(descs), -1)
    n, n_comp = descs.shape
    dirs = np.ascontiguousarray(dirs, np.float64).reshape(n, 3)
    value = np.empty(n); gen = np.empty((n, 3))
    if lib().orc_pdf_eval(n, n_comp, descs.ctypes.data, dirs.ctypes.data, seed, skip, value.ctypes.data, gen.ctypes.data) < 0:
        raise OracleError(lib().orc_last_error().decode())
    return value, gen
