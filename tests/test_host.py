"""CPU tests (-m "not gpu"): host-side scene construction / BVH build / flattening against the oracle,
the C-ABI libraries load and export every declared symbol, golden vectors pin the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_ffi as orc
import raytracing_amd as rt
from raytracing_amd import _ffi
from scene_text import SceneText, scale, translate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


def test_abi_library_exports_every_declared_symbol():
    # no compute calls here (no GPU): the library must load and export what include/rt_abi.h declares
    lib = C.CDLL(_ffi.HIP_LIB)
    names = _declared("rt_abi.h", "rt_")
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(names) == sorted(_ffi.ABI_SYMBOLS)


def test_host_library_exports_every_declared_symbol():
    lib = C.CDLL(_ffi.HOST_LIB)
    names = _declared("rt_host.h", "rth_")
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(names) == sorted(_ffi.HOST_SYMBOLS)


def test_struct_layouts_match_header():
    assert C.sizeof(_ffi.Node) == 64 and C.sizeof(_ffi.Prim) == 32 and C.sizeof(_ffi.Camera) == 21 * 8
    hs = rt.HostScene.builtin("simple_triangle")
    h = hs.header()
    assert h.magic == 0x42535452 and h.version == 2 and h.total_bytes == len(hs.blob())


def _obj_of(prims, p):
    if p == rt.NONE:
        return None
    q = prims[p]
    return int(q["inst_id"]) if q["inst_id"] != rt.NONE else int(q["object_id"])


@pytest.mark.parametrize("name,seed,axis", [("random", 1, "seeded"), ("random", 7, "seeded"), ("mesh", 3, "seeded"), ("single_sphere_light", 2, "seeded"),
                                            ("random", 1, "min_area"), ("mesh", 3, "min_area"), ("random_blur", 2, "min_area")])
def test_host_bvh_equals_oracle_bvh(name, seed, axis):
    # bvh.rs:36-87 built twice, independently (host library vs oracle), from the same axis source (the seeded stream, or the
    # min-area choice among the three axes bvh.rs:53 can draw): same pre-order node boxes (bitwise) and the same leaf objects
    hs = rt.HostScene.builtin(name, bvh_seed=seed, bvh_axis=axis)
    osc = orc.OracleScene(hs.describe())
    boxes, kids = osc.bvh_dump()
    h = hs.header(); n = hs.nodes()[:h.root_end]; pr = hs.prims()
    assert len(boxes) == h.root_end
    assert np.array_equal(boxes[:, :3], n["bmin"]) and np.array_equal(boxes[:, 3:], n["bmax"])
    for i in range(len(boxes)):
        l, r = _obj_of(pr, n[i]["prim0"]), _obj_of(pr, n[i]["prim1"])
        kl, kr = int(kids[i][0]), int(kids[i][1])
        if kl == rt.NONE:
            assert l is None and r is None
        else:
            assert l == kl and (r == kr or (r is None and kl == kr and n[i]["flags"] & 2))


def _median_split_tree(boxes, ids, pick_axis):
    """bvh.rs:36-87 in numpy over object boxes [n,6]: returns the pre-order list of node boxes; `pick_axis(boxes, ids)` is the RNG."""
    out = []
    def rec(ids):
        at = len(out); out.append(None)
        if len(ids) > 2:
            order = ids[np.argsort(boxes[ids, pick_axis(boxes, ids)], kind="stable")]
            mid = len(order) // 2
            rec(order[:mid]); rec(order[mid:])
        out[at] = np.concatenate([boxes[ids, :3].min(axis=0), boxes[ids, 3:].max(axis=0)])
    rec(ids)
    return np.array(out)


def test_min_area_tree_is_a_tree_the_reference_can_draw():
    # the min-area mode only chooses WHICH of bvh.rs:53's three possible draws a node takes: feeding those choices, as an axis
    # sequence, to a plain restatement of bvh.rs:36-87 gives the same tree; and its area cost is below the seeded stream's
    hs = rt.HostScene.builtin("random", bvh_seed=1, bvh_axis="min_area")
    h = hs.header(); n = hs.nodes()[:h.root_end]
    text = hs.describe().decode()
    osc = orc.OracleScene(text)
    boxes, kids = osc.bvh_dump()
    assert np.array_equal(boxes[:, :3], n["bmin"]) and np.array_equal(boxes[:, 3:], n["bmax"])
    line = next(l for l in text.splitlines() if l.startswith("scene_objects"))
    obj_ids = [int(x) for x in line.split()[2:]]
    ob = np.array([osc.object_bbox(i) for i in obj_ids])      # the objects' own boxes (shape.rs bounding_box), by position in the list
    ids = np.arange(len(obj_ids))

    def half_cost(b, sel):
        d = b[sel, 3:].max(axis=0) - b[sel, :3].min(axis=0)
        return (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]) * len(sel)

    def min_area(b, sel):
        costs = []
        for a in range(3):
            o = sel[np.argsort(b[sel, a], kind="stable")]
            costs.append(half_cost(b, o[:len(o) // 2]) + half_cost(b, o[len(o) // 2:]))
        return int(np.argmin(costs))          # first minimum = lowest axis on ties

    mine = _median_split_tree(ob, ids, min_area)
    assert np.array_equal(mine, boxes)
    seeded = rt.HostScene.builtin("random", bvh_seed=1)
    ns = seeded.nodes()[:seeded.header().root_end]
    def area(q):   # summed half-area of the nodes that do not hold the 1000-unit ground sphere
        d = q["bmax"] - q["bmin"]
        d = d[d.max(axis=1) < 100.0]
        return float((d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]).sum())
    assert area(n) < 0.95 * area(ns)


@pytest.mark.parametrize("axis", ["seeded", "min_area"])
def test_instance_sub_bvhs_equal_oracle(axis):
    # cornell: Transformed(BVHNode::from_mesh(cube)) x6 over 4 distinct cube BVHs (main.rs:117-188)
    hs = rt.HostScene.builtin("cornell", bvh_seed=4, bvh_axis=axis)
    osc = orc.OracleScene(hs.describe())
    h = hs.header(); nodes = hs.nodes(); pr = hs.prims()
    inst = [p for p in pr if p["kind"] == 4]
    assert len(inst) == 6 and len({int(p["index"]) for p in inst}) == 4
    for p in inst:
        bvh_obj = int(p["object_id"])
        boxes, kids = osc.bvh_dump(bvh_obj)
        sub = nodes[p["index"]:p["sub_end"]]
        assert len(sub) == len(boxes)
        assert np.array_equal(boxes[:, :3], sub["bmin"]) and np.array_equal(boxes[:, 3:], sub["bmax"])
        assert sub["skip"].max() == p["sub_end"] and (sub["skip"] > np.arange(p["index"], p["sub_end"])).all()
        assert p["xform"] != rt.NONE and p["inst_id"] != rt.NONE


@pytest.mark.parametrize("name", ["random", "cornell", "simple", "simple_triangle", "mesh", "simple_light", "sphere"])
def test_flatten_invariants(name):
    hs = rt.HostScene.builtin(name, bvh_seed=2, legacy_bvh=(name == "simple_triangle"))
    h = hs.header(); nodes = hs.nodes(); pr = hs.prims()
    idx = np.arange(h.n_nodes)
    assert (nodes["skip"] > idx).all() and (nodes["skip"] <= h.n_nodes).all()
    assert (nodes["skip"][h.root_node:h.root_end] <= h.root_end).all()
    used = np.concatenate([nodes["prim0"], nodes["prim1"]]); used = used[used != rt.NONE]
    assert len(np.unique(used)) == len(used) == h.n_prims       # every primitive referenced exactly once
    inner = nodes["prim0"] == rt.NONE
    assert (nodes["skip"][~inner] == idx[~inner] + 1).all()     # leaves have no subtree
    assert (nodes["prim1"][inner] == rt.NONE).all()
    assert h.n_tris == (pr["kind"] == 0).sum() and h.n_spheres == (pr["kind"] == 1).sum() and h.n_moving_spheres == (pr["kind"] == 2).sum()


def test_reference_scene_constants():
    # what the reference scene functions construct: object counts, lights, camera
    h = rt.HostScene.builtin("mesh").header()
    assert (h.n_tris, h.n_spheres, h.n_lights, h.n_xforms, h.background) == (29702, 3, 2, 1, 0)     # main.rs:789-889
    assert h.n_nodes == 32767
    assert list(h.camera.origin) == [-2.0, 30.0, 30.0] and h.camera.lens_radius == 0.05 and h.camera.time1 == 0.0
    h = rt.HostScene.builtin("simple_triangle").header()
    assert (h.n_tris, h.n_spheres, h.n_lights, h.background, h.n_nodes) == (10, 1, 0, 1, 11)         # main.rs:547-633
    h = rt.HostScene.builtin("simple").header()
    assert (h.n_spheres, h.n_xforms, h.n_texel_bytes, h.background) == (6, 1, 1024 * 512 * 3, 1)    # main.rs:281-362
    assert h.camera.time1 == 1.0
    h = rt.HostScene.builtin("random").header()
    assert h.n_lights == 2 and h.n_spheres + h.n_moving_spheres == h.n_prims and 400 < h.n_prims <= 490   # main.rs:365-494
    h = rt.HostScene.builtin("cornell").header()
    assert (h.n_tris, h.n_spheres, h.n_xforms) == (48, 1, 6)                                         # 4 cube meshes x 12, 6 transforms


def test_describe_round_trip():
    for name in ("random", "cornell", "simple_triangle"):
        hs = rt.HostScene.builtin(name, bvh_seed=9)
        text = hs.describe()
        hs2 = rt.HostScene.from_description(text)
        assert hs2.describe() == text
        assert hs2.blob() == hs.blob()


def test_a_description_does_not_change_the_callers_axis_settings():
    # scene_from_description parses `bvh_seed` / `bvh_axis_mode` records into the process-wide build settings; they must be
    # back on the way out, or a later rth_scene_builtin from C / Rust silently builds the other tree
    from raytracing_amd import _ffi
    L = _ffi.host()
    text = rt.HostScene.builtin("random", bvh_seed=9, bvh_axis="min_area").describe()
    assert b"bvh_axis_mode 1" in text
    L.rth_set_bvh_axis_seed(4); L.rth_set_bvh_axis_mode(0)
    want = rt.HostScene(L.rth_scene_builtin(b"random", rt.ASSETS.encode(), 0)).blob()          # seeded stream, seed 4
    rt.HostScene.from_description(text)                                                          # min-area tree, seed 9
    got = rt.HostScene(L.rth_scene_builtin(b"random", rt.ASSETS.encode(), 0)).blob()
    assert got == want and got != rt.HostScene.from_description(text).blob()


def test_mat4_inverse_and_sphere_transform_match_oracle():
    rng = np.random.default_rng(0)
    L = _ffi.host()
    mats = [np.array(scale(1.4, 1.4, 1.4), float), np.array(translate(5, 5, 5), float), np.array(translate(1.5, 0, 0), float)]
    for _ in range(50):
        mats.append(rng.normal(size=16))
    for m in mats:
        m = np.ascontiguousarray(m, np.float64)
        out = np.empty(16)
        L.rth_mat4_inverse(m.ctypes.data, out.ctypes.data)
        assert np.array_equal(out.view(np.uint64), orc.mat4_inverse(m).view(np.uint64))
    # translation inverse is exact (transform.rs:17 for the pagoda wrapper, main.rs:816)
    out = np.empty(16); m = np.array(translate(5, 5, 5), float)
    L.rth_mat4_inverse(m.ctypes.data, out.ctypes.data)
    assert list(out) == translate(-5, -5, -5)
    # Sphere::transform (shape.rs:23-34) as used by create_simple_scene (main.rs:323-333): scale(.5) * translate(0,2,0)
    s = np.array([0, 0, 0, 1.0]); m = np.array(scale(.5, .5, .5), float).reshape(4, 4).T @ np.array(translate(0, 2, 0), float).reshape(4, 4).T
    m = np.ascontiguousarray(m.T.reshape(16)); out = np.empty(4)
    L.rth_sphere_transform(s.ctypes.data, m.ctypes.data, out.ctypes.data)
    assert list(out) == [0.0, 1.0, 0.0, 0.5]


def test_image_tile_kats_host_api():  # renderer/src/image.rs:100-162 against the host mirror of Image / ImageTile
    L = _ffi.host()
    out = (C.c_uint32 * 4)(); xy = (C.c_uint32 * 2)()
    L.rth_image_get_tile(100, 100, 0, 0, 100, 100, out); assert list(out) == [0, 0, 100, 100]
    L.rth_tile_get_xy(0, 0, 100, 100, 100 * 100 - 1, xy); assert list(xy) == [99, 99]
    L.rth_image_get_tile(100, 100, 50, 50, 50, 50, out); assert list(out) == [50, 50, 50, 50]
    L.rth_tile_get_xy(50, 50, 50, 50, 49, xy); assert list(xy) == [99, 50]
    L.rth_image_get_tile(100, 100, 90, 0, 50, 50, out); assert list(out) == [90, 0, 10, 50]
    L.rth_tile_get_xy(90, 0, 10, 50, 10, xy); assert list(xy) == [90, 1]
    # merge_tile_works (image.rs:130-162)
    img = np.zeros((4, 4, 3)); t1 = np.full((2, 2, 3), 10.0); t2 = np.full((2, 2, 3), 100.0)
    assert L.rth_image_merge_tile(img.ctypes.data, 4, 4, 0, 0, 2, 2, t1.ctypes.data) == 0
    assert L.rth_image_merge_tile(img.ctypes.data, 4, 4, 2, 2, 2, 2, t2.ctypes.data) == 0
    flat = img.reshape(16, 3)
    for i in (0, 1, 4, 5): assert list(flat[i]) == [10.0] * 3
    for i in (10, 11, 14, 15): assert list(flat[i]) == [100.0] * 3
    for i in (2, 3, 6, 7, 8, 9, 12, 13): assert list(flat[i]) == [0.0] * 3
    # Image::put scrubs NaN channels (image.rs:48-56)
    t3 = np.array([[[np.nan, 1.0, np.nan]]])
    L.rth_image_merge_tile(img.ctypes.data, 4, 4, 1, 1, 1, 1, t3.ctypes.data)
    assert list(flat[5]) == [0.0, 1.0, 0.0]


def test_ppm_writer_matches_oracle(tmp_path):  # exporters/ppm.rs:6-38
    rng = np.random.default_rng(5)
    img = rng.uniform(-0.1, 1.3, (7, 5, 3)); img[0, 0, 0] = np.nan
    p = str(tmp_path / "o.ppm")
    rt.write_ppm(p, img)
    toks = open(p).read().split()
    assert toks[:4] == ["P3", "5", "7", "255"]
    vals = np.array(toks[4:], np.uint32).reshape(7, 5, 3)
    assert np.array_equal(vals, orc.quantize_ppm(img))      # rows bottom-up, sqrt gamma, clamp .999, *256 truncation
    assert vals.max() == 255 and vals[-1, 0, 0] == 0


def test_pfm_and_png_writers(tmp_path):
    # SURVEY.md 8(f)-2 "PNG/PFM output for diffing": no counterpart in the reference; the PNG must hold what the PPM prints
    import zlib, struct
    rng = np.random.default_rng(6)
    for shape in ((7, 5, 3), (300, 257, 3)):                 # the larger one spans several stored deflate blocks
        img = rng.uniform(-0.1, 1.3, shape); img[0, 0, 0] = np.nan
        h, w = shape[:2]
        p = str(tmp_path / "o.pfm")
        rt.write_pfm(p, img)
        raw = open(p, "rb").read()
        head = f"PF\n{w} {h}\n-1.0\n".encode()
        assert raw.startswith(head) and len(raw) == len(head) + 4 * img.size
        back = np.frombuffer(raw, "<f4", img.size, len(head)).reshape(shape)
        assert np.array_equal(back, img.astype(np.float32), equal_nan=True)          # rows bottom-up, as the accumulator
        p = str(tmp_path / "o.png")
        rt.write_png(p, img)
        raw = open(p, "rb").read()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n"
        off, chunks = 8, []
        while off < len(raw):
            n, = struct.unpack(">I", raw[off:off + 4]); typ = raw[off + 4:off + 8]; data = raw[off + 8:off + 8 + n]
            assert struct.unpack(">I", raw[off + 8 + n:off + 12 + n])[0] == zlib.crc32(typ + data)
            chunks.append((typ, data)); off += 12 + n
        assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
        assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 2, 0, 0, 0)
        px = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)
        assert not px[:, 0].any()
        assert np.array_equal(px[:, 1:].reshape(h, w, 3), orc.quantize_ppm(img))       # top row first, same bytes as the PPM


def test_error_behaviour():
    with pytest.raises(rt.RtError):
        rt.HostScene.from_description("not a scene")
    s = SceneText(); m = s.lambertian_rgb(.5, .5, .5)
    r = s.aarect(0, 1, 0, 1, 0, m)
    with pytest.raises(rt.RtError):      # only Sphere is a SampleableLight (shape.rs:36)
        rt.HostScene.from_description(s.finish([r], lights=[r]))
    s = SceneText(); m = s.lambertian_rgb(.5, .5, .5)
    sp = s.sphere((0, 0, 0), 1, m); x = s.xform(scale(2, 2, 2)); t1 = s.transformed(x, sp); t2 = s.transformed(x, t1)
    hs = rt.HostScene.from_description(s.finish([t2]))
    with pytest.raises(rt.RtError):      # nesting the flattener does not support is an error code, never a crash
        hs.blob()
    with pytest.raises(rt.RtError):
        rt.HostScene.builtin("no_such_scene")
    with pytest.raises(rt.RtError):      # bvh.rs:36 on an empty slice: the reference would recurse forever / panic
        rt.HostScene.from_description(SceneText().finish([], build_bvh=True))


def test_golden_vectors_pin_the_oracle():
    g = np.load(os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz"))
    for name in ("simple_triangle", "random", "cornell", "simple", "sphere", "mesh"):
        hs = rt.HostScene.builtin(name, bvh_seed=5)
        osc = orc.OracleScene(hs.describe())
        w, h = map(int, g[f"{name}_wh"])
        rays = osc.primary_rays(w, h, 0, 0x5EED)
        assert np.array_equal(rays.view(np.uint64), g[f"{name}_rays"].view(np.uint64))
        prim, inst, t, cnt = osc.trace_closest(rays, 0.001, 100000.0)
        assert np.array_equal(prim, g[f"{name}_prim"]) and np.array_equal(inst, g[f"{name}_inst"])
        assert np.array_equal(t.view(np.uint64), g[f"{name}_t"].view(np.uint64)) and np.array_equal(cnt, g[f"{name}_cnt"])
        picks = g[f"{name}_picks"][:6]
        for integ in range(4):
            got = np.array([osc.sample_radiance(w, h, int(x), int(y), int(s), 5, integ, 77) for x, y, s in picks])
            assert np.allclose(got, g[f"{name}_rad{integ}"][:6], rtol=1e-12, atol=1e-15, equal_nan=True)


def test_oracle_rng_and_camera_statistics():
    # Philox stream sanity: uniform in [0,1), distinct per (pixel, sample, level)
    a = orc.rng_stream(1, 2, 3, 4, 4096); b = orc.rng_stream(1, 2, 3, 5, 4096)
    assert a.min() >= 0 and a.max() < 1 and abs(a.mean() - 0.5) < 0.02 and not np.array_equal(a, b)
    # cosine pdf integrates to 1 / cone pdf = 1 / solid angle are exercised through the furnace test (test_oracle_kat.py)
