"""GPU parity under stress of the fp32 filters of k_trace.

The production kernel classifies box and triangle tests in fp32 and hands undecided lanes to the exact f64
tests; results may only depend on the error bounds being valid.  These cases aim rays AT the decision
boundaries (triangle edges and vertices, box faces and corners, coplanar rays), scale the scene to the
ends of the fp32 range where the bounds are disabled, and use rays whose magnitudes make the margins
huge.  Every case must be bit-exact against the oracle, from the production kernel and from the counting
kernel, and the counting kernel (which evaluates filter AND exact test for every decision) must report
zero contradicted decisions."""
import numpy as np
import pytest

import oracle_ffi as orc
import raytracing_amd as rt
from scene_text import SceneText, scale, translate

pytestmark = pytest.mark.gpu


def _soup(seed, n_tri, extent, wrap=None, bvh_seed=4):
    """Random triangle soup (one mesh, BVH over the triangles, optionally every triangle under one transform)."""
    rng = np.random.default_rng(seed)
    s = SceneText(bvh_seed=bvh_seed)
    m = s.lambertian_rgb(.5, .5, .5)
    centres = rng.uniform(-extent, extent, (n_tri, 3))
    verts, faces = [], []
    for k in range(n_tri):
        for _ in range(3):
            verts.append(tuple(centres[k] + rng.normal(size=3) * extent * 0.08))
        faces.append((3 * k, 3 * k + 1, 3 * k + 2))
    me = s.mesh(verts, [(0, 0, 1)] * len(verts), [(0, 0)] * len(verts), faces, m)
    xf = s.xform(wrap) if wrap is not None else None
    ids = []
    for k in range(n_tri):
        t = s.triangle(me, k)
        ids.append(s.transformed(xf, t) if xf is not None else t)
    return s.finish(ids, build_bvh=True), np.array(verts).reshape(n_tri, 3, 3), rng


def _check(gpu_ctx, text, rays, t_min=0.001, t_max=1e5):
    osc = orc.OracleScene(text)
    gs = gpu_ctx.upload(rt.HostScene.from_description(text))
    o = osc.trace_closest(rays, t_min, t_max)
    g = gs.trace_closest(rays, t_min, t_max)
    c = gs.trace_closest(rays, t_min, t_max, counters=True)
    gs.close()
    for r in (g, c):
        assert np.array_equal(r[0], o[0]), f"{(r[0] != o[0]).sum()} primitive ids differ"
        assert np.array_equal(r[2].view(np.uint64), o[2].view(np.uint64)), "t differs in bits"
    assert list(map(int, c[3][:3])) == list(map(int, o[3][:3])), (c[3], o[3])
    assert int(c[3][3]) == 0, f"{int(c[3][3])} filter decisions contradicted by the exact test"
    return o


def _rays_at(points, origins, jitter_ulps=0, rng=None):
    d = points - origins
    if jitter_ulps:
        d = d * (1.0 + rng.integers(-jitter_ulps, jitter_ulps + 1, d.shape) * 2.0 ** -52)
    rays = np.zeros((len(points), 7))
    rays[:, :3] = origins
    rays[:, 3:6] = d
    return rays


def test_rays_at_triangle_edges_vertices_and_planes(gpu_ctx):
    text, tris, rng = _soup(1, 300, 4.0)
    n = 4000
    k = rng.integers(0, len(tris), n)
    a, b = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    kind = rng.integers(0, 4, n)
    # barycentric targets: on edge u = 0, on edge u + v = 1, at a vertex, strictly inside
    u = np.where(kind == 0, 0.0, np.where(kind == 1, a, np.where(kind == 2, 1.0, a * (1 - b))))
    v = np.where(kind == 0, a, np.where(kind == 1, 1 - a, np.where(kind == 2, 0.0, b * (1 - a * (1 - b)))))
    p = tris[k, 0] + u[:, None] * (tris[k, 1] - tris[k, 0]) + v[:, None] * (tris[k, 2] - tris[k, 0])
    org = rng.uniform(-9, 9, (n, 3))
    hit = _check(gpu_ctx, text, _rays_at(p, org, jitter_ulps=3, rng=rng))
    assert (hit[0] != rt.NONE).sum() > n // 4
    # rays lying (almost) in the plane of a triangle: det ~ 0, every quantity of the u test is tiny
    e1, e2 = tris[k, 1] - tris[k, 0], tris[k, 2] - tris[k, 0]
    org = tris[k, 0] + 5.0 * e1 - 3.0 * e2
    tgt = tris[k, 0] + a[:, None] * e1 + b[:, None] * e2
    _check(gpu_ctx, text, _rays_at(tgt, org, jitter_ulps=2, rng=rng))


def test_rays_at_box_faces_and_corners(gpu_ctx):
    # every BVH box is the union of triangle boxes, so triangle-box corners and face points are box faces
    text, tris, rng = _soup(2, 200, 3.0)
    lo, hi = tris.min(axis=1), tris.max(axis=1)
    n = 4000
    k = rng.integers(0, len(tris), n)
    pick = rng.integers(0, 2, (n, 3))
    corner = np.where(pick == 0, lo[k], hi[k])
    face = corner.copy()
    ax = rng.integers(0, 3, n)
    other = (ax + 1) % 3
    face[np.arange(n), other] = rng.uniform(lo[k, other], hi[k, other])
    org = rng.uniform(-8, 8, (n, 3))
    # axis-parallel origins put exact zeros into the direction for a part of the rays (literal f64 path)
    org[::7, 0] = corner[::7, 0]
    _check(gpu_ctx, text, _rays_at(corner, org))
    _check(gpu_ctx, text, _rays_at(face, org, jitter_ulps=1, rng=rng))
    # shadow-ray style windows: t_max exactly at / just around a box entry distance
    rays = _rays_at(corner, org)
    _check(gpu_ctx, text, rays, t_min=0.001, t_max=1.0)
    _check(gpu_ctx, text, rays, t_min=1.0, t_max=1.0 + 2.0 ** -30)


@pytest.mark.parametrize("s", [1e-25, 1e-12, 1e9, 1e28, 1e33])
def test_scene_scales_up_to_where_the_filters_switch_off(gpu_ctx, s):
    # 1e28 .. 1e33: products leave the fp32 range, the filters must answer "unknown", never "certain"
    text, tris, rng = _soup(3, 120, 2.0 * s)
    n = 1500
    k = rng.integers(0, len(tris), n)
    p = tris[k].mean(axis=1) + rng.normal(size=(n, 3)) * 0.05 * s
    org = rng.uniform(-6, 6, (n, 3)) * s
    o = _check(gpu_ctx, text, _rays_at(p, org), t_min=0.001, t_max=1e300)
    if s >= 1.0:  # below, |det| < EPSILON for every triangle: the reference itself hits nothing (mesh.rs:59-61)
        assert (o[0] != rt.NONE).sum() > 0


def test_extreme_ray_magnitudes_and_far_origins(gpu_ctx):
    text, tris, rng = _soup(4, 200, 3.0, wrap=translate(5.0, -2.0, 1.0))
    n = 3000
    k = rng.integers(0, len(tris), n)
    p = tris[k].mean(axis=1) + np.array([5.0, -2.0, 1.0]) + rng.normal(size=(n, 3)) * 0.1
    org = rng.uniform(-7, 7, (n, 3))
    rays = _rays_at(p, org)
    mag = 10.0 ** rng.uniform(-36, 36, n)   # |d| from 1e-36 to 1e36: 1/d leaves the filter's range on both sides
    rays[:, 3:6] *= mag[:, None]
    _check(gpu_ctx, text, rays, t_min=0.0, t_max=1e300)
    far = _rays_at(p, org * 1e7)            # origins 1e7 away: margins of order 1, nearly everything is "unknown"
    _check(gpu_ctx, text, far)
    mixed = rays.copy()
    mixed[::5, 3] = 0.0; mixed[::11, 4] = -0.0; mixed[::13, 5] = 0.0   # zero components inside otherwise plain waves
    _check(gpu_ctx, text, mixed, t_min=0.0, t_max=1e300)


def test_single_triangle_leaves_and_mixed_leaves(gpu_ctx):
    # odd object counts give 1-object leaves (duplicate right child, bvh.rs:37-43); spheres mixed in give
    # sphere leaves, triangle+sphere leaves (generic exact path) and leaves whose halves use different matrices
    rng = np.random.default_rng(9)
    s = SceneText(bvh_seed=2)
    m = s.lambertian_rgb(.5, .5, .5)
    verts, faces = [], []
    for k in range(37):
        c = rng.uniform(-3, 3, 3)
        for _ in range(3):
            verts.append(tuple(c + rng.normal(size=3) * 0.4))
        faces.append((3 * k, 3 * k + 1, 3 * k + 2))
    me = s.mesh(verts, [(0, 0, 1)] * len(verts), [(0, 0)] * len(verts), faces, m)
    xa, xb = s.xform(translate(0.5, 0, 0)), s.xform(scale(1.5, 0.75, 1.25))
    ids = []
    for k in range(37):
        t = s.triangle(me, k)
        ids.append(t if k % 3 == 0 else s.transformed(xa if k % 3 == 1 else xb, t))
    for k in range(20):
        sp = s.sphere(tuple(rng.uniform(-3, 3, 3)), float(rng.uniform(0.1, 0.6)), m)
        ids.append(sp if k % 2 else s.transformed(xa, sp))
    ids.append(s.aarect(-1, 1, -1, 1, 0.5, m))
    text = s.finish(ids, build_bvh=True)
    n = 4000
    org = rng.uniform(-6, 6, (n, 3))
    tgt = rng.uniform(-3, 3, (n, 3))
    _check(gpu_ctx, text, _rays_at(tgt, org))


def _plane_rays(tris, rng, n):
    """Rays built to lie in (or within eta of) the plane of some triangle, starting on or just off that plane: the
    rays for which Moller-Trumbore's det is tiny and the reference may accept anything (mesh.rs:59 rejects only
    |det| < 2.2e-16) -- the case strict pruning must recognise and never prune on."""
    k = rng.integers(0, len(tris), n)
    v0, e1, e2 = tris[k, 0], tris[k, 1] - tris[k, 0], tris[k, 2] - tris[k, 0]
    nh = np.cross(e1, e2); nh /= np.linalg.norm(nh, axis=1)[:, None]
    a, b = rng.normal(size=n), rng.normal(size=n)
    eta = 10.0 ** rng.uniform(-14, -3, n) * rng.choice([-1, 1], n)
    eta[::4] = 0.0
    off = 10.0 ** rng.uniform(-12, -1, n) * rng.choice([-1, 1], n)
    off[::3] = 0.0
    d = a[:, None] * e1 + b[:, None] * e2
    d = d / np.abs(d).max(1)[:, None] + eta[:, None] * nh
    o = v0 + rng.normal(size=(n, 1)) * e1 * 4 + rng.normal(size=(n, 1)) * e2 * 4 + off[:, None] * nh
    rays = np.zeros((n, 7)); rays[:, :3] = o; rays[:, 3:6] = d
    return rays


@pytest.mark.parametrize("wrap", [None, scale(1.5, 0.75, 1.25), translate(5.0, 5.0, 5.0)])
def test_strict_pruning_never_loses_a_hit(gpu_ctx, wrap):
    # 400 triangles (pruning is on from 64), optionally under a transform so that the pruned space is not the world
    text, tris, rng = _soup(21, 400, 4.0, wrap=wrap)
    hs = rt.HostScene.from_description(text)
    assert hs.prune_info()["on"] == 1.0
    n = 6000
    m = np.array(wrap if wrap is not None else scale(1, 1, 1), float).reshape(4, 4).T   # space of the triangles -> world
    org = rng.uniform(-9, 9, (n, 3)); tgt = rng.uniform(-4, 4, (n, 3)) @ m[:3, :3].T + m[:3, 3]
    o = _check(gpu_ctx, text, _rays_at(tgt, org))                       # ordinary rays
    assert (o[0] != rt.NONE).sum() > n // 10
    sky = _rays_at(org * 30.0, org * 3.0)                               # rays that leave the scene: pruned at the top
    assert (_check(gpu_ctx, text, sky)[0] != rt.NONE).sum() < n // 5
    # in-plane rays, in the space of the triangles: map them to world space for the wrapped scenes
    pr = _plane_rays(tris, rng, n)
    pr[:, :3] = pr[:, :3] @ m[:3, :3].T + m[:3, 3]
    pr[:, 3:6] = pr[:, 3:6] @ m[:3, :3].T
    _check(gpu_ctx, text, pr, t_min=0.001, t_max=1e5)
    _check(gpu_ctx, text, pr, t_min=0.0, t_max=1e300)
    axis = _rays_at(tgt, org); axis[:, 3:6] = np.eye(3)[rng.integers(0, 3, n)] * rng.choice([-1.0, 1.0], (n, 1))   # zero components: literal path, no pruning
    _check(gpu_ctx, text, axis)


def test_strict_pruning_on_the_pagoda_with_surface_and_plane_rays(gpu_ctx):
    hs = rt.HostScene.builtin("mesh", bvh_seed=3)
    text = hs.describe()
    geo = hs.geoms(); tri = geo[geo["kind"] == 0]["d"]
    tris = np.stack([tri[:, 0:3], tri[:, 0:3] + tri[:, 3:6], tri[:, 0:3] + tri[:, 6:9]], axis=1)
    rng = np.random.default_rng(31)
    n = 8000
    k = rng.integers(0, len(tris), n)
    u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n); fl = u + v > 1; u[fl], v[fl] = 1 - u[fl], 1 - v[fl]
    p = tris[k, 0] + u[:, None] * (tris[k, 1] - tris[k, 0]) + v[:, None] * (tris[k, 2] - tris[k, 0]) + 5.0   # main.rs:816: the pagoda sits at +5
    rays = np.zeros((n, 7)); rays[:, :3] = p; rays[:, 3:6] = rng.normal(size=(n, 3))                         # bounce-like rays from the surfaces
    o = _check(gpu_ctx, text, rays)
    assert (o[0] != rt.NONE).sum() > n // 3
    pr = _plane_rays(tris, rng, n); pr[:, :3] += 5.0
    _check(gpu_ctx, text, pr)


def test_strict_pruning_of_spheres_and_moving_spheres(gpu_ctx):
    # the random scene (main.rs:365-494) is spheres and moving spheres only: every inner node is prunable, no certificate needed.
    # Grazing rays (distance from the centre = r (1 + eps), eps down to 0) put the discriminant of shape.rs:93-121 at its rounding
    # error; times outside the camera's shutter must switch pruning off, not move a sphere out of its box.
    hs = rt.HostScene.builtin("random", bvh_seed=3)
    info = hs.prune_info()
    assert info["on"] == 1.0 and info["normals"] == 0.0
    text = hs.describe()
    geo = hs.geoms()
    rng = np.random.default_rng(41)
    n = 6000
    k = rng.integers(0, len(geo), n)
    tm = rng.uniform(0.0, 1.0, n)
    tm[::5] = rng.uniform(-0.5, 1.5, len(tm[::5]))
    mv = geo["kind"][k] == 2
    d = geo["d"][k]
    frac = np.where(mv, (tm - d[:, 6]) / np.where(mv, d[:, 7] - d[:, 6], 1.0), 0.0)
    c = np.where(mv[:, None], d[:, 0:3] + (d[:, 3:6] - d[:, 0:3]) * frac[:, None], d[:, 0:3])
    r = np.abs(np.where(mv, d[:, 8], d[:, 3]))
    dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    perp = np.cross(dirs, rng.normal(size=(n, 3))); perp /= np.linalg.norm(perp, axis=1)[:, None]
    eps = 10.0 ** rng.uniform(-16, -3, n) * rng.choice([-1, 1], n); eps[::4] = 0.0
    p = c + perp * (r * (1 + eps))[:, None]                       # point of closest approach, at r (1 + eps) from the centre
    rays = np.zeros((n, 7)); rays[:, :3] = p - dirs * rng.uniform(0.5, 30, n)[:, None]; rays[:, 3:6] = dirs * 10.0 ** rng.uniform(-3, 3, n)[:, None]; rays[:, 6] = tm
    o = _check(gpu_ctx, text, rays)
    assert 0 < (o[0] != rt.NONE).sum()
    org = rng.uniform(-15, 15, (n, 3)); org[:, 1] = rng.uniform(0.1, 8, n)
    rnd = _rays_at(rng.uniform(-11, 11, (n, 3)) * np.array([1, 0.1, 1]), org); rnd[:, 6] = rng.uniform(0, 1, n)
    _check(gpu_ctx, text, rnd)


def test_strict_pruning_next_to_instances_mixed_leaves_and_rectangles(gpu_ctx):
    # the nested-walk kernel variant with pruning on: a triangle soup (pruned space = world), spheres and moving spheres in
    # the same space (covered), an instanced cube BVH under a transform, a rectangle and a sphere under another transform
    # (their ancestors are never pruned)
    rng = np.random.default_rng(51)
    s = SceneText(bvh_seed=6)
    m = s.lambertian_rgb(.5, .5, .5)
    verts, faces = [], []
    for k in range(150):
        c = rng.uniform(-4, 4, 3)
        for _ in range(3):
            verts.append(tuple(c + rng.normal(size=3) * 0.35))
        faces.append((3 * k, 3 * k + 1, 3 * k + 2))
    me = s.mesh(verts, [(0, 0, 1)] * len(verts), [(0, 0)] * len(verts), faces, m)
    ids = [s.triangle(me, k) for k in range(150)]
    for k in range(30):
        c = rng.uniform(-4, 4, 3)
        ids.append(s.sphere(tuple(c), float(rng.uniform(0.1, 0.5)), m) if k % 2 else
                   s.moving_sphere(tuple(c), tuple(c + rng.normal(size=3) * 0.3), 0.0, 1.0, float(rng.uniform(0.1, 0.4)), m))
    cube_v = [(x, y, z) for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]
    cube_f = [(0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1), (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)]
    cm = s.mesh(cube_v, [(0, 0, 1)] * 8, [(0, 0)] * 8, cube_f, m)
    cube = s.bvh([s.triangle(cm, k) for k in range(12)])
    ids.append(s.transformed(s.xform(translate(2.0, 1.0, -1.0)), cube))
    ids.append(s.aarect(-2, 2, -2, 2, -3.0, m))
    ids.append(s.transformed(s.xform(scale(1.3, 0.7, 1.1)), s.sphere((1.0, -2.0, 0.5), 0.6, m)))
    text = s.finish(ids, build_bvh=True)
    hs = rt.HostScene.from_description(text)
    info = hs.prune_info()
    assert info["on"] == 1.0 and 0 < info["prunable_nodes"] < (hs.nodes()["prim0"] == rt.NONE).sum()
    n = 6000
    org = rng.uniform(-9, 9, (n, 3)); tgt = rng.uniform(-4, 4, (n, 3))
    rays = _rays_at(tgt, org); rays[:, 6] = rng.uniform(-0.2, 1.2, n)
    o = _check(gpu_ctx, text, rays)
    assert (o[0] != rt.NONE).sum() > n // 5 and (o[1] != rt.NONE).sum() > 0      # some hits inside the instance
    tris = np.array(verts).reshape(150, 3, 3)
    pr = _plane_rays(tris, rng, n); pr[:, 6] = rng.uniform(0, 1, n)
    _check(gpu_ctx, text, pr)


def test_production_stream_without_always_entered_nodes(gpu_ctx):
    # big far spheres next to a small triangle cloud under a transform: the spheres' ancestors hold every prunable box and
    # have no record in the stream the pruning kernel walks.  Rays that MISS those ancestors' boxes
    # in the reference (from outside, pointing away or grazing their faces) must still give the reference's answer.
    rng = np.random.default_rng(61)
    s = SceneText(bvh_seed=4)
    m = s.lambertian_rgb(.5, .5, .5)
    nt = 300
    v0 = rng.uniform(-1, 1, (nt, 3))
    tri = np.stack([v0, v0 + rng.normal(0, 0.1, (nt, 3)), v0 + rng.normal(0, 0.1, (nt, 3))], 1)
    flat = tri.reshape(-1, 3)
    me = s.mesh([tuple(map(float, p)) for p in flat], [(0, 0, 1)] * len(flat), [(0, 0)] * len(flat), [(3 * k, 3 * k + 1, 3 * k + 2) for k in range(nt)], m)
    shift = np.array([0.5, 2.0, -0.25])
    xf = s.xform(translate(*shift))
    ids = [s.transformed(xf, s.triangle(me, k)) for k in range(nt)]
    ids += [s.sphere((0.0, -1003.0 - 40.0 * i, 5.0 * i), 1000.0, m) for i in range(4)]
    text = s.finish(ids, build_bvh=True)
    hs = rt.HostScene.from_description(text)
    info = hs.prune_info()
    assert info["on"] == 1.0 and info["elided_nodes"] >= 1
    nodes = hs.nodes()
    n = 6000
    org = rng.uniform(-6, 6, (n, 3)) + shift; tgt = rng.uniform(-1, 1, (n, 3)) + shift
    o = _check(gpu_ctx, text, _rays_at(tgt, org))
    assert (o[0] != rt.NONE).sum() > n // 10
    # from far outside the root box, towards and away from it, and along its faces
    root = nodes[hs.header().root_node]
    lo, hi = root["bmin"], root["bmax"]
    far = rng.uniform(-1, 1, (n, 3)); far /= np.linalg.norm(far, axis=1)[:, None]
    far_o = (lo + hi) / 2 + far * 4000.0
    _check(gpu_ctx, text, _rays_at(tgt, far_o), t_max=1e7)
    _check(gpu_ctx, text, _rays_at(far_o * 2.0, far_o), t_max=1e7)
    face = rng.uniform(lo, hi, (n, 3)); k = rng.integers(0, 3, n); side = rng.integers(0, 2, n)
    face[np.arange(n), k] = np.where(side == 1, hi[k], lo[k])
    _check(gpu_ctx, text, _rays_at(face, org, jitter_ulps=3, rng=rng), t_max=1e7)
    slide = _rays_at(face, org); slide[:, :3] = face; slide[:, 3:6] = rng.normal(size=(n, 3)); slide[np.arange(n), 3 + k] *= 1e-9
    _check(gpu_ctx, text, slide, t_max=1e7)
    pr = _plane_rays(tri, rng, n); pr[:, :3] += shift
    _check(gpu_ctx, text, pr)
