"""GPU parity, rows (a1,a2,a14..a21): radiance from the wavefront pipeline vs the oracle's recursive
integrators (integrator.rs:31-397) on identical Philox streams, through the C ABI.

Tolerances (stated per test): geometry is bit-exact (test_gpu_trace.py); radiance differs from the
oracle only through libm-vs-device sin/cos/acos/atan2 (<= 1-2 ulp, amplified along the path) and,
for accumulated images, the fp32 accumulator.  Sample level: |gpu - oracle| <= 1e-9 * (|oracle| + 1e-3)
for EVERY sample whose path took the same discrete decisions on both sides.  That is checked, not assumed: both
sides report each path's signature (a hash over the object hit and its side per level, checker branch / texel,
kind of scatter, reflect-or-refract, light picked and whether its sample counted, visible or not, the MIS .001
gate; rt_sample_radiance_sig / orc_sample_radiance_sig).  A sample out of tolerance must have a differing
signature (a direction moved by an ulp across a discontinuity: explained), the unexplained remainder must be
ZERO, and -- so that a bug which changes a discrete decision cannot hide behind "explained" -- at most 0.5 % of the
samples may be out of tolerance at all (the bound of round 1, before signatures existed) and at most 1 % may carry a
differing signature.  Image level: 1e-5 relative (fp32 sums).
"""
import numpy as np
import pytest
import torch  # before the first HIP call of librt_hip.so: torch ships its own ROCm runtime and must initialise first

import oracle_ffi as orc
import raytracing_amd as rt
from scene_text import SceneText

pytestmark = pytest.mark.gpu

INTEGRATORS = [rt.INT_MIS, rt.INT_LIGHT_IS, rt.INT_BRDF, rt.INT_UNIFORM]


def _pair(name, legacy_bvh=False, seed=5, axis="seeded"):
    hs = rt.HostScene.builtin(name, bvh_seed=seed, legacy_bvh=legacy_bvh, bvh_axis=axis)
    return hs, orc.OracleScene(hs.describe())


def _sample_compare(gs, osc, w, h, picks, depth, integ, seed, tol=1e-9, max_bad=0.005, max_flipped=0.01):
    px = np.array([p[1] * w + p[0] for p in picks], np.uint32)
    sm = np.array([p[2] for p in picks], np.uint32)
    g, gsig = gs.sample_radiance_sig(w, h, px, sm, depth, integ, seed)
    both = [osc.sample_radiance_sig(w, h, x, y, s, depth, integ, seed) for x, y, s in picks]
    o = np.array([b[0] for b in both]); osig = np.array([b[1] for b in both], np.uint64)
    assert np.array_equal(g, gs.sample_radiance(w, h, px, sm, depth, integ, seed), equal_nan=True)   # the signature pass changes no radiance
    both_nan = np.isnan(g) & np.isnan(o)
    ok = (both_nan | (np.abs(g - o) <= tol * (np.abs(o) + 1e-3))).all(axis=1)
    same_path = gsig == osig
    unexplained = ~ok & same_path
    assert not unexplained.any(), f"{unexplained.sum()} samples took the same decisions as the oracle's path but differ by more than {tol}: {np.flatnonzero(unexplained)[:5]}"
    # a differing signature explains a deviation but does not excuse any number of them -- a wrong discrete decision
    # (a shadow walk that stops early on the wrong side, a wrong hit) would flip far more than an ulp at a discontinuity does
    assert (~ok).mean() <= max_bad, f"{(~ok).mean():.4f} of the samples are out of tolerance (all with another signature than the oracle's path)"
    assert (~same_path).mean() <= max_flipped, f"{(~same_path).mean():.4f} of the paths took another branch than the oracle's"
    assert len(set(osig.tolist())) > len(picks) // 20      # the signature does tell paths apart
    return g, o


@pytest.mark.parametrize("integ", INTEGRATORS)
@pytest.mark.parametrize("name,depth", [("random", 6), ("simple", 6), ("cornell", 5), ("simple_triangle", 5), ("two_spheres", 5), ("simple_light", 5)])
def test_sample_radiance_matches_oracle(gpu_ctx, name, depth, integ):
    hs, osc = _pair(name)
    gs = gpu_ctx.upload(hs)
    w, h = 80, 45
    rng = np.random.default_rng(integ * 100 + depth)
    picks = [(int(rng.integers(w)), int(rng.integers(h)), int(rng.integers(0, 1000))) for _ in range(700)]
    _sample_compare(gs, osc, w, h, picks, depth, integ, seed=0xC0FFEE)
    gs.close()


@pytest.mark.parametrize("integ,axis", [(rt.INT_MIS, "seeded"), (rt.INT_BRDF, "seeded"), (rt.INT_MIS, "min_area")])
def test_pagoda_sample_radiance(gpu_ctx, integ, axis):
    hs, osc = _pair("mesh", axis=axis)
    gs = gpu_ctx.upload(hs)
    w, h = 64, 36
    rng = np.random.default_rng(3)
    picks = [(int(rng.integers(w)), int(rng.integers(h)), int(rng.integers(0, 100))) for _ in range(300)]
    _sample_compare(gs, osc, w, h, picks, 5, integ, seed=11)
    gs.close()


def test_depth_50_random_spheres(gpu_ctx):
    # BASELINE.json configs[0]: RTIOW random spheres, depth 50, MIS
    hs, osc = _pair("random")
    gs = gpu_ctx.upload(hs)
    w, h = 100, 56
    rng = np.random.default_rng(50)
    picks = [(int(rng.integers(w)), int(rng.integers(h)), int(rng.integers(0, 10))) for _ in range(400)]
    _sample_compare(gs, osc, w, h, picks, 50, rt.INT_MIS, seed=1)
    gs.close()


@pytest.mark.parametrize("name,integ,depth", [("random", rt.INT_MIS, 5), ("simple", rt.INT_BRDF, 5), ("cornell", rt.INT_MIS, 5)])
def test_image_sums_match_oracle(gpu_ctx, name, integ, depth):
    # whole frames, several passes (samples_per_pass=2 over 5 spp => ragged last pass), fp32 accumulators
    hs, osc = _pair(name)
    gs = gpu_ctx.upload(hs)
    w, h, spp = 96, 54, 5
    acc, st = gs.render_host(w, h, 0, spp, depth, integ, seed=42, samples_per_pass=2)
    ref, ost = osc.render(w, h, 0, spp, depth, integ, seed=42, nthreads=8)
    assert st.rays == ost["rays"] and st.paths == ost["paths"] == w * h * spp
    err = np.abs(acc - ref) / (np.abs(ref) + 1e-3)
    assert (err <= 1e-5).mean() >= 0.995, f"{(err > 1e-5).mean():.4f} of channels off by more than 1e-5"
    # resolve: mean -> ACES -> NaN scrub (main.rs:1009-1013, image.rs:43-59), and the PPM bytes (ppm.rs:27-31)
    g_img = rt.resolve_host(acc, spp, tonemap=True)
    o_img = orc.resolve(ref, spp, tonemap=True)
    assert np.abs(g_img - o_img).max() <= 1e-4 or (np.abs(g_img - o_img) <= 1e-4).mean() >= 0.995
    gq = orc.quantize_ppm(g_img.astype(np.float64)); oq = orc.quantize_ppm(o_img)
    assert (np.abs(gq.astype(np.int64) - oq.astype(np.int64)) <= 1).mean() >= 0.995
    gs.close()


def test_spp_ranges_compose(gpu_ctx):
    # rendering [0,3) then [3,7) into one accumulator == rendering [0,7): what spp sharding relies on
    hs, _ = _pair("random")
    gs = gpu_ctx.upload(hs)
    w, h = 64, 36
    a, _ = gs.render_host(w, h, 0, 7, 5, rt.INT_MIS, seed=9, samples_per_pass=7)
    b1, _ = gs.render_host(w, h, 0, 3, 5, rt.INT_MIS, seed=9)
    b2, _ = gs.render_host(w, h, 3, 7, 5, rt.INT_MIS, seed=9)
    assert np.allclose(a, b1 + b2, rtol=2e-6, atol=1e-6, equal_nan=True)
    # and the result does not depend on how many samples are in flight per pass
    c, _ = gs.render_host(w, h, 0, 7, 5, rt.INT_MIS, seed=9, samples_per_pass=1)
    assert np.allclose(a, c, rtol=2e-6, atol=1e-6, equal_nan=True)
    gs.close()


@pytest.mark.parametrize("name,integ", [("random", rt.INT_MIS), ("mesh", rt.INT_BRDF), ("cornell", rt.INT_MIS)])
def test_two_render_lanes_give_the_frame_of_one(gpu_ctx, name, integ):
    # rt_render keeps two passes in flight on two streams; each pass still adds its sums to the accumulator after the pass
    # before it, so the frame is bit-identical to running the same passes one after the other (RT_RENDER_ONE_LANE)
    hs, _ = _pair(name)
    gs = gpu_ctx.upload(hs)
    w, h = 96, 54
    for spp, per_pass in ((7, 2), (5, 1), (6, 0), (9, 0)):
        a, sa = gs.render_host(w, h, 1, 1 + spp, 5, integ, seed=77, samples_per_pass=per_pass)
        b, sb = gs.render_host(w, h, 1, 1 + spp, 5, integ, seed=77, samples_per_pass=per_pass, flags=rt.RENDER_ONE_LANE)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert sa.rays == sb.rays and sa.paths == sb.paths == w * h * spp and sa.trace_launches == sb.trace_launches
        a2, _ = gs.render_host(w, h, 1, 1 + spp, 5, integ, seed=77, samples_per_pass=per_pass)
        assert np.array_equal(a.view(np.uint32), a2.view(np.uint32))       # and repeatable
    gs.close()


@pytest.mark.parametrize("name,integ,w,h", [("mesh", rt.INT_MIS, 480, 270), ("mesh", rt.INT_BRDF, 320, 180), ("random", rt.INT_MIS, 200, 112)])
def test_per_lane_walk_gives_the_frame_of_the_wave_walk(gpu_ctx, name, integ, w, h):
    # RT_RENDER_LANE_WALK traces secondary and shadow rays with k_walk (per-lane cursors, vector loads, ray replacement): the same
    # records in the same order with the same filters and exact tests, so the accumulators are bit-identical to k_trace's
    hs, _ = _pair(name)
    gs = gpu_ctx.upload(hs)
    a, sa = gs.render_host(w, h, 0, 6, 5, integ, seed=5)
    b, sb = gs.render_host(w, h, 0, 6, 5, integ, seed=5, flags=rt.RENDER_LANE_WALK)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert sa.rays == sb.rays and sb.walk_run_steps > 0 and sa.walk_run_steps == 0
    gs.close()


@pytest.mark.parametrize("name,integ,w,h,spp", [("cornell", rt.INT_MIS, 480, 270, 16), ("cornell", rt.INT_BRDF, 480, 270, 16), ("mesh", rt.INT_MIS, 480, 270, 6)])
def test_sorted_bounce_queues_give_the_frame_of_unsorted_ones(gpu_ctx, name, integ, w, h, spp):
    # bounce and shadow rays are radix-sorted by a coherence key before they are traced (scenes of >= 64 leaves; from r03x also scenes of a few
    # INSTANCE leaves: cornell's grid is the union of its objects' world boxes without the ground sphere).  RT_RENDER_NO_SORT traces them in the
    # order shading appended them: a path's radiance does not depend on its queue slot, so the accumulators are bit-identical.  One pass of
    # > 2^20 paths, so that cornell's queues stay above the small-queue mode (which never sorts).
    hs, _ = _pair(name)
    gs = gpu_ctx.upload(hs)
    a, sa = gs.render_host(w, h, 0, spp, 5, integ, seed=21, samples_per_pass=spp)
    b, sb = gs.render_host(w, h, 0, spp, 5, integ, seed=21, samples_per_pass=spp, flags=rt.RENDER_NO_SORT)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert sa.rays == sb.rays and sa.paths == sb.paths == w * h * spp
    assert sa.launches > sb.launches      # the sorts ran (and are counted)
    gs.close()


@pytest.mark.parametrize("name,integ,w,h,spp", [("mesh", rt.INT_MIS, 320, 180, 4), ("random", rt.INT_MIS, 200, 112, 8), ("random_blur", rt.INT_BRDF, 160, 90, 4)])
def test_production_hierarchy_gives_the_frame_of_the_reference_tree(gpu_ctx, monkeypatch, name, integ, w, h, spp):
    # The stream the kernel walks holds the reference's leaves in the reference's order under a hierarchy of the library's own (re-bracketed by surface
    # area, inner records most waves enter anyway left out: DESIGN.md section 5).  Every hit and tie is decided at a leaf, so the frame -- and every
    # closest hit -- is the one the reference's own tree gives: RT_REBRACKET=0 RT_ELIDE_AREA=0 (read at upload) walks that tree record for record.
    hs, _ = _pair(name)
    rays = gpu_ctx.upload(hs)
    a, sa = rays.render_host(w, h, 0, spp, 5, integ, seed=33)
    pr = rays.primary_rays(96, 54, 0, 0x5EED)
    pa, ia, ta = rays.trace_closest(pr, 0.001, 1e30)
    rays.close()
    monkeypatch.setenv("RT_REBRACKET", "0"); monkeypatch.setenv("RT_ELIDE_AREA", "0")
    ref = gpu_ctx.upload(hs)
    b, sb = ref.render_host(w, h, 0, spp, 5, integ, seed=33)
    pb, ib, tb = ref.trace_closest(pr, 0.001, 1e30)
    ref.close()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert sa.rays == sb.rays and sa.paths == sb.paths == w * h * spp
    assert np.array_equal(pa, pb) and np.array_equal(ia, ib) and np.array_equal(ta.view(np.uint64), tb.view(np.uint64))
    if name == "mesh":
        assert sa.trace_wave_steps == 0 or sb.trace_wave_steps == 0 or sa.trace_wave_steps < sb.trace_wave_steps


@pytest.mark.parametrize("name,integ,depth", [("random", rt.INT_MIS, 12), ("simple", rt.INT_BRDF, 6), ("sphere", rt.INT_LIGHT_IS, 5)])
def test_material_binned_shading_gives_the_frame_of_queue_order(gpu_ctx, name, integ, depth):
    # opt-in RT_RENDER_MATERIAL_BINS: before every shade launch the queue's slots are regrouped by the material class of their hits
    # (stable partition: ballots + prefix sums, k_bin_count / k_bin_scan / k_bin_scatter).  A path's radiance does not depend on where
    # in the launch it is shaded, so the frame is bit-identical to shading in queue order.
    hs, _ = _pair(name)
    gs = gpu_ctx.upload(hs)
    w, h = 160, 90
    a, sa = gs.render_host(w, h, 0, 5, depth, integ, seed=3, flags=rt.RENDER_MATERIAL_BINS)
    b, sb = gs.render_host(w, h, 0, 5, depth, integ, seed=3)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert sa.rays == sb.rays and sa.launches > sb.launches     # the three binning launches per bounce are counted
    gs.close()


@pytest.mark.parametrize("name,integ", [("random", rt.INT_MIS), ("mesh", rt.INT_BRDF)])
def test_pipelined_jobs_give_the_frame_of_sequential_calls(gpu_ctx, name, integ):
    # rt_render_begin / rt_render_end: two calls in flight on the scene's two lanes; each job adds its sums after the job begun before
    # it, so the accumulator is bit-identical to blocking calls over the same sample ranges with the same pass size
    hs, _ = _pair(name)
    gs = gpu_ctx.upload(hs)
    w, h, chunks = 96, 54, [(0, 3), (3, 7), (7, 8), (8, 13), (13, 15)]
    ref = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda"); acc = torch.zeros_like(ref)
    torch.cuda.synchronize()
    for b, e in chunks:
        gs.render(ref.data_ptr(), w, h, b, e, 5, integ, seed=9, samples_per_pass=e - b, flags=rt.RENDER_ONE_LANE)
    pending, stats = [], []
    for b, e in chunks:
        pending.append(gs.render_begin(acc.data_ptr(), w, h, b, e, 5, integ, seed=9))
        if len(pending) == 2:
            stats.append(gs.render_end(pending.pop(0)))
    with pytest.raises(rt.RtError, match="in flight"):           # the scene takes no other call while jobs are in flight
        gs.render(acc.data_ptr(), w, h, 0, 1, 5, integ)
    while pending:
        stats.append(gs.render_end(pending.pop(0)))
    torch.cuda.synchronize()
    assert torch.equal(ref, acc)
    assert [s.paths for s in stats] == [w * h * (e - b) for b, e in chunks] and all(s.rays > 0 and s.lanes == 1 for s in stats)
    with pytest.raises(rt.RtError, match="empty sample range"):
        gs.render_begin(acc.data_ptr(), w, h, 4, 4, 5, integ)
    a = gs.render_begin(acc.data_ptr(), w, h, 20, 21, 5, integ); b2 = gs.render_begin(acc.data_ptr(), w, h, 21, 22, 5, integ)
    with pytest.raises(rt.RtError, match="two jobs"):
        gs.render_begin(acc.data_ptr(), w, h, 22, 23, 5, integ)
    gs.close()                                                   # closing the scene ends the jobs nobody waited for
    del a, b2


def test_pass_budget_sets_the_default_pass_size(gpu_ctx):
    # samples_per_pass = 0 is a pure function of the parameters and the scene's pass budget (rt_scene_set_pass_budget)
    hs, _ = _pair("random")
    gs = gpu_ctx.upload(hs)
    w, h, depth = 96, 54, 5
    gs.set_pass_budget(3 * w * h * (depth * 96 + 320) + 10)       # room for 3 samples of every pixel per lane
    a, sa = gs.render_host(w, h, 0, 8, depth, rt.INT_MIS, seed=4)
    b, sb = gs.render_host(w, h, 0, 8, depth, rt.INT_MIS, seed=4, samples_per_pass=3)
    assert sa.samples_per_pass == sb.samples_per_pass == 3 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    gs.set_pass_budget(0)                                          # back to the default (80 GiB): half the call per lane
    c, sc_ = gs.render_host(w, h, 0, 8, depth, rt.INT_MIS, seed=4)
    assert sc_.samples_per_pass == 4
    gs.close()


def test_default_pass_size_steps_down_when_the_gpu_is_occupied(gpu_ctx):
    # with samples_per_pass = 0 (what the Rust GpuRenderer::render passes) a call must render on a part-occupied
    # GPU instead of returning RT_ERR_OOM: the budget goes down its ladder (80 -> 40 -> ... -> 2.5 GiB), the pass size with it, and
    # rt_stats.samples_per_pass reports it; the frame equals the one an explicit samples_per_pass of that size gives.
    hs = rt.HostScene.builtin("simple_triangle", bvh_seed=3)
    gs = gpu_ctx.upload(hs)
    w, h, depth, spp = 1920, 1080, 5, 8                            # 1.0 GB of path state per sample of every pixel (a scene without lights: 32-byte bounce records)
    torch.cuda.synchronize()
    free = torch.cuda.mem_get_info()[0]
    hog = torch.empty(free - (2 << 30), dtype=torch.uint8, device="cuda")     # leaves 2 GiB: the default pass (4 spp, 3.6 GB) cannot fit, 2 spp can
    try:
        a, sa = gs.render_host(w, h, 0, spp, depth, rt.INT_MIS, seed=6)
        assert 1 <= sa.samples_per_pass < 4 and sa.paths == w * h * spp
        b, sb = gs.render_host(w, h, 0, spp, depth, rt.INT_MIS, seed=6, samples_per_pass=sa.samples_per_pass)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    finally:
        gs.close()
        del hog
        torch.cuda.empty_cache()


def test_converged_pagoda_mis_independent_seeds(gpu_ctx):
    # the second north-star correctness test on the scene it names: pagoda, MIS (integrator.rs:31-171), at a reduced frame the oracle
    # renders in seconds.  Stated tolerance, as below: rRMSE(gpu, oracle) <= 1.6 * rRMSE(oracle seed B, oracle seed C) + 0.01.
    hs, osc = _pair("mesh")
    gs = gpu_ctx.upload(hs)
    w, h, spp = 64, 36, 48
    g, _ = gs.render_host(w, h, 0, spp, 5, rt.INT_MIS, seed=1001)
    o1, _ = osc.render(w, h, 0, spp, 5, rt.INT_MIS, seed=2002, nthreads=16)
    o2, _ = osc.render(w, h, 0, spp, 5, rt.INT_MIS, seed=3003, nthreads=16)

    def rrmse(a, b):
        a = a / spp; b = b / spp
        return float(np.sqrt(np.mean(((a - b) / (0.5 * (a + b) + 1e-2)) ** 2)))
    assert rrmse(g, o1) <= 1.6 * rrmse(o1, o2) + 0.01, (rrmse(g, o1), rrmse(o1, o2))
    assert abs(g.mean() - o1.mean()) / o1.mean() < 0.03
    gs.close()


def test_converged_images_independent_seeds(gpu_ctx):
    # north-star "converged images": GPU (seed A) vs oracle (seed B) are independent estimates of the same
    # image; per-pixel relative RMSE over the frame must be at the Monte Carlo noise level.
    # Stated tolerance: rRMSE(gpu, oracle) <= 1.6 * rRMSE(oracle seed B, oracle seed C) + 0.01.
    hs, osc = _pair("simple")
    gs = gpu_ctx.upload(hs)
    w, h, spp = 64, 36, 64
    g, _ = gs.render_host(w, h, 0, spp, 5, rt.INT_BRDF, seed=1001)
    o1, _ = osc.render(w, h, 0, spp, 5, rt.INT_BRDF, seed=2002, nthreads=8)
    o2, _ = osc.render(w, h, 0, spp, 5, rt.INT_BRDF, seed=3003, nthreads=8)

    def rrmse(a, b):
        a = a / spp; b = b / spp
        return float(np.sqrt(np.mean(((a - b) / (0.5 * (a + b) + 1e-2)) ** 2)))
    assert rrmse(g, o1) <= 1.6 * rrmse(o1, o2) + 0.01, (rrmse(g, o1), rrmse(o1, o2))
    # and the means agree tightly
    assert abs(g.mean() - o1.mean()) / o1.mean() < 0.02
    gs.close()


def test_furnace_on_gpu(gpu_ctx):
    # analytic check: Lambertian albedo a inside a unit-radiance emissive shell, BRDF integrator -> exactly a
    s = SceneText()
    lit = s.diffuse_light(s.solid(1, 1, 1))
    env = s.sphere((0, 0, 0), -50.0, lit)
    ball = s.sphere((0, 0, 0), 1.0, s.lambertian_rgb(0.25, 0.5, 0.75))
    gs = gpu_ctx.upload(rt.HostScene.from_description(s.finish([env, ball])))
    out = gs.sample_radiance(16, 9, [4 * 16 + 8] * 4, [0, 1, 2, 3], 4, rt.INT_BRDF, 5)
    assert np.allclose(out, [[0.25, 0.5, 0.75]] * 4, rtol=1e-12)
    gs.close()


def test_edge_cases(gpu_ctx):
    hs, _ = _pair("random")
    gs = gpu_ctx.upload(hs)
    # max_depth 0: every sample returns 0 (integrator.rs:33-35); empty spp range: no-op
    a, st = gs.render_host(32, 18, 0, 2, 0, rt.INT_MIS)
    assert not a.any() and st.paths == 32 * 18 * 2 and st.rays == 0
    a, st = gs.render_host(32, 18, 5, 5, 5, rt.INT_MIS)
    assert not a.any() and st.paths == 0
    with pytest.raises(rt.RtError):
        gs.render_host(1, 18, 0, 1, 5, rt.INT_MIS)       # width-1 == 0 divides by zero in main.rs:996
    with pytest.raises(rt.RtError):
        gs.render_host(32, 18, 0, 1, 5, 7)               # unknown integrator
    with pytest.raises(rt.RtError):
        gs.render_host(32, 18, 3, 1, 5, rt.INT_MIS)      # spp_end < spp_begin
    gs.close()
    # a scene with no objects at all: every ray misses -> background
    s = SceneText()
    gs = gpu_ctx.upload(rt.HostScene.from_description(s.finish([], background="sky")))
    out = gs.sample_radiance(16, 9, [0, 50, 143], [0, 0, 0], 5, rt.INT_MIS, 1)
    assert (out > 0.4).all() and (out <= 1.0 + 1e-12).all()
    gs.close()


# BASELINE.json configs[0..4] exactly as bench.py's WORKLOADS runs them (scene, frame, integrator, depth, the bench's default tree)
BASELINE_CONFIGS = [("random", 400, 225, rt.INT_MIS, 50, 4), ("mesh", 1920, 1080, rt.INT_BRDF, 5, 1), ("mesh", 1920, 1080, rt.INT_MIS, 5, 1),
                    ("simple", 1920, 1080, rt.INT_BRDF, 5, 1), ("simple_triangle", 3840, 2160, rt.INT_MIS, 5, 1),
                    ("cornell", 3840, 2160, rt.INT_MIS, 5, 1)]      # configs[4] names both scenes: the broken cube and the instanced red/green box


@pytest.mark.parametrize("name,w,h,integ,depth,spp", BASELINE_CONFIGS, ids=["configs0", "configs1", "configs2", "configs3", "configs4", "configs4_cornell"])
def test_baseline_configs_at_their_full_frame_size(gpu_ctx, name, w, h, integ, depth, spp):
    # every BASELINE.json config at ITS OWN width x height and depth, on the tree the bench renders (min-area axes, seed 3):
    #  (1) 300 samples drawn over the whole frame and the first 1000 spp against the oracle's recursive integrator (1e-9 / signatures, as above);
    #  (2) whole frames: the production kernel and the reference-order counting kernel give bit-identical accumulators, no shortcut contradicted.
    hs = rt.HostScene.builtin(name, bvh_seed=3, bvh_axis="min_area")
    osc = orc.OracleScene(hs.describe())
    gs = gpu_ctx.upload(hs)
    rng = np.random.default_rng(w + depth)
    picks = [(int(rng.integers(w)), int(rng.integers(h)), int(rng.integers(0, 1000))) for _ in range(300)]
    _sample_compare(gs, osc, w, h, picks, depth, integ, seed=0xBA5E)
    a, sa = gs.render_host(w, h, 0, spp, depth, integ, seed=0xBA5E)
    b, sb = gs.render_host(w, h, 0, spp, depth, integ, seed=0xBA5E, flags=rt.RENDER_COUNT_TESTS)
    gs.close()
    sa, sb = sa.as_dict(), sb.as_dict()
    assert sa["paths"] == sb["paths"] == w * h * spp and sa["rays"] == sb["rays"] > sa["paths"]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert sb["filter_violations"] == 0 and sb["box_tests"] + sb["tri_tests"] + sb["sphere_tests"] >= sb["rays"]   # the earth scene has no BVH: one sphere test per ray


@pytest.mark.parametrize("integ,axis", [(rt.INT_BRDF, "seeded"), (rt.INT_MIS, "seeded"), (rt.INT_MIS, "min_area")])
def test_full_size_pruned_walk_equals_reference_order_walk(gpu_ctx, integ, axis):
    # BASELINE.json's full frame (pagoda, 1920x1080, depth 5): the production kernel (fp32 filters, strict pruning, ~1 % of
    # the tests) and the counting kernel (the reference's complete visited set, every test in f64 beside its filter) must
    # produce the SAME accumulator bit for bit -- every ray of every bounce got the same (primitive, t) -- with zero
    # contradicted shortcuts; and rendering the frame twice is deterministic.  12-20 M rays per integrator.
    hs = rt.HostScene.builtin("mesh", bvh_seed=3, bvh_axis=axis)
    gs = gpu_ctx.upload(hs)
    w, h = 1920, 1080
    a, sa = gs.render_host(w, h, 0, 3, 5, integ, seed=0x5EED, flags=rt.RENDER_COUNT_STEPS)
    b, sb = gs.render_host(w, h, 0, 3, 5, integ, seed=0x5EED, flags=rt.RENDER_COUNT_TESTS)
    c, sc_ = gs.render_host(w, h, 0, 3, 5, integ, seed=0x5EED)
    gs.close()
    assert sc_.as_dict()["trace_wave_steps"] == 0 and sa.as_dict()["trace_wave_steps"] > 0      # the step counter (an atomic per wave) runs only on request
    sa, sb = sa.as_dict(), sb.as_dict()
    assert sa["rays"] == sb["rays"] > 10_000_000
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), c.view(np.uint32))
    assert sb["filter_violations"] == 0 and sb["box_tests"] > (4000 if axis == "seeded" else 1500) * sb["rays"]
    assert sa["trace_wave_steps"] * 64 < 0.1 * sb["box_tests"]          # lane-records stepped through: a few percent of the reference's box tests
