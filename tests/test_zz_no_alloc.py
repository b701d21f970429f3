"""rt_trace_closest_device_noalloc (the entry point for callers that cannot allocate) against rt_trace_closest_device:
bit-identical outputs through the C ABI on two calls in a row, the second with a smaller input (the fixed scratch
array keeps the first call's contents)."""
import numpy as np
import pytest
import torch

import raytracing_amd as rt

pytestmark = pytest.mark.gpu


def _both(gs, ctx, rays):
    n = rays.shape[0]
    d = torch.from_numpy(np.ascontiguousarray(rays)).cuda()
    out = []
    for fn in (gs.trace_closest_device, gs.trace_closest_device_noalloc):
        prim = torch.full((n,), 0x7FFFFFFF, dtype=torch.int32, device="cuda")
        t = torch.full((n,), -7.0, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        fn(n, d.data_ptr(), 0.001, 100000.0, prim.data_ptr(), t.data_ptr())
        ctx.synchronize()
        out.append((prim.cpu().numpy().view(np.uint32), t.cpu().numpy().view(np.uint64)))
    return out


def test_noalloc_entry_matches_existing_entry_twice(gpu_ctx):
    hs = rt.HostScene.builtin("mesh", bvh_seed=3)
    gs = gpu_ctx.upload(hs)
    big = gs.primary_rays(480, 270, 0, 0x5EED)          # 129 600 rays
    rng = np.random.default_rng(3)
    small = gs.primary_rays(160, 90, 1, 0x5EED)[::3]      # 4 800 rays, other directions
    small[:, 3:6] *= rng.uniform(0.5, 2.0, (small.shape[0], 1))
    for rays in (big, small):
        (p0, t0), (p1, t1) = _both(gs, gpu_ctx, rays)
        assert (p0 != rt.NONE).sum() > rays.shape[0] // 4
        assert np.array_equal(p0, p1), f"{(p0 != p1).sum()} primitive indices differ"
        assert np.array_equal(t0, t1), f"{(t0 != t1).sum()} t values differ in bits"
    # what the entry point exists for: the call itself takes no device memory (everything it needs is in the fixed array)
    d = torch.from_numpy(np.ascontiguousarray(small)).cuda()
    prim = torch.zeros(small.shape[0], dtype=torch.int32, device="cuda"); t = torch.zeros(small.shape[0], dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    gs.trace_closest_device_noalloc(small.shape[0], d.data_ptr(), 0.001, 100000.0, prim.data_ptr(), t.data_ptr())
    gpu_ctx.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    gs.close()


def test_noalloc_entry_at_its_maximum_size(gpu_ctx):
    # n == RT_NOALLOC_MAX_RAYS fills the scratch array to its last ray block
    hs = rt.HostScene.builtin("random", bvh_seed=3)
    gs = gpu_ctx.upload(hs)
    rays = gs.primary_rays(512, 512, 0, 0x5EED)
    assert rays.shape[0] == rt.NOALLOC_MAX_RAYS
    (p0, t0), (p1, t1) = _both(gs, gpu_ctx, rays)
    assert np.array_equal(p0, p1) and np.array_equal(t0, t1) and (p0 != rt.NONE).sum() > rays.shape[0] // 4
    gs.close()


def test_noalloc_entry_waits_for_render_jobs(gpu_ctx):
    # like its siblings it launches on the context stream (lane 0 of the render jobs): refused while jobs are in flight
    hs = rt.HostScene.builtin("simple_triangle", bvh_seed=3)
    gs = gpu_ctx.upload(hs)
    w, h = 64, 36
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    rays = torch.zeros((8, 7), dtype=torch.float64, device="cuda"); rays[:, 5] = 1.0
    prim = torch.zeros(8, dtype=torch.int32, device="cuda"); t = torch.zeros(8, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    job = gs.render_begin(acc.data_ptr(), w, h, 0, 2, 5, rt.INT_MIS)
    with pytest.raises(rt.RtError, match="in flight"):
        gs.trace_closest_device_noalloc(8, rays.data_ptr(), 0.001, 1e5, prim.data_ptr(), t.data_ptr())
    gs.render_end(job)
    gs.trace_closest_device_noalloc(8, rays.data_ptr(), 0.001, 1e5, prim.data_ptr(), t.data_ptr())
    gpu_ctx.synchronize()
    gs.close()


def test_noalloc_entry_rejects_oversize(gpu_ctx):
    hs = rt.HostScene.builtin("simple_triangle", bvh_seed=3)
    gs = gpu_ctx.upload(hs)
    n = rt.NOALLOC_MAX_RAYS + 1
    rays = torch.zeros((n, 7), dtype=torch.float64, device="cuda")
    prim = torch.zeros(n, dtype=torch.int32, device="cuda"); t = torch.zeros(n, dtype=torch.float64, device="cuda")
    with pytest.raises(rt.RtError, match="rays per call"):
        gs.trace_closest_device_noalloc(n, rays.data_ptr(), 0.001, 1e5, prim.data_ptr(), t.data_ptr())
    gs.close()
