"""N>1 path on CPU: world_size-2 gloo processes run the spp-sharding plumbing of
raytracing_amd/distributed.py (scene-bytes broadcast, disjoint sample ranges, accumulator all-reduce)
with the ORACLE as the per-rank renderer (tests only), and the result must equal one process
rendering every sample."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH, INTEG, SEED = 24, 14, 5, 4, 0, 123


def _worker(rank, world, port, out_dir, progressive=False):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle_ffi as orc
    import raytracing_amd as rt
    from raytracing_amd import distributed as rtd
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    blob, desc = b"", b""
    if rank == 0:
        hs = rt.HostScene.builtin("random", bvh_seed=1)
        blob, desc = hs.blob(), hs.describe()
    blob = rtd.broadcast_blob(blob, dev)
    desc = rtd.broadcast_blob(desc, dev)      # the oracle consumes the description instead of the blob
    osc = orc.OracleScene(desc)

    def render_fn(accum, b, e):
        img, _ = osc.render(W, H, b, e, DEPTH, INTEG, seed=SEED)
        accum += torch.from_numpy(img.astype(np.float32))
    if progressive:
        # each rank renders its own sample range chunk by chunk with a checkpoint file of its own; rank 1 is interrupted after
        # its first chunk and a NEW ProgressiveRender picks the file up; the all-reduce comes last
        from raytracing_amd import progressive as pg
        lo, hi = rtd.spp_range(rank, world, SPP)
        fn = lambda a, b: osc.render(W, H, a, b, DEPTH, INTEG, seed=SEED)[0].astype(np.float32)
        path = os.path.join(out_dir, "frame.ckpt")
        pr = pg.ProgressiveRender(fn, blob, W, H, lo, hi, DEPTH, INTEG, SEED, chunk=1, path=path, rank=rank, world=world)
        pr.run(stop=lambda: rank == 1)
        if rank == 1:
            assert pr.spp_done == 1 and not pr.complete
            pr = pg.ProgressiveRender.resume(fn, blob, path, chunk=1, rank=rank, world=world, expect=dict(seed=SEED, max_depth=DEPTH))
            pr.run()
        assert pr.complete and pr.spp_done == hi - lo
        acc = torch.from_numpy(pr.accum.copy())
        rtd.allreduce_accum(acc)
    else:
        acc = rtd.render_sharded(render_fn, W, H, SPP, dev)
    np.save(os.path.join(out_dir, f"acc{rank}.npy"), acc.numpy())
    np.save(os.path.join(out_dir, f"range{rank}.npy"), np.array(rtd.spp_range(rank, world, SPP)))
    with open(os.path.join(out_dir, f"blob{rank}.bin"), "wb") as f:
        f.write(blob)
    dist.destroy_process_group()


def test_spp_range_partition():
    from raytracing_amd.distributed import spp_range
    for world in (1, 2, 3, 8):
        for spp in (0, 1, 5, 8, 1000):
            r = [spp_range(k, world, spp, 10) for k in range(world)]
            assert r[0][0] == 10 and r[-1][1] == 10 + spp
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
    assert [spp_range(k, 8, 1000) for k in range(8)] == [(125 * k, 125 * k + 125) for k in range(8)]   # BASELINE.json configs[2]


def test_two_rank_gloo_render_equals_single(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a0, a1 = np.load(tmp_path / "acc0.npy"), np.load(tmp_path / "acc1.npy")
    assert np.array_equal(a0, a1)                                   # all-reduce leaves the same sum everywhere
    assert list(np.load(tmp_path / "range0.npy")) == [0, 2] and list(np.load(tmp_path / "range1.npy")) == [2, 5]
    assert open(tmp_path / "blob0.bin", "rb").read() == open(tmp_path / "blob1.bin", "rb").read()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as orc
    import raytracing_amd as rt
    osc = orc.OracleScene(rt.HostScene.builtin("random", bvh_seed=1).describe())
    full, _ = osc.render(W, H, 0, SPP, DEPTH, INTEG, seed=SEED)
    assert np.allclose(a0, full, rtol=1e-6, atol=1e-6)              # equal up to fp32 summation order


def test_two_rank_progressive_render_with_an_interrupted_rank(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    a0, a1 = np.load(tmp_path / "acc0.npy"), np.load(tmp_path / "acc1.npy")
    assert np.array_equal(a0, a1)
    assert sorted(f for f in os.listdir(tmp_path) if "ckpt" in f) == ["frame.ckpt.rank0", "frame.ckpt.rank1"]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as orc
    import raytracing_amd as rt
    from raytracing_amd import progressive as pg
    osc = orc.OracleScene(rt.HostScene.builtin("random", bvh_seed=1).describe())
    full, _ = osc.render(W, H, 0, SPP, DEPTH, INTEG, seed=SEED)
    assert np.allclose(a0, full, rtol=1e-6, atol=1e-6)
    h0, s0 = pg.load_checkpoint(str(tmp_path / "frame.ckpt.rank0")); h1, s1 = pg.load_checkpoint(str(tmp_path / "frame.ckpt.rank1"))
    assert (h0["spp_begin"], h0["spp_done"], h1["spp_begin"], h1["spp_done"], h1["world"]) == (0, 2, 2, 5, 2)
    assert np.allclose(s0.astype(np.float64) + s1, full, rtol=1e-6, atol=1e-6)     # the ranks' files alone reproduce the frame
