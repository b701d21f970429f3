"""Progressive / checkpointed rendering: checkpoint format, interrupt and resume.

CPU part: the host logic with a synthetic render function.  GPU part: the same through rt_render_host, and the
rt_desktop binary (--chunk / --checkpoint / --stop-after / --resume) against an uninterrupted run."""
import os
import subprocess
import sys

import numpy as np
import pytest

import raytracing_amd as rt
from raytracing_amd import progressive as pg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H = 24, 10


def _fake_render(calls=None):
    """Sums of a per-sample image that depends only on the absolute sample index (what Philox keying gives the GPU)."""
    def fn(a, b):
        if calls is not None:
            calls.append((a, b))
        out = np.zeros((H, W, 3), np.float64)
        for s in range(a, b):
            out += np.random.default_rng(1000 + s).random((H, W, 3)) * 3.0
        return out.astype(np.float32)
    return fn


def test_header_matches_the_c_struct(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rt_checkpoint.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(rt_checkpoint_header),'
                   ' offsetof(rt_checkpoint_header, width), offsetof(rt_checkpoint_header, spp_end), offsetof(rt_checkpoint_header, max_depth),'
                   ' offsetof(rt_checkpoint_header, seed), offsetof(rt_checkpoint_header, scene_id)); return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = list(map(int, subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()))
    assert got == [64, 8, 24, 28, 48, 56] and pg.HEADER.size == 64
    packed = pg.HEADER.pack(pg.MAGIC, *range(1, 13))
    assert packed[8:12] == (1).to_bytes(4, "little") and packed[24:28] == (5).to_bytes(4, "little")
    assert packed[28:32] == (6).to_bytes(4, "little") and packed[48:56] == (11).to_bytes(8, "little") and packed[56:64] == (12).to_bytes(8, "little")


def test_interrupted_and_resumed_render_equals_the_uninterrupted_one(tmp_path):
    blob = b"scene bytes"
    path = str(tmp_path / "frame.ckpt")
    ref = pg.ProgressiveRender(_fake_render(), blob, W, H, 3, 20, 5, rt.INT_MIS, 77, chunk=4)
    acc_ref, n_ref = ref.run()
    assert n_ref == 17 and ref.complete
    calls = []
    a = pg.ProgressiveRender(_fake_render(calls), blob, W, H, 3, 20, 5, rt.INT_MIS, 77, chunk=4, path=path)
    _, n = a.run(stop=lambda: a.spp_done >= 8)                          # "ctrl-c" while the second chunk is in flight
    assert n == 8 and not a.complete and calls == [(3, 7), (7, 11)]
    assert not os.path.exists(path + ".tmp")
    h, sums = pg.load_checkpoint(path)
    assert (h["spp_begin"], h["spp_done"], h["spp_end"], h["seed"], h["scene_id"]) == (3, 11, 20, 77, pg.scene_id(blob))
    assert np.array_equal(sums, a.accum)
    calls.clear()
    b = pg.ProgressiveRender.resume(_fake_render(calls), blob, path, chunk=4, expect=dict(width=W, height=H, max_depth=5, integrator=rt.INT_MIS, seed=77, spp_end=20))
    acc, n = b.run()
    assert n == 17 and b.complete and calls == [(11, 15), (15, 19), (19, 20)]
    assert np.array_equal(acc.view(np.uint32), acc_ref.view(np.uint32))                       # bit-identical sums
    assert pg.load_checkpoint(path)[0]["spp_done"] == 20
    # a resumed render that stops between chunk boundaries of ANOTHER chunk size still lands on its own boundaries
    c = pg.ProgressiveRender.resume(_fake_render(calls), blob, path, chunk=4)
    assert c.run()[1] == 17 and c.complete                                                    # nothing left to do
    # resuming a complete render with more samples asked for: refused through `expect`
    with pytest.raises(pg.CheckpointError):
        pg.ProgressiveRender.resume(_fake_render(), blob, path, chunk=4, expect=dict(spp_end=40))


def test_checkpoints_of_other_renders_and_damaged_files_are_refused(tmp_path):
    blob = b"scene A"
    path = str(tmp_path / "a.ckpt")
    a = pg.ProgressiveRender(_fake_render(), blob, W, H, 0, 6, 5, rt.INT_BRDF, 1, chunk=2, path=path)
    a.run(stop=lambda: True)
    with pytest.raises(pg.CheckpointError, match="chunk"):          # another grouping of the same samples would not be bit-identical
        pg.ProgressiveRender.resume(_fake_render(), blob, path, chunk=3)
    with pytest.raises(pg.CheckpointError, match="another scene"):
        pg.ProgressiveRender.resume(_fake_render(), b"scene B", path, chunk=2)
    for k, v in dict(width=W + 1, max_depth=6, integrator=rt.INT_MIS, seed=2).items():
        with pytest.raises(pg.CheckpointError, match=k):
            pg.ProgressiveRender.resume(_fake_render(), blob, path, chunk=2, expect={k: v})
    with pytest.raises(pg.CheckpointError, match="rank"):
        pg.ProgressiveRender.resume(_fake_render(), blob, path, chunk=2, rank=1, world=1)
    raw = open(path, "rb").read()
    bad = tmp_path / "bad.ckpt"
    for data, what in ((raw[:-4], "bytes of sums"), (raw + b"\0" * 4, "bytes of sums"), (b"XXXXXXXX" + raw[8:], "magic"), (raw[:20], "header")):
        bad.write_bytes(data)
        with pytest.raises(pg.CheckpointError, match=what):
            pg.load_checkpoint(str(bad))
    hdr = bytearray(raw); hdr[20:24] = (99).to_bytes(4, "little")        # spp_done beyond spp_end
    bad.write_bytes(bytes(hdr))
    with pytest.raises(pg.CheckpointError, match="out of order"):
        pg.load_checkpoint(str(bad))
    with pytest.raises(ValueError):
        pg.ProgressiveRender(_fake_render(), blob, W, H, 0, 6, 5, rt.INT_BRDF, 1, chunk=0)


def test_ranks_checkpoint_their_own_sample_ranges(tmp_path):
    # spp sharding: rank r renders [r*S/R, (r+1)*S/R); each rank has its own file; the sum of the ranks'
    # sums is the frame whatever rank was interrupted
    blob, path, S, R = b"scene", str(tmp_path / "frame.ckpt"), 12, 3
    whole = _fake_render()(0, S).astype(np.float64)
    total = np.zeros((H, W, 3), np.float64)
    for r in range(R):
        lo, hi = r * S // R, (r + 1) * S // R
        pr = pg.ProgressiveRender(_fake_render(), blob, W, H, lo, hi, 5, rt.INT_MIS, 9, chunk=3, path=path, rank=r, world=R)
        if r == 1:
            pr.run(stop=lambda: True)
            assert pr.spp_done == 3 and os.path.exists(f"{path}.rank1")
            pr = pg.ProgressiveRender.resume(_fake_render(), blob, path, chunk=3, rank=1, world=R)
        acc, n = pr.run()
        assert n == hi - lo
        total += acc
    assert sorted(os.listdir(tmp_path)) == [f"frame.ckpt.rank{r}" for r in range(R)]
    assert np.allclose(total, whole, rtol=1e-5)


@pytest.mark.gpu
def test_gpu_render_resumes_bit_identically(gpu_ctx, tmp_path):
    hs = rt.HostScene.builtin("random", bvh_seed=5)
    gs = gpu_ctx.upload(hs)
    w, h, depth, seed = 96, 54, 5, 31
    fn = lambda a, b: gs.render_host(w, h, a, b, depth, rt.INT_MIS, seed=seed)[0]
    path = str(tmp_path / "g.ckpt")
    ref, _ = pg.ProgressiveRender(fn, hs.blob(), w, h, 0, 10, depth, rt.INT_MIS, seed, chunk=3).run()
    a = pg.ProgressiveRender(fn, hs.blob(), w, h, 0, 10, depth, rt.INT_MIS, seed, chunk=3, path=path)
    a.run(stop=lambda: a.spp_done >= 4)
    assert a.spp_done == 6
    acc, n = pg.ProgressiveRender.resume(fn, hs.blob(), path, chunk=3).run()
    assert n == 10 and np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
    one, _ = gs.render_host(w, h, 0, 10, depth, rt.INT_MIS, seed=seed)                         # one call: same samples, other summation order
    assert np.allclose(acc, one, rtol=2e-6, atol=1e-6, equal_nan=True)
    gs.close()


@pytest.mark.gpu
def test_rt_desktop_interrupt_and_resume(tmp_path):
    exe = os.path.join(ROOT, "raytracing_amd", "host", "rt_desktop")
    assert os.path.exists(exe), "rt_desktop is built by __graft_entry__.build()"
    base = [exe, "--scene", "random", "--width", "160", "--spp", "7", "--depth", "4", "--integrator", "mis", "--assets", os.path.join(ROOT, "assets"), "--chunk", "2"]
    run = lambda extra: subprocess.run(base + extra, check=True, capture_output=True, text=True, timeout=170, cwd=str(tmp_path)).stdout
    run(["--out", "full.ppm", "--checkpoint", "full.ckpt"])
    out = run(["--out", "part.ppm", "--checkpoint", "part.ckpt", "--stop-after", "3"])
    assert "Interrupted after 4 of 7" in out
    h, sums = pg.load_checkpoint(str(tmp_path / "part.ckpt"))
    assert (h["spp_done"], h["spp_end"], h["width"], h["height"]) == (4, 7, 160, 90) and np.isfinite(sums).all() and sums.max() > 0
    partial = (tmp_path / "part.ppm").read_bytes()
    assert partial[:2] == b"P3" and partial != (tmp_path / "full.ppm").read_bytes()           # the partial frame was written (main.rs:970-974)
    out = run(["--out", "part.ppm", "--checkpoint", "part.ckpt", "--resume"])
    assert "Resuming at sample 4 of 7" in out
    assert (tmp_path / "part.ppm").read_bytes() == (tmp_path / "full.ppm").read_bytes()
    hf, sf = pg.load_checkpoint(str(tmp_path / "full.ckpt"))
    hp, sp = pg.load_checkpoint(str(tmp_path / "part.ckpt"))
    assert hf == hp and np.array_equal(sf.view(np.uint32), sp.view(np.uint32))
    # a checkpoint of other settings is refused
    other = list(base); other[other.index("--depth") + 1] = "5"
    r = subprocess.run(other + ["--checkpoint", "part.ckpt", "--resume"], capture_output=True, text=True, timeout=170, cwd=str(tmp_path))
    assert r.returncode != 0 and "another scene or other settings" in r.stderr
