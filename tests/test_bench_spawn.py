"""`python bench.py --gpus N` must run N ranks and render THE frame: the parent starts the ranks itself when
no launcher did and fails fast when one of them dies; the line reports the live world size; strong sharding (the default) splits the
steps x spp_per_step samples of the frame into contiguous per-rank ranges rendered as jobs of >= spp_per_step samples
(tonemap after the sum, main.rs:1009-1011), weak sharding gives every rank a frame of its own; a launcher world that
differs from --gpus is an error.  gloo on the CPU with an injected renderer (tests/fake_bench_renderer.py)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_bench_renderer.py")
W, H = 8, 4
sys.path.insert(0, ROOT)


def _run(extra, env=None, shape=("--steps", "3", "--warmup", "1", "--spp-per-step", "4")):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--renderer", FAKE] + list(shape) + extra,
                       env=e, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def _sum(samples):
    return float(W * H * 3 * sum(s + 1 for s in samples))


def test_plan_jobs_keeps_jobs_at_least_a_step_large():
    import bench
    assert bench.plan_jobs(0, 1000, 50) == [(50 * i, 50 * i + 50) for i in range(20)]
    assert bench.plan_jobs(125, 250, 50) == [(125, 187), (187, 250)]          # 125 spp on 8 ranks: two jobs of 62 and 63, not 6-spp slivers
    assert bench.plan_jobs(0, 30, 50) == [(0, 30)] and bench.plan_jobs(7, 7, 50) == []
    # with a memory-bound job size (72 spp at 1080p): the fewest jobs that fit, an even number of them, never below a step
    big = bench.plan_jobs(0, 1000, 50, 72)
    assert len(big) == 14 and {e - b for b, e in big} == {71, 72} and big[0][0] == 0 and big[-1][1] == 1000
    assert bench.plan_jobs(125, 250, 50, 72) == [(125, 187), (187, 250)]      # 8 ranks: unchanged
    assert [e - b for b, e in bench.plan_jobs(0, 500, 50, 72)] == [62, 63, 62, 63, 62, 63, 62, 63]   # 7 jobs of 72 would leave one alone: 8 of 62-63
    assert bench.plan_jobs(0, 60, 50, 72) == [(0, 60)] and bench.plan_jobs(0, 100, 50, 72) == [(0, 50), (50, 100)]
    # a 4K frame: a 25-spp step does not fit one pass (18 spp): jobs of ONE pass each, an even number, instead of jobs of three short passes
    assert [e - b for b, e in bench.plan_jobs(0, 200, 25, 18)] == [16, 17, 17, 16, 17, 17, 16, 17, 17, 16, 17, 17]
    assert [e - b for b, e in bench.plan_jobs(0, 25, 25, 18)] == [12, 13] and bench.plan_jobs(0, 3, 25, 1) == [(0, 1), (1, 2), (2, 3)]
    assert bench.plan_jobs(0, 1000, 50, 50) == bench.plan_jobs(0, 1000, 50)
    for world in (1, 2, 3, 4, 8, 16):                                        # the ranks' shares tile the frame exactly
        cuts = [bench.rank_range("strong", r, world, 1000) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == 1000 and all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        for b, e in cuts:
            js = bench.plan_jobs(b, e, 50)
            assert js[0][0] == b and js[-1][1] == e and all(js[i][1] == js[i + 1][0] for i in range(len(js) - 1))
            assert all(y - x >= 50 for x, y in js) and all(y - x < 100 for x, y in js)


def test_gpus_2_default_is_the_frame_split_over_the_ranks():
    rc, d, err = _run(["--gpus", "2"])
    assert rc == 0, err
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["spp_total"] == 12 and d["config"]["per_rank_spp"] == [6, 6]
    assert d["config"]["sharding"].startswith("spp-range x2")
    assert d["accum_checksum"] == _sum(range(12)) and d["paths"] == 3 * 4 * W * H and d["rays"] == 2 * d["paths"]
    assert abs(d["value"] - d["rays"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) < 1e-6 * d["value"]
    assert d["allreduce_ms"] >= 0 and d["render_ms"]["max"] >= d["render_ms"]["min"] > 0 and d["per_rank_mrays_s"]["min"] > 0


def test_gpus_2_weak_gives_every_rank_a_frame_of_its_own():
    rc, d, err = _run(["--gpus", "2", "--scaling", "weak"])
    assert rc == 0, err
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["spp_total"] == 24
    assert d["accum_checksum"] == _sum(range(24)) and d["paths"] == 2 * 3 * 4 * W * H


def test_gpus_8_renders_the_one_rank_frame_in_jobs_of_62_and_63(tmp_path):
    shape = ("--steps", "20", "--warmup", "2", "--spp-per-step", "50")
    rc, d, err = _run(["--gpus", "8"], env={"FAKE_BENCH_LOG": str(tmp_path)}, shape=shape)
    assert rc == 0, err
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["spp_total"] == 1000 and d["config"]["per_rank_spp"] == [125] * 8
    assert d["config"]["rank0_jobs_spp"] == [62, 63]
    rc1, d1, err1 = _run(["--gpus", "1"], shape=shape)
    assert rc1 == 0, err1
    assert d["accum_checksum"] == d1["accum_checksum"] == _sum(range(1000)) and d["paths"] == d1["paths"] == 1000 * W * H
    for r in range(8):   # rank r: two warm-up jobs past the frame, then [125r, 125r + 62), [125r + 62, 125r + 125)
        got = [tuple(x) for x in json.load(open(tmp_path / f"rank{r}.json"))]
        assert got[2:] == [(125 * r, 125 * r + 62), (125 * r + 62, 125 * r + 125)], (r, got)
        assert len(got) == 4 and all(b >= 8000 and e - b == 63 for b, e in got[:2]), (r, got)


def test_a_rank_that_dies_ends_the_run_at_once():
    t0 = time.time()
    rc, d, err = _run(["--gpus", "4"], env={"FAKE_BENCH_FAIL_RANK": "3"})
    took = time.time() - t0
    assert rc != 0 and d is None
    assert "rank 3 exited with code" in err and "fails on purpose" in err, err      # its stderr is kept
    assert took < 90, took   # the three ranks left in the all-reduce are terminated, nobody waits for a collective time limit


def test_single_rank_needs_no_process_group():
    rc, d, err = _run(["--gpus", "1"])
    assert rc == 0, err
    assert d["n_gpus"] == 1 and d["config"]["sharding"] == "none" and d["accum_checksum"] == _sum(range(12)) and d["allreduce_ms"] >= 0


def test_rehearsal_of_one_rank_renders_that_ranks_share():
    rc, d, err = _run(["--gpus", "1", "--rehearse-rank", "1/2"])
    assert rc == 0, err
    assert d["rehearsal"]["rank"] == 1 and d["config"]["spp_total"] == 6 and d["accum_checksum"] == _sum(range(6, 12))


def test_world_size_must_equal_gpus():
    rc, d, err = _run(["--gpus", "2"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and d is None and "WORLD_SIZE=1" in err


def test_under_the_drivers_launcher_every_rank_is_one_of_its_ranks():
    """The driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--renderer", FAKE, "--gpus", "2", "--steps", "3", "--warmup", "1", "--spp-per-step", "4"],
                       env=e, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, p.stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["accum_checksum"] == _sum(range(12))
