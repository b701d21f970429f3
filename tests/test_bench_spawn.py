"""`python bench.py --gpus N` must run N ranks: the parent starts them itself when no launcher did,
the line reports the live world size, weak and strong sharding cover the intended absolute sample indices, and a launcher
world that differs from --gpus is an error.  gloo on the CPU with an injected renderer (tests/fake_bench_renderer.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_bench_renderer.py")
W, H = 8, 4


def _run(extra, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--renderer", FAKE, "--steps", "3", "--warmup", "1",
                        "--spp-per-step", "4"] + extra, env=e, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def _sum(samples):
    return float(W * H * 3 * sum(s + 1 for s in samples))


def test_gpus_2_spawns_two_ranks_weak():
    rc, d, err = _run(["--gpus", "2"])
    assert rc == 0, err
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["sharding"] == "spp-range x2" and d["config"]["spp_total"] == 24
    # rank r owns samples [(r*4 + i)*4, ...+4) for steps i = 1..3 (step 0 is the warm-up): both ranks' sums arrive through the all-reduce
    want = _sum(list(range(4, 16)) + list(range(20, 32)))
    assert d["accum_checksum"] == want, (d["accum_checksum"], want)
    assert d["paths"] == 2 * 3 * 4 * W * H and d["rays"] == 2 * d["paths"]
    assert abs(d["value"] - d["rays"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) < 1e-6 * d["value"]


def test_gpus_2_strong_scaling_splits_each_step():
    rc, d, err = _run(["--gpus", "2", "--scaling", "strong"])
    assert rc == 0, err
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["spp_total"] == 12
    assert d["accum_checksum"] == _sum(range(4, 16)) and d["paths"] == 3 * 4 * W * H


def test_single_rank_needs_no_process_group():
    rc, d, err = _run(["--gpus", "1"])
    assert rc == 0, err
    assert d["n_gpus"] == 1 and d["config"]["sharding"] == "none" and d["accum_checksum"] == _sum(range(4, 16))


def test_world_size_must_equal_gpus():
    rc, d, err = _run(["--gpus", "2"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and d is None and "WORLD_SIZE=1" in err
