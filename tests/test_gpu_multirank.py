"""The N-rank path end to end with the REAL renderer ("R = 1 vs R = 8 image equal to fp32 reassociation error"): bench.py
spawns the ranks, the scene blob is broadcast, every rank renders its contiguous share of the frame's samples on the GPU and the fp32
sums are all-reduced.  The GPU box has one device and RCCL refuses two ranks on one device, so the collectives run over gloo (staged
through host memory, --gloo-gpu) -- everything but the RCCL transport itself is the code the driver's multi-GPU run executes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(gpus, extra=()):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gloo-gpu", "--gpus", str(gpus), "--workload", "random_spheres_400x225_mis", "--steps", "4",
                        "--warmup", "1", "--spp-per-step", "6", "--no-cpu-baseline", "--no-seeded-compare"] + list(extra), env=e, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr[-2000:]
    return json.loads(lines[-1])


def test_three_ranks_render_the_frame_of_one():
    one, three = _bench(1), _bench(3)
    assert three["n_gpus"] == 3 and three["scaling"] == "strong" and three["config"]["per_rank_spp"] == [8, 8, 8] and one["config"]["per_rank_spp"] == [24]
    # Philox counters are keyed on absolute (pixel, sample, level): the ranks trace the same paths as one GPU ...
    assert three["rays"] == one["rays"] and three["paths"] == one["paths"] == 400 * 225 * 24
    # ... and the frame differs only by the order of the fp32 additions (three partial sums added by the all-reduce)
    assert abs(three["accum_checksum"] - one["accum_checksum"]) <= 2e-6 * abs(one["accum_checksum"])
    assert three["allreduce_ms"] > 0 and three["render_ms"]["min"] > 0


def test_weak_scaling_renders_disjoint_frames():
    two = _bench(2, ["--scaling", "weak"])
    assert two["config"]["spp_total"] == 48 and two["paths"] == 400 * 225 * 48
