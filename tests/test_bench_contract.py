"""The committed bench evidence keeps the shape the driver's contract asks for (no GPU needed): the newest
profiles/rNN_bench.json is a line `python bench.py` printed, with `roofline` and `cpu_baseline` objects, and the
kernel-trace summary committed beside it names the kernel the roofline is about."""
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_bench.json")))
    assert files, "no committed bench line"
    return files[-1]


def test_latest_bench_line_has_the_contract_fields():
    path = _latest()
    d = json.loads(open(path).read().strip().splitlines()[-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert base["metric"].startswith(d["metric"]) and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 * max(1.0, r["frac"])
    assert r["traffic"] is None or r["traffic"] > 0
    # achieved = algorithmic bytes per launch / average launch duration, both stated in the line
    want = r["algorithmic_bytes_per_ray"] * r["rays_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
    assert abs(want - r["achieved"]) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    # whole-job throughput = rays of the timed steps / time
    assert abs(d["value"] - d["rays"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) < 1e-6 * d["value"]


def test_kernel_trace_summary_beside_it_names_the_roofline_kernel():
    path = _latest()
    tag = re.match(r"(r\d+[a-z]*)_bench\.json", os.path.basename(path)).group(1)
    stats = os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats.csv")
    assert os.path.exists(stats), stats
    rows = list(csv.DictReader(open(stats)))
    trace = [r for r in rows if r["Name"].startswith("void k_trace<false")]
    assert len(trace) == 1 and int(trace[0]["Calls"]) > 0
    d = json.loads(open(path).read().strip().splitlines()[-1])
    prof = os.path.join(ROOT, "profiles", f"{tag}_bench_under_rocprof.json")
    if os.path.exists(prof):   # the line printed by the profiled command itself: its HIP-event launch time must agree with the trace
        p = json.loads(open(prof).read().strip().splitlines()[-1])["roofline"]
        calls, total_ms = int(trace[0]["Calls"]), float(trace[0]["TotalDurationNs"]) / 1e6
        timed = d["steps"] * round(p["rays_per_launch"] and (d["rays"] / d["steps"]) / p["rays_per_launch"])     # launches in the timed region
        extra = calls - timed - d["warmup"] * (timed // d["steps"])                                            # the untimed one-lane step
        assert extra >= 0
        shared_avg = (total_ms - extra * p.get("avg_launch_ms_one_lane", 0.0)) / (calls - extra)
        assert abs(shared_avg - p["avg_launch_ms"]) < 0.15 * p["avg_launch_ms"], (shared_avg, p["avg_launch_ms"])
