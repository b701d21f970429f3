"""The committed bench evidence keeps the shape the driver's contract asks for (no GPU needed): the newest
profiles/rNN_bench.json is a line `python bench.py` printed on the headline workload, its `roofline` is a roofline
(0 < frac <= 1, reproducible from the fields beside it), `cpu_baseline` carries the multithreaded and the one-thread figure,
and the kernel-trace summary committed beside it names the kernel the roofline is about with an agreeing launch duration."""
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    # rNNx_bench.json is a headline line; rNNx_<workload>_bench.json (the other BASELINE configs, tools/measure_configs.sh) is not
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_bench.json")) if re.fullmatch(r"r\d+[a-z]*_bench\.json", os.path.basename(f)))
    assert files, "no committed bench line"
    return files[-1]


def _line(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def test_latest_bench_line_has_the_contract_fields():
    d = _line(_latest())
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert base["metric"].startswith(d["metric"]) and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["vs_baseline"] is None
    assert d["n_gpus"] == 1 and d["value"] > 0 and "model" not in d["config"]
    # the headline: pagoda, 1080p, MIS, 1000 spp (BASELINE.json metric / configs[2])
    c = d["config"]
    assert c["workload"] == "pagoda_1080p_mis" and c["integrator"] == "mis" and (c["width"], c["height"]) == (1920, 1080)
    assert c["spp_total"] == d["steps"] * c["spp_per_step"] == 1000
    assert d["paths"] == 1920 * 1080 * 1000
    # whole-job throughput = rays of the timed steps / time
    assert abs(d["value"] - d["rays"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) < 1e-6 * d["value"]


def test_roofline_is_a_roofline():
    d = _line(_latest())
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert 0.0 < r["frac"] <= 1.0 and 0.0 < r["one_lane"]["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / average launch duration, every factor stated in the line
    want = r["algorithmic_bytes_per_ray"] * r["rays_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9
    assert abs(want - r["achieved"]) < 1e-6 * r["achieved"]
    assert abs(sum(r["algorithmic_bytes_per_ray_parts"].values()) - r["algorithmic_bytes_per_ray"]) < 1e-6
    # measured traffic (PMC) is per launch like `achieved`, and the measured rate cannot beat the peak either
    if r["traffic"] is not None:
        assert 0 < r["traffic"] / (r["one_lane"]["avg_launch_ms"] * 1e-3) / 1e9 <= r["peak"]
        pm = r["issue_pmc"]
        assert 0 < pm["valu_issue_utilisation"] <= 1 and 0 < pm["scalar_issue_utilisation"] <= 1 and 0 < pm["measured_hbm_frac"] <= 1
        if "measured_frac" in r:   # from r03 on: the measured-traffic fraction stands beside the modelled one, and cannot exceed it by much
            assert r["measured_frac"] == pm["measured_hbm_frac"] and r["measured_frac"] <= 1.25 * r["one_lane"]["frac"]
    # the reference-order byte model lives under its own name and is a ratio, not a bandwidth
    assert d["algorithmic_saving"]["ratio"] > 1 and "roofline_valu" not in d


def test_cpu_baseline_has_both_thread_counts():
    d = _line(_latest())
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    one = c["single_thread"]
    assert one["cores"] == 1 and one["value"] > 0 and one["unit"] == d["unit"] and one["sample"]
    assert one["value"] <= c["value"] * 1.05 or c["cores"] == 1


def test_kernel_trace_summary_beside_it_names_the_roofline_kernel():
    path = _latest()
    tag = re.match(r"(r\d+[a-z]*)_bench\.json", os.path.basename(path)).group(1)
    stats = os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats.csv")
    assert os.path.exists(stats), stats
    rows = list(csv.DictReader(open(stats)))
    trace = [r for r in rows if r["Name"].startswith("void k_trace<false")]      # path-ray and shadow-ray instantiations of the production kernel
    assert trace and all(int(r["Calls"]) > 0 for r in trace)
    prof = os.path.join(ROOT, "profiles", f"{tag}_bench_under_rocprof.json")
    assert os.path.exists(prof), prof
    # the line printed by the profiled command itself: its HIP-event launch time must agree with the trace
    p = _line(prof)
    r = p["roofline"]
    calls, total_ms = sum(int(x["Calls"]) for x in trace), sum(float(x["TotalDurationNs"]) for x in trace) / 1e6
    timed = r["launches"]
    jobs = len(p["config"].get("rank0_jobs_spp") or []) or p["steps"]      # from r03w on a job is as large as a lane's pass allows, not one --spp-per-step
    per_step = timed // jobs                                               # launches per job (one pass each): warm-up and the two extra runs are whole jobs
    extra = calls - timed - p["warmup"] * per_step      # the untimed one-lane job, and (from r02h on) the job that counts record steps
    assert extra in (per_step, 2 * per_step), (calls, timed, extra)
    # the counting step's launches run like the timed ones (two lanes inside the call), a few percent slower for their counter: priced at the timed average
    shared_avg = (total_ms - per_step * r["one_lane"]["avg_launch_ms"] - (extra - per_step) * r["avg_launch_ms"]) / (calls - extra)
    assert abs(shared_avg - r["avg_launch_ms"]) < 0.1 * r["avg_launch_ms"], (shared_avg, r["avg_launch_ms"])


def test_every_other_baseline_config_has_a_tracked_line_and_kernel_stats():
    # configs[0], [1], [3], [4] (both scenes of [4]) measured where the driver can see it -- the line `python bench.py --workload X`
    # printed and the `rocprofv3 --kernel-trace --stats` summary of the same command, under profiles/ (tools/measure_configs.sh)
    import sys
    sys.path.insert(0, ROOT)
    import bench
    for wl in sorted(set(bench.WORKLOADS) - {"pagoda_1080p_mis"}):
        lines = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]*_{wl}_bench.json")))
        assert lines, wl
        d = _line(lines[-1])
        scene, _, w, h, integ, depth = bench.WORKLOADS[wl]
        c = d["config"]
        assert (c["workload"], c["scene"], c["width"], c["height"], c["max_depth"]) == (wl, scene, w, h, depth) and c["integrator"] == bench.INTEGRATOR_NAMES[integ]
        assert d["n_gpus"] == 1 and d["value"] > 0 and d["paths"] == w * h * c["spp_total"]
        assert 0 < d["roofline"]["frac"] <= 1 and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["single_thread"]["value"] > 0
        stats = lines[-1].replace("_bench.json", "_kernel_stats.csv")
        assert os.path.exists(stats), stats
        rows = list(csv.DictReader(open(stats)))
        assert any(r["Name"].startswith("void k_trace<false") and int(r["Calls"]) > 0 for r in rows), wl
