"""Copies one profiling round (tools/profile_round.sh <tag>) from gpurun_out/ into profiles/ and refreshes profiles/trace_pmc_latest.json,
the PMC summary bench.py quotes in `roofline.traffic` / `roofline.issue_pmc`.  usage: publish_profiles.py <tag>"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for suffix in ("bench.json", "bench_under_rocprof.json", "bench_kernel_stats.csv", "pass_timeline.txt", "pmc.json", "diag.txt"):
    f = os.path.join(src, f"{tag}_{suffix}")
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(dst, f"{tag}_{suffix}"))
pm = json.load(open(os.path.join(src, f"{tag}_pmc.json")))
latest = {k: pm[k] for k in ("workload", "spp_per_step", "hbm_bytes_per_launch", "measured_hbm_frac", "valu_issue_utilisation", "scalar_issue_utilisation",
                             "valu_insts_per_ray", "hbm_bytes_per_ray")}
latest["k_trace_launches"] = pm["k_trace"]["n"]
latest["bvh_axis"] = pm.get("bvh_axis", "seeded")
latest["source"] = f"profiles/{tag}_pmc.json (rocprofv3 --pmc, one pass of {pm['spp_per_step']} spp on one lane: FETCH_SIZE x2 + WRITE_SIZE, SQ_INSTS_*, GRBM_GUI_ACTIVE)"
diag = os.path.join(src, f"{tag}_diag.txt")
if os.path.exists(diag):
    use = [{"rays": int(m.group(1)), "steps_per_wave": float(m.group(2)), "lane_use": float(m.group(3)) / 100.0}
           for m in re.finditer(r"rays (\d+) waves \d+\s+steps/wave (\d+).*?lane use ([\d.]+)%", open(diag).read())]
    if use:
        latest["lane_use_by_launch"] = use
json.dump(latest, open(os.path.join(dst, "trace_pmc_latest.json"), "w"), indent=1)
print(json.dumps(latest)[:600])
