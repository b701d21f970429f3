"""Assembles the PMC passes of tools/profile_round.sh into one JSON: per-dispatch counters of k_trace / k_shade for ONE
one-lane pass of the bench workload (dispatch order: path rays, shadow rays, path rays, ...), HBM-side traffic
(FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md "HBM"; both in KiB) and issue utilisation.
usage: pmc_collect.py <gpurun_out prefix> <workload> <spp per pass> <stats json of the profiled render> <kernel-trace dir>"""
import os, csv, glob, json, sys, collections

prefix, workload, spp, stats_path, kt_dir = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
st = json.load(open(stats_path))


def dispatches(suffix, kernel):
    """per dispatch (in dispatch order): {counter: value}"""
    by = collections.OrderedDict()
    for f in glob.glob(f"{prefix}_{suffix}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void " + kernel) or r["Kernel_Name"].startswith(kernel):
                by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [by[k] for k in sorted(by)]


def durations(kernel):
    rows = []
    for f in glob.glob(kt_dir + "/**/*kernel_trace.csv", recursive=True):
        rows += [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].replace("void ", "").startswith(kernel)]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]


out = {"workload": workload, "bvh_axis": os.environ.get("RT_BVH_AXIS", "min_area"), "spp_per_step": spp, "render_stats": st,
       "source": "tools/profile_round.sh: one rt_render of one pass on ONE lane per counter pass (counter collection serialises dispatches)"}
for kernel, key in (("k_trace", "k_trace"), ("k_shade", "k_shade")):
    f, w, a, b = dispatches("f", kernel), dispatches("w", kernel), dispatches("a", kernel), dispatches("b", kernel)
    ms = durations(kernel)
    n = min(len(x) for x in (f, w, a, b)) if all((f, w, a, b)) else 0
    rows = []
    for i in range(n):
        gui = b[i].get("GRBM_GUI_ACTIVE", 0.0) / 8.0   # summed over the 8 XCDs
        fetch_b, write_b = 2.0 * f[i]["FETCH_SIZE"] * 1024.0, w[i]["WRITE_SIZE"] * 1024.0
        row = {"launch": i, "ms": ms[i] if i < len(ms) else None, "hbm_bytes": fetch_b + write_b, "fetch_bytes": fetch_b, "write_bytes": write_b,
               "waves": a[i].get("SQ_WAVES"), "valu_insts": a[i].get("SQ_INSTS_VALU"), "salu_insts": a[i].get("SQ_INSTS_SALU"),
               "smem_insts": a[i].get("SQ_INSTS_SMEM"), "branch_insts": a[i].get("SQ_INSTS_BRANCH"),
               "valu_issue_utilisation": a[i].get("SQ_INSTS_VALU", 0.0) * 4 / (1024 * gui) if gui else None,
               "scalar_issue_utilisation": (a[i].get("SQ_INSTS_SALU", 0.0) + a[i].get("SQ_INSTS_SMEM", 0.0) + a[i].get("SQ_INSTS_BRANCH", 0.0)) / (256 * gui) if gui else None,
               "thread_cycles_valu_per_inst": (b[i].get("SQ_THREAD_CYCLES_VALU", 0.0) / max(b[i].get("SQ_ACTIVE_INST_VALU", 1.0), 1.0)) if "SQ_ACTIVE_INST_VALU" in b[i] else None}
        if row["ms"]:
            row["hbm_gbs"] = row["hbm_bytes"] / row["ms"] / 1e6
            row["hbm_frac_of_8TBs"] = row["hbm_gbs"] / 8000.0
        rows.append(row)
    tot_ms = sum(r["ms"] or 0.0 for r in rows)
    tot_b = sum(r["hbm_bytes"] for r in rows)
    out[key] = {"launches": rows, "n": n, "total_ms": tot_ms, "hbm_bytes_total": tot_b,
                "hbm_bytes_per_launch": tot_b / n if n else None,
                "measured_hbm_frac": tot_b / tot_ms / 1e6 / 8000.0 if tot_ms else None}
    gui_all = sum(x.get("GRBM_GUI_ACTIVE", 0.0) for x in b[:n]) / 8.0
    if gui_all:
        out[key]["valu_issue_utilisation"] = sum(x.get("SQ_INSTS_VALU", 0.0) for x in a[:n]) * 4 / (1024 * gui_all)
        out[key]["scalar_issue_utilisation"] = sum(x.get("SQ_INSTS_SALU", 0.0) + x.get("SQ_INSTS_SMEM", 0.0) + x.get("SQ_INSTS_BRANCH", 0.0) for x in a[:n]) / (256 * gui_all)
t = out.get("k_trace", {})
rays = st.get("rays", 0)
out.update({"hbm_bytes_per_launch": t.get("hbm_bytes_per_launch"), "measured_hbm_frac": t.get("measured_hbm_frac"),
            "valu_issue_utilisation": t.get("valu_issue_utilisation"), "scalar_issue_utilisation": t.get("scalar_issue_utilisation"),
            "valu_insts_per_ray": (sum(r["valu_insts"] or 0.0 for r in t.get("launches", [])) / rays) if rays else None,   # wave-level instructions / rays
            "hbm_bytes_per_ray": (t.get("hbm_bytes_total", 0.0) / rays) if rays else None})
print(json.dumps(out, indent=1))
