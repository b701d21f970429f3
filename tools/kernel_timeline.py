"""Per-dispatch table of a `rocprofv3 --kernel-trace` run: kernel_timeline.py <dir> -> order, short kernel name, duration (ms), and totals by kernel."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tot = collections.defaultdict(lambda: [0, 0.0])
def short(n):
    n = n.replace("void ", "")
    for k in ("k_trace", "k_shade", "k_raygen", "k_unwind", "k_accumulate", "k_iota", "k_set_u32", "k_bin", "k_scatter", "k_hist"):
        if n.startswith(k): return n.split("(")[0][:40]
    return ("rocprim:" + n.split("::")[-1].split("<")[0])[:40] if "rocprim" in n else n[:40]
quiet = len(sys.argv) > 2 and sys.argv[2] == "-q"
for i, r in enumerate(rows):
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    s = short(r["Kernel_Name"])
    tot[s][0] += 1; tot[s][1] += ms
    if not quiet and ms >= 0.05: print(f"{i:5d} {s:40s} {ms:9.3f} ms  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}")
print("---- totals")
all_ms = sum(v[1] for v in tot.values())
for s, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]): print(f"{s:40s} {n:6d} calls {ms:10.3f} ms  {100 * ms / all_ms:5.1f} %")
if rows: print(f"span {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6:.3f} ms, busy sum {all_ms:.3f} ms")
