"""Dispatch-by-dispatch view of the LAST render call in a `rocprofv3 --kernel-trace` run: frame_gaps.py <dir> [max rows] [which-th last render call].
Prints start offset, duration, idle gap before each dispatch (per queue), and the busy / idle split of the frame's span."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last k_raygen starts the last frame
starts = [i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"]]
which = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # the which-th render call from the end
fr = rows[starts[-which]:(starts[-which + 1] if which > 1 else len(rows))]
fr = [r for r in fr if not r["Kernel_Name"].startswith("at::")]
t0 = int(fr[0]["Start_Timestamp"])
def short(n):
    n = n.replace("void ", "")
    return (n.split("(")[0] if not "rocprim" in n else "rocprim:" + n.split("::")[-1].split("<")[0])[:44]
end_prev = t0
busy = 0; last_end = t0
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for i, r in enumerate(fr):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - last_end
    if i < limit: print(f"{(s - t0) / 1e3:9.1f} us  {short(r['Kernel_Name']):44s} {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} q{r.get('Queue_Id', '?')}")
    if e > last_end:
        busy += e - max(s, last_end); last_end = e
span = last_end - t0
print(f"frame: {len(fr)} dispatches, span {span / 1e6:.3f} ms, some kernel running {busy / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms")
