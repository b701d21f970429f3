#!/bin/bash
# one gpurun call: production stream with inner records left out by the area rule (RT_ELIDE_AREA = q; 0 = only the ground-sphere ancestors, as before r03y)
# usage: tools/ab_elide.sh <tag> <workload> q0 q1 ...
tag=$1; wl=$2; shift 2
out=gpurun_out/${tag}_ab_elide_${wl}.txt; : > $out
for rep in 1 2; do
  for q in "$@"; do
    export RT_ELIDE_AREA=$q
    echo "== RT_ELIDE_AREA=$q (rep $rep)" >> $out
    timeout -k 10 300 python3 bench.py --workload $wl --steps 6 --warmup 1 --no-cpu-baseline --no-seeded-compare > gpurun_out/${tag}_ab_line.json 2>> $out || exit 1
    python3 - gpurun_out/${tag}_ab_line.json >> $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o = d["roofline"]["one_lane"]
print("  %.0f Mrays/s  %.1f ms per step  checksum %s  one-lane: trace %.1f shade %.1f other %.1f ms  steps/ray %.3f" % (d["value"], d["ms_per_step"], d.get("accum_checksum"), o["trace_ms"], o["shade_ms"], o["other_ms"], d["roofline"]["executed"]["record_steps_per_ray"]))
PY
  done
done
grep -v amdgpu.ids $out
