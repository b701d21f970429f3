#!/bin/bash
# one gpurun call: production hierarchy re-bracketed by surface area over the reference's leaf order (RT_REBRACKET=1, default) against the reference's own tree (0)
# usage: tools/ab_rebracket.sh <tag> <workload> [more env settings as VAR=value ...]
tag=$1; wl=$2; shift 2
out=gpurun_out/${tag}_ab_rebracket_${wl}.txt; : > $out
for rep in 1 2; do
  for rb in 0 1; do
    export RT_REBRACKET=$rb
    echo "== RT_REBRACKET=$rb (rep $rep)" >> $out
    timeout -k 10 300 python3 bench.py --workload $wl --steps 6 --warmup 2 --no-cpu-baseline --no-seeded-compare > gpurun_out/${tag}_ab_line.json 2>> $out || exit 1
    python3 - gpurun_out/${tag}_ab_line.json >> $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o = d["roofline"]["one_lane"]
print("  %.0f Mrays/s  %.1f ms per step  checksum %s  one-lane: trace %.1f shade %.1f other %.1f ms  steps/ray %.3f" % (d["value"], d["ms_per_step"], d.get("accum_checksum"), o["trace_ms"], o["shade_ms"], o["other_ms"], d["roofline"]["executed"]["record_steps_per_ray"]))
PY
  done
done
grep -v amdgpu.ids $out
