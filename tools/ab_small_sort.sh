#!/bin/bash
# A/B inside ONE gpurun call: ray sorting for scenes of a few INSTANCE leaves (cornell) on / off (RT_NO_SMALL_SORT=1), same library.
# usage: tools/ab_small_sort.sh <tag>
tag=${1:-ab}
export RT_PRINT_SORTGRID=1
out=gpurun_out/${tag}_ab_small_sort.txt
: > $out
for rep in 1 2; do
  for off in 1 0; do
    if [ $off = 1 ]; then export RT_NO_SMALL_SORT=1; else unset RT_NO_SMALL_SORT; fi
    echo "== RT_NO_SMALL_SORT=${RT_NO_SMALL_SORT:-unset} (rep $rep)" >> $out
    timeout -k 10 300 python3 bench.py --workload cornell_4k_mis --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_ab_line.json 2>> $out || exit 1
    python3 - gpurun_out/${tag}_ab_line.json >> $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o = d["roofline"]["one_lane"]
print("  %.0f Mrays/s  %.1f ms per step  checksum %s  one-lane: trace %.1f shade %.1f other %.1f ms" % (d["value"], d["ms_per_step"], d.get("accum_checksum"), o["trace_ms"], o["shade_ms"], o["other_ms"]))
PY
  done
done
cat $out
