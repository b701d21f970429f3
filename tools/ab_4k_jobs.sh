#!/bin/bash
# one gpurun call: the 4K configs with jobs of one 25-spp step (three passes of 9 / 8 / 8) against jobs that are ONE pass each (16-17 spp)
# usage: tools/ab_4k_jobs.sh <tag>
tag=${1:-ab}; out=gpurun_out/${tag}_ab_4k_jobs.txt; : > $out
for wl in broken_cube_4k_mis cornell_4k_mis; do
  for js in 25 16 12; do
    echo "== $wl --job-spp $js" >> $out
    timeout -k 10 300 python3 bench.py --workload $wl --job-spp $js --no-cpu-baseline --no-seeded-compare > gpurun_out/${tag}_ab_line.json 2>> $out || exit 1
    python3 - gpurun_out/${tag}_ab_line.json >> $out <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("  %.0f Mrays/s  %.1f ms per step  jobs %s  spp/pass %s  checksum %s" % (d["value"], d["ms_per_step"], d["config"]["rank0_jobs_spp"], d.get("samples_per_pass"), d.get("accum_checksum")))
PY
  done
done
grep -v amdgpu.ids $out
