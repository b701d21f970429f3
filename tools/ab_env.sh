#!/bin/bash
# A/B of environment knobs (RT_TUNE_*) on one bench workload inside ONE gpurun call.  usage: tools/ab_env.sh <workload> "<bench args>" "ENV=.. ENV=.." "ENV=.." ...
wl=$1; args=$2; shift 2
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms',{}); print('%.0f Mrays/s %.2f ms/step | trace %.0f shade %.0f other %.0f tail %.0f | checksum %.6f' % (d['value'], d['ms_per_step'], k.get('trace',0), k.get('shade',0), k.get('other',0), k.get('tail',0), d['accum_checksum']))"; }
for rep in 1 2; do
for cfg in "$@"; do
  echo -n "$wl [$cfg] rep $rep: "; env $cfg timeout -k 10 300 python3 bench.py --workload $wl --no-cpu-baseline --no-seeded-compare $args 2>/dev/null | val || exit 1
done
done
