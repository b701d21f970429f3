#!/bin/bash
# every BASELINE.json config measured where the driver can see it, one gpurun call.
# usage: tools/measure_configs.sh <tag> [workload ...] ->
#   gpurun_out/<tag>_<workload>_bench.json          the line of `python bench.py --workload <workload>`
#   gpurun_out/<tag>_<workload>_kernel_stats.csv    `rocprofv3 --kernel-trace --stats` summary of the same command
#   gpurun_out/<tag>_<workload>_under_rocprof.json  the line that profiled command printed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
wls=${@:-pagoda_1080p_brdf random_spheres_400x225_mis earth_1080p_brdf broken_cube_4k_mis cornell_4k_mis}
mkdir -p gpurun_out
for wl in $wls; do
  o=gpurun_out/${tag}_${wl}
  extra=""
  # 4K frames: 20 steps x 50 spp = 8.3 G paths; 8 steps x 25 spp keep the call short (rates do not depend on the step count)
  case $wl in broken_cube_4k_mis|cornell_4k_mis) extra="--steps 8 --spp-per-step 25";; esac
  echo "== $wl"
  timeout -k 10 300 python3 bench.py --workload $wl $extra > ${o}_bench.json 2> ${o}_bench.err || { tail -5 ${o}_bench.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d ${o}_kt -- python3 bench.py --workload $wl $extra --no-cpu-baseline --no-seeded-compare > ${o}_under_rocprof.json 2> ${o}_kt.log || { tail -5 ${o}_kt.log; exit 1; }
  cp "$(ls -t ${o}_kt/*/*kernel_stats.csv | head -1)" ${o}_kernel_stats.csv || exit 1
  rm -rf ${o}_kt
  python3 -c "import json; d=json.loads(open('${o}_bench.json').read().strip().splitlines()[-1]); print('%s: %.0f Mrays/s  %.1f ms/step  spp/pass %s  frac %.3f  cpu %.3f' % (d['config']['workload'], d['value'], d['ms_per_step'], d['samples_per_pass'], d['roofline']['frac'], d['cpu_baseline']['value']))"
done
echo done
