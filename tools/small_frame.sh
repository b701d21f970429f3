#!/bin/bash
# configs[0] at its own size (400x225, 10 spp, depth 50, MIS): one blocking frame at a time, frames two in flight, and the 256-spp rate beside them.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r03}; o=gpurun_out/${tag}_c0
B="python3 bench.py --workload random_spheres_400x225_mis --no-cpu-baseline --no-seeded-compare"
show() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2: %.0f Mrays/s  %.2f ms/step  (%d spp/step, %d steps, %d rays)' % (d['value'], d['ms_per_step'], d['config']['spp_per_step'], d['steps'], d['rays']))"; }
timeout -k 10 120 $B --steps 20 --warmup 2 --spp-per-step 10 --job-spp 10 --one-lane > ${o}_10spp_blocking.json 2> ${o}.err || exit 1
show ${o}_10spp_blocking.json "10-spp frames, one at a time (blocking rt_render)"
timeout -k 10 120 $B --steps 20 --warmup 2 --spp-per-step 10 --job-spp 10 > ${o}_10spp_pipelined.json 2>> ${o}.err || exit 1
show ${o}_10spp_pipelined.json "10-spp frames, two in flight"
timeout -k 10 120 $B --steps 4 --warmup 2 --spp-per-step 256 --job-spp 256 > ${o}_256spp.json 2>> ${o}.err || exit 1
show ${o}_256spp.json "256 spp per call, two in flight"
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d ${o}_kt -- $B --steps 2 --warmup 1 --spp-per-step 10 --job-spp 10 --one-lane > /dev/null 2>> ${o}.err || exit 1
python3 tools/kernel_timeline.py ${o}_kt -q > ${o}_10spp_timeline.txt; tail -25 ${o}_10spp_timeline.txt
rm -rf ${o}_kt
