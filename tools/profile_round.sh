#!/bin/bash
# usage: tools/profile_round.sh <tag> <spp per pass>: bench line, kernel-trace stats of the same command, PMC passes -> gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; spp=${2:-60}
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_kt -- python3 bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_prof.json 2> gpurun_out/${tag}_kt.log || exit 1
cp "$(ls -t gpurun_out/${tag}_kt/*/*kernel_stats.csv | head -1)" gpurun_out/${tag}_bench_kernel_stats.csv || exit 1
bash tools/pmc_traffic.sh ${tag} $spp > gpurun_out/${tag}_trace_traffic.json || exit 1
bash tools/pmc_render.sh ${tag} $spp 5 > gpurun_out/${tag}_pmc_bench_sq.txt || exit 1
bash tools/pmc_render.sh ${tag}p $spp 1 > gpurun_out/${tag}_pmc_primary_sq.txt || exit 1
