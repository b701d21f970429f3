#!/bin/bash
# usage: tools/profile_round.sh <tag> [workload] [integrator id] : everything the round's profiles/ holds, in one gpurun call ->
#   gpurun_out/<tag>_bench.json               the line of `python bench.py`
#   gpurun_out/<tag>_bench_under_rocprof.json the line printed under `rocprofv3 --kernel-trace --stats`, <tag>_bench_kernel_stats.csv its summary
#   gpurun_out/<tag>_pass_timeline.txt        per-dispatch durations of ONE one-lane pass (kernel trace of tools/render_once.py)
#   gpurun_out/<tag>_pmc.json                 PMC passes over that same single pass (FETCH_SIZE, WRITE_SIZE, SQ a, SQ/GRBM b: one run each)
#   gpurun_out/<tag>_diag.txt                 lane use / steps per wave per launch from the diagnostic build (if librt_hip_diag.so exists)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; wl=${2:-pagoda_1080p_mis}; integ=${3:-0}
o=gpurun_out/$tag
mkdir -p gpurun_out
timeout -k 10 400 python3 bench.py --workload $wl > ${o}_bench.json 2> ${o}_bench.err || exit 1
spass=$(python3 -c "import json,sys; print(json.loads(open('${o}_bench.json').read().strip().splitlines()[-1])['samples_per_pass'])") || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d ${o}_kt -- python3 bench.py --workload $wl --no-cpu-baseline --no-seeded-compare > ${o}_bench_under_rocprof.json 2> ${o}_kt.log || exit 1
cp "$(ls -t ${o}_kt/*/*kernel_stats.csv | head -1)" ${o}_bench_kernel_stats.csv || exit 1
R="python3 tools/render_once.py mesh $integ $spass 5"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d ${o}_kt1 -- $R > ${o}_pass_stats.txt 2> ${o}_kt1.log || exit 1
python3 tools/kernel_timeline.py ${o}_kt1 > ${o}_pass_timeline.txt || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d ${o}_f -- $R > ${o}_f.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d ${o}_w -- $R > ${o}_w.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH --output-format csv -d ${o}_a -- $R > ${o}_a.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d ${o}_b -- $R > ${o}_b.log 2>&1 || exit 1
python3 -c "import ast,json,sys; print(json.dumps(ast.literal_eval(open('${o}_pass_stats.txt').read().strip().splitlines()[-1])))" > ${o}_pass_stats.json || exit 1
python3 tools/pmc_collect.py $o $wl $spass ${o}_pass_stats.json ${o}_kt1 > ${o}_pmc.json || exit 1
if [ -f raytracing_amd/csrc/librt_hip_diag.so ]; then
  RT_HIP_LIB=$PWD/raytracing_amd/csrc/librt_hip_diag.so timeout -k 10 200 $R 2>&1 | grep "^\[diag\]" > ${o}_diag.txt
fi
echo done
