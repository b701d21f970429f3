#!/bin/bash
# per-dispatch timeline of ONE one-lane 9-spp cornell 4K MIS pass, with and without the ray sort (RT_NO_SMALL_SORT=1), plus the diagnostic build's lane use
# usage: tools/cornell_timeline.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; o=gpurun_out/$tag
export RT_W=3840 RT_H=2160
R="python3 tools/render_once.py cornell 0 9 5"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d ${o}_kt_sort -- $R > ${o}_sort_stats.txt 2> ${o}_kt_sort.log || exit 1
python3 tools/kernel_timeline.py ${o}_kt_sort > ${o}_cornell_pass_timeline_sorted.txt || exit 1
export RT_NO_SMALL_SORT=1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d ${o}_kt_nosort -- $R > ${o}_nosort_stats.txt 2> ${o}_kt_nosort.log || exit 1
python3 tools/kernel_timeline.py ${o}_kt_nosort > ${o}_cornell_pass_timeline_unsorted.txt || exit 1
unset RT_NO_SMALL_SORT
if [ -f raytracing_amd/csrc/librt_hip_diag.so ]; then
  RT_HIP_LIB=$PWD/raytracing_amd/csrc/librt_hip_diag.so timeout -k 10 200 $R 2>&1 | grep "^\[diag\]" > ${o}_cornell_diag_sorted.txt
  RT_NO_SMALL_SORT=1 RT_HIP_LIB=$PWD/raytracing_amd/csrc/librt_hip_diag.so timeout -k 10 200 $R 2>&1 | grep "^\[diag\]" > ${o}_cornell_diag_unsorted.txt
fi
echo done
