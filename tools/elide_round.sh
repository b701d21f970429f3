#!/bin/bash
# one gpurun call: step log of one one-lane pass with the diagnostic build, then tools/elide_eval.py over it
# usage: tools/elide_round.sh <tag> [spp]
tag=$1; spp=${2:-71}
d=/tmp/elog_$tag; mkdir -p $d gpurun_out
export RT_HIP_LIB=$PWD/raytracing_amd/csrc/librt_hip_diag.so RT_DIAG_LOG=$d RT_DIAG_LOG_GROUP=16 RT_DIAG_LOG_EVERY=100
timeout -k 10 300 python3 tools/render_once.py mesh 0 $spp 5 > gpurun_out/${tag}_elide_render.txt 2>&1 || { tail -5 gpurun_out/${tag}_elide_render.txt; exit 1; }
unset RT_DIAG_LOG
ls -la $d | head -20
RT_ELIDE_SAVE=gpurun_out/${tag}_elide_counts.npz timeout -k 10 600 python3 tools/elide_eval.py $d > gpurun_out/${tag}_elide_eval.txt 2>&1 || { tail -20 gpurun_out/${tag}_elide_eval.txt; exit 1; }
cat gpurun_out/${tag}_elide_eval.txt
