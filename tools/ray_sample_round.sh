#!/bin/bash
# one gpurun call: ray log of one one-lane pass with the diagnostic build -> gpurun_out/<tag>_ray_sample.npz
tag=$1; spp=${2:-71}
d=/tmp/rlog_$tag; mkdir -p $d gpurun_out
export RT_HIP_LIB=$PWD/raytracing_amd/csrc/librt_hip_diag.so RT_DIAG_LOG=$d RT_DIAG_LOG_GROUP=4 RT_DIAG_LOG_EVERY=400
timeout -k 10 300 python3 tools/render_once.py mesh 0 $spp 5 > gpurun_out/${tag}_ray_render.txt 2>&1 || { tail -5 gpurun_out/${tag}_ray_render.txt; exit 1; }
unset RT_DIAG_LOG
timeout -k 10 300 python3 tools/save_ray_sample.py $d gpurun_out/${tag}_ray_sample.npz 300
