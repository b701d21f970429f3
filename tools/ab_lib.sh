#!/bin/bash
# A/B of librt_hip builds inside ONE gpurun call (devices differ by ~10 % in clock: never compare across calls).
# usage: tools/ab_lib.sh <workload> <spp> libA.so libB.so ...   (paths relative to raytracing_amd/csrc)
wl=$1; spp=$2; shift 2
for rep in 1 2; do
for lib in "$@"; do
  export RT_HIP_LIB=$PWD/raytracing_amd/csrc/$lib
  echo "== $lib (rep $rep)"
  timeout -k 10 200 python3 tools/ab_render.py $wl $spp 2 2>&1 | grep -v amdgpu.ids || exit 1
done
done
