#!/bin/bash
# A/B of librt_hip builds inside ONE gpurun call (devices differ by ~10 % in clock: never compare across calls).
# usage: tools/ab_lib.sh libA.so libB.so ...   (paths relative to raytracing_amd/csrc)
set -e
for rep in 1 2; do
for lib in "$@"; do
  export RT_HIP_LIB=$PWD/raytracing_amd/csrc/$lib
  echo "== $lib (rep $rep)"
  timeout -k 10 100 python tools/prof_trace.py mesh 1920 1080 3 2>&1 | grep -v amdgpu.ids | tail -1
  timeout -k 10 120 python tools/ab_sort.py mesh 2 2>&1 | grep "^sort" | tail -1
done
done
