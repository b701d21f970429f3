#!/bin/bash
# usage: ab_multi.sh libA.so libB.so : several workloads, both libs, one call
for wl in earth_1080p_brdf:50 broken_cube_4k_mis:12 random_spheres_400x225_mis:256 pagoda_1080p_brdf:50 pagoda_1080p_mis:50; do
for rep in 1 2; do
for lib in "$@"; do
  export RT_HIP_LIB=$PWD/raytracing_amd/csrc/$lib
  echo "== ${wl%%:*} $lib"
  timeout -k 10 200 python3 tools/ab_render.py ${wl%%:*} ${wl##*:} 2 0 2>&1 | grep -v amdgpu.ids || exit 1
done
done
done
