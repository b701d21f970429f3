#!/bin/bash
# usage: tools/ab_scene.sh <scene> <integrator> libA.so libB.so ... : 1080p 4 spp depth 5 render of a builtin scene per build, one gpurun call
scene=$1; integ=$2; shift 2
for lib in "$@"; do export RT_HIP_LIB=$PWD/raytracing_amd/csrc/$lib; echo -n "$scene $lib: "; timeout -k 10 200 python tools/ab_sort.py $scene $integ 2>&1 | grep "^sort" | tail -1; done
