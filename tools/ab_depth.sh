#!/bin/bash
# per-bounce times of the bench workload (16 spp) for several builds inside one gpurun call
for lib in "$@"; do export RT_HIP_LIB=$PWD/raytracing_amd/csrc/$lib; echo "== $lib"; timeout -k 10 300 python tools/depth_times.py mesh 2 16 2>&1 | grep "^depth" | tail -5; done
