#!/bin/bash
# A/B of librt_hip builds over the other bench workloads, inside ONE gpurun call.  usage: tools/ab_workloads.sh libA.so libB.so ...
for wl in random_spheres_400x225_mis earth_1080p_brdf broken_cube_4k_mis pagoda_1080p_mis; do
for lib in "$@"; do
  export RT_HIP_LIB=$PWD/raytracing_amd/csrc/$lib
  spp=4; [ $wl = random_spheres_400x225_mis ] && spp=64; [ $wl = broken_cube_4k_mis ] && spp=2
  echo -n "$wl $lib: "
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --steps 2 --warmup 1 --spp-per-step $spp 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f Mrays/s  %.1f ms/step  trace %.1f shade %.1f other %.1f'%(d['value'], d['ms_per_step'], d['kernel_ms']['trace'], d['kernel_ms']['shade'], d['kernel_ms']['other']))" || echo FAILED
done
done
