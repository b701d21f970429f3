# usage: tools/ab_axis.sh : every bench workload on the seeded-axis tree and on the min-area tree (tools/ab_render.py, RT_BVH_AXIS)
for wl in pagoda_1080p_brdf random_spheres_400x225_mis earth_1080p_brdf broken_cube_4k_mis; do
for ax in seeded min_area; do
  spp=50; [ $wl = random_spheres_400x225_mis ] && spp=256; [ $wl = broken_cube_4k_mis ] && spp=12
  echo "== $wl $ax"
  RT_BVH_AXIS=$ax timeout -k 10 300 python3 tools/ab_render.py $wl $spp 2 0 2>&1 | grep -v amdgpu.ids || exit 1
done
done
