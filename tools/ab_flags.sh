# usage: tools/ab_flags.sh <workload> <spp> <flags>... : the same library with different rt_render flags, in one call
wl=$1; spp=$2; shift 2
for rep in 1 2; do for fl in "$@"; do
  echo "== flags $fl (rep $rep)"; timeout -k 10 200 python3 tools/ab_render.py $wl $spp 2 $fl 2>&1 | grep -v amdgpu.ids || exit 1
done; done
