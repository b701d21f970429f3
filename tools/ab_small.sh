#!/bin/bash
# A/B of run_paths' small-queue knobs (env, read once per process) inside ONE gpurun call: configs[0] frames of 10 spp, blocking and two in flight.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --workload random_spheres_400x225_mis --no-cpu-baseline --no-seeded-compare --steps 20 --warmup 2 --spp-per-step 10 --job-spp 10"
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f Mrays/s %.2f ms/frame checksum %.6f' % (d['value'], d['ms_per_step'], d['accum_checksum']))"; }
for cfg in "RT_TUNE_MIN_LANES=1 RT_TUNE_SMALL_WAVES=8192" "RT_TUNE_MIN_LANES=1 RT_TUNE_SMALL_WAVES=8192 RT_TUNE_SMALL_QUEUE=524288" "RT_TUNE_MIN_LANES=1 RT_TUNE_SMALL_WAVES=8192 RT_TUNE_SMALL_QUEUE=1048576" "RT_TUNE_MIN_LANES=1 RT_TUNE_SMALL_WAVES=8192 RT_TUNE_SMALL_QUEUE=4194304" "RT_TUNE_MIN_LANES=1 RT_TUNE_SMALL_WAVES=16384 RT_TUNE_SMALL_QUEUE=1048576" "RT_TUNE_MIN_LANES=2 RT_TUNE_SMALL_WAVES=8192 RT_TUNE_SMALL_QUEUE=1048576" "RT_TUNE_MIN_LANES=1 RT_TUNE_SMALL_WAVES=8192 RT_TUNE_SMALL_QUEUE=1048576 RT_TUNE_SHADOW_SIDE=0"; do
  echo -n "$cfg | blocking: "; env $cfg timeout -k 10 100 $B --one-lane 2>/dev/null | val || exit 1
  echo -n "$cfg | two in flight: "; env $cfg timeout -k 10 100 $B 2>/dev/null | val || exit 1
done
P="python3 bench.py --no-cpu-baseline --no-seeded-compare --steps 8 --warmup 2"
for cfg in "RT_TUNE_SHADOW_SIDE=0" "RT_TUNE_SHADOW_SIDE=1" "RT_TUNE_SHADOW_SIDE=0" "RT_TUNE_SHADOW_SIDE=1"; do
  echo -n "pagoda $cfg | "; env $cfg timeout -k 10 200 $P 2>/dev/null | val || exit 1
done
for cfg in "RT_TUNE_SHADOW_SIDE=0" "RT_TUNE_SHADOW_SIDE=1"; do
  echo -n "pagoda one-lane $cfg | "; env $cfg timeout -k 10 200 $P --one-lane 2>/dev/null | val || exit 1
done
