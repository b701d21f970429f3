"""A/B helper: N-spp renders of a bench workload with the library RT_HIP_LIB names; prints Mrays/s per repetition and the
accumulator checksum (equal checksums = same frame).  usage: ab_render.py <workload> <spp> [reps] [flags]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as rt
from bench import WORKLOADS
wl = sys.argv[1] if len(sys.argv) > 1 else "pagoda_1080p_mis"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 50
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 0
scene, legacy, W, H, integ, depth = WORKLOADS[wl]
ctx = rt.Context(0); gs = ctx.upload(rt.HostScene.builtin(scene, bvh_seed=3, legacy_bvh=legacy, bvh_axis=os.environ.get("RT_BVH_AXIS", "min_area")))
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
for r in range(reps + 1):
    acc.zero_(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = gs.render(acc.data_ptr(), W, H, 0, spp, depth, integ, flags=flags).as_dict()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if r: print(f"{wl} {spp} spp: {st['rays'] / dt / 1e6:8.1f} Mrays/s  wall {dt * 1e3:7.1f} ms  trace {st['ms_trace']:7.1f} shade {st['ms_shade']:6.1f} other {st['ms_other']:5.1f}  rays {st['rays']}  sum {float(acc.double().sum()):.9e}", flush=True)
