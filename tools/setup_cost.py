"""What does a k_trace wave cost BEFORE and AFTER its walk?  setup_cost.py <mode> [n rays]: traces n primary rays of the pagoda frame through
rt_trace_closest_device (the render loop's kernel) -- mode `full`: as they are; mode `miss`: the same rays with the hit window [t_min, t_max] = [1e30, 2e30], so
every wave fails the first record it meets and ends: its instructions are the set-up (ray, 1/d, certificate, filter constants) and the write-back.
Run each mode under `rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES` (tools/setup_cost.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracing_amd as rt
mode = sys.argv[1] if len(sys.argv) > 1 else "full"
hs = rt.HostScene.builtin("mesh", bvh_seed=3, bvh_axis="min_area")
ctx = rt.Context(0); gs = ctx.upload(hs)
W, H = 1920, 1080
rays = np.concatenate([gs.primary_rays(W, H, s, 0x5EED) for s in range(2)])
n = len(rays)
d_rays = torch.from_numpy(rays).cuda()
d_prim = torch.zeros(n, dtype=torch.int32, device="cuda"); d_t = torch.zeros(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
lo, hi = (0.001, 1e30) if mode == "full" else (1e30, 2e30)
for _ in range(3):
    gs.trace_closest_device(n, d_rays.data_ptr(), lo, hi, d_prim.data_ptr(), d_t.data_ptr())
ctx.synchronize()
print(mode, n, "rays, hits", int((d_prim != -1).sum()))
