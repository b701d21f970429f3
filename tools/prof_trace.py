"""Times the traversal kernel alone on resident primary rays (diagnostic; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracing_amd as rt

name = sys.argv[1] if len(sys.argv) > 1 else "mesh"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 480
h = int(sys.argv[3]) if len(sys.argv) > 3 else 270
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ctx = rt.Context(0)
hs = rt.HostScene.builtin(name, bvh_seed=3)
gs = ctx.upload(hs)
rays = torch.from_numpy(gs.primary_rays(w, h, 0, 7)).cuda()
n = w * h
prim = torch.empty(n, dtype=torch.int32, device="cuda")
t = torch.empty(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for r in range(reps):
    t0 = time.time()
    gs.trace_closest_device(n, rays.data_ptr(), 0.001, 1e5, prim.data_ptr(), t.data_ptr())
    ctx.synchronize()
    dt = time.time() - t0
    print(f"{name} {w}x{h}: {n} rays in {dt*1e3:.2f} ms = {n/dt/1e6:.3f} Mrays/s, hits {(prim != -1).sum().item()}")
