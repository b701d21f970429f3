"""Trace time of the bench workload by max_depth (diagnostic): the increments are the per-bounce costs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as rt
name, integ = (sys.argv[1] if len(sys.argv) > 1 else "mesh"), int(sys.argv[2]) if len(sys.argv) > 2 else rt.INT_BRDF
W, H = (1920, 1080)
SPP = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = rt.Context(0); gs = ctx.upload(rt.HostScene.builtin(name, bvh_seed=3))
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
prev_t, prev_r = 0.0, 0
for depth in (1, 1, 2, 3, 4, 5):
    acc.zero_(); torch.cuda.synchronize()
    st = gs.render(acc.data_ptr(), W, H, 0, SPP, depth, integ).as_dict()
    dr, dt = st['rays'] - prev_r, st['ms_trace'] - prev_t
    print(f"depth {depth}: rays {st['rays']} trace {st['ms_trace']:.1f} ms total {st['ms_total']:.1f} ms | this bounce: {dr} rays {dt:.1f} ms = {dr/max(dt,1e-9)/1e3:.1f} Mrays/s")
    if depth > 1 or prev_r: prev_t, prev_r = st['ms_trace'], st['rays']
    else: prev_t, prev_r = 0.0, 0
