"""A/B: ray sorting on/off on the bench workload (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as rt
name, integ = (sys.argv[1] if len(sys.argv) > 1 else "mesh"), int(sys.argv[2]) if len(sys.argv) > 2 else rt.INT_BRDF
W, H = (1920, 1080)
ctx = rt.Context(0); gs = ctx.upload(rt.HostScene.builtin(name, bvh_seed=3))
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
for flags in (0, rt.RENDER_NO_SORT, 0, rt.RENDER_NO_SORT):
    acc.zero_(); torch.cuda.synchronize()
    st = gs.render(acc.data_ptr(), W, H, 0, 4, 5, integ, flags=flags).as_dict()
    print("sort" if flags == 0 else "nosort", f"{st['rays']/st['ms_total']/1e3:.2f} Mrays/s total {st['ms_total']:.1f} ms trace {st['ms_trace']:.1f} shade {st['ms_shade']:.1f} other {st['ms_other']:.1f} sum {float(acc.sum()):.6e}")
