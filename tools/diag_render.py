"""One render of the bench workload with the diagnostic library (make diag): per-launch traversal statistics on stderr."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as rt
name, integ = (sys.argv[1] if len(sys.argv) > 1 else "mesh"), int(sys.argv[2]) if len(sys.argv) > 2 else rt.INT_BRDF
W, H = (1920, 1080) if name != "random" else (400, 225)
ctx = rt.Context(0); gs = ctx.upload(rt.HostScene.builtin(name, bvh_seed=3, bvh_axis=os.environ.get("RT_BVH_AXIS", "min_area")))
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
st = gs.render(acc.data_ptr(), W, H, 0, int(os.environ.get("RT_DIAG_SPP", "0")) or (4 if name != "random" else 64), 5 if name != "random" else 50, integ, flags=int(os.environ.get("RT_DIAG_FLAGS", "0")),
               samples_per_pass=int(os.environ.get("RT_DIAG_SPP", "0"))).as_dict()
print(st)
