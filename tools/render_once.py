"""One rt_render call on the bench scene (profiling target): render_once.py <scene> <integrator> <spp> <depth> [samples_per_pass].
samples_per_pass defaults to spp: ONE pass on one lane, so that per-launch counters are those of a pass of that size."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as rt
name = sys.argv[1] if len(sys.argv) > 1 else "mesh"
integ = int(sys.argv[2]) if len(sys.argv) > 2 else rt.INT_BRDF
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 4
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 5
spass = int(sys.argv[5]) if len(sys.argv) > 5 else spp
W, H = (int(os.environ.get("RT_W", "1920")), int(os.environ.get("RT_H", "1080")))
ctx = rt.Context(0); gs = ctx.upload(rt.HostScene.builtin(name, bvh_seed=3, bvh_axis=os.environ.get("RT_BVH_AXIS", "min_area")))
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
st = gs.render(acc.data_ptr(), W, H, 0, spp, depth, integ, samples_per_pass=spass, flags=int(os.environ.get("RT_RENDER_FLAGS", str(rt.RENDER_SERIAL_SHADOW)))).as_dict()   # one stream: a kernel's elapsed time is its own
print(st)
