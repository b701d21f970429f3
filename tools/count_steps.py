"""Wave steps of one render (RT_RENDER_COUNT_STEPS): count_steps.py <scene> <integrator> <spp> <depth>.  For A/B of librt_hip builds (RT_HIP_LIB):
the same frame must take the same number of record steps unless the walk itself was changed."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as rt
name = sys.argv[1] if len(sys.argv) > 1 else "mesh"
integ = int(sys.argv[2]) if len(sys.argv) > 2 else rt.INT_MIS
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 8
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 5
W, H = 1920, 1080
ctx = rt.Context(0); gs = ctx.upload(rt.HostScene.builtin(name, bvh_seed=3, bvh_axis=os.environ.get("RT_BVH_AXIS", "min_area")))
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
st = gs.render(acc.data_ptr(), W, H, 0, spp, depth, integ, flags=rt.RENDER_COUNT_STEPS).as_dict()
print(os.path.basename(os.environ.get("RT_HIP_LIB", "librt_hip.so")), "rays", st["rays"], "wave steps", st["trace_wave_steps"], "steps/ray %.4f" % (st["trace_wave_steps"] / st["rays"]), "sum %.9e" % float(acc.double().sum()))
