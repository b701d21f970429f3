"""Renders the north-star frame once and keeps the evidence: pagoda OBJ, 1920x1080, MIS (integrator.rs:31-171), depth 5, 1000 spp,
ACES, written as PNG (the bytes write_image would print: sqrt gamma, clamp .999, *256) + a stats JSON.
usage: render_target.py <out prefix> [spp] [spp per call]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import raytracing_amd as rt
out = sys.argv[1]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
per_call = int(sys.argv[3]) if len(sys.argv) > 3 else 50
W, H, depth = 1920, 1080, 5
hs = rt.HostScene.builtin("mesh", bvh_seed=3, bvh_axis=os.environ.get("RT_BVH_AXIS", "min_area"))
ctx = rt.Context(0); gs = ctx.upload(hs)
acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
gs.render(acc.data_ptr(), W, H, 0, per_call, depth, rt.INT_MIS)   # warm-up (buffers)
acc.zero_(); torch.cuda.synchronize()
t0 = time.perf_counter(); rays = paths = 0
for b in range(0, spp, per_call):
    st = gs.render(acc.data_ptr(), W, H, b, min(b + per_call, spp), depth, rt.INT_MIS)
    rays += st.rays; paths += st.paths
torch.cuda.synchronize(); dt = time.perf_counter() - t0
px = torch.empty_like(acc)
ctx.resolve(acc.data_ptr(), W, H, spp, True, px.data_ptr()); ctx.synchronize()
img = px.cpu().numpy().astype(np.float64)
b8 = (np.clip(np.sqrt(img), 0.0, 0.999) * 256.0).astype(np.uint8)[::-1]     # ppm.rs:15,27-32: rows bottom-up, sqrt gamma
try:
    from PIL import Image
    Image.fromarray(b8, "RGB").save(out + ".png", optimize=True)
except Exception:
    rt.write_png(out + ".png", img)
a = acc.cpu().numpy()
stats = {"scene": "mesh (pagoda OBJ, Lambertian 0.5, two sphere lights, checker ground)", "width": W, "height": H, "integrator": "mis", "max_depth": depth,
         "spp": spp, "spp_per_call": per_call, "seed": 0x5EED, "bvh_seed": 3, "bvh_axis": os.environ.get("RT_BVH_AXIS", "min_area"), "seconds": dt, "rays": rays, "paths": paths, "mrays_per_s": rays / dt / 1e6,
         "paths_per_s": paths / dt, "mean_radiance": [float(x) for x in a.reshape(-1, 3).mean(0) / spp], "nan_pixels": int(np.isnan(a).any(axis=2).sum()),
         "accum_checksum": float(a.astype(np.float64).sum())}
json.dump(stats, open(out + ".json", "w"), indent=1)
print(json.dumps(stats))
