"""GPU half of tools/compare_reference_renders.py: renders, at the resolution of each render the reference
holds under img/, the scene of desktop/src/main.rs that render shows, and writes the frames write_image would print (sqrt gamma,
clamp .999, *256; rows top-down) as PNGs under gpurun_out/ -- the reference's images cannot travel to the GPU box, so the comparison
itself runs afterwards on the CPU.  usage: render_for_compare.py <out dir>"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
import raytracing_amd as rt

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
os.makedirs(out, exist_ok=True)
# (tag, builtin scene, edit of its description text, width, height, integrator, depth, spp, reference file, what is known to differ)
JOBS = [
    ("pagoda_light", "mesh", None, 1024, 576, rt.INT_MIS, 5, 1000, "1000sppPagodaLight.png",
     "mesh_scene() main.rs:789-889 as it stands; albedo.png is absent from the checkout, so the pagoda is Lambertian(0.5) instead of textured"),
    ("pagoda_light_one_light", "mesh", [("lights 2 59404 59405", "lights 1 59404"), ("scene_objects 29705 ", "scene_objects 29704 "), (" 59404 59405 59406", " 59404 59406")],
     1024, 576, rt.INT_MIS, 5, 1000, "1000sppPagodaLight.png",
     "mesh_scene() without its second light sphere at (-40,40,-20) (main.rs:851-859): the stored render shows only the light at (40,40,20), i.e. an earlier version of the scene; albedo as above"),
    ("broken_cube_4k", "simple_triangle", None, 3840, 2160, rt.INT_MIS, 5, 64, "brokencube_2_4k.png",
     "simple_triangle_scene() main.rs:547-633 as it stands HAS a checker ground sphere; the stored render shows none (an older scene version)"),
    ("balls_aces", "random", [("background black", "background sky")], 900, 506, rt.INT_MIS, 50, 256, "balls_aces.png",
     "create_random_scene main.rs:365-494 with the commented skybox (main.rs:492) switched on; the stored render has another camera, aperture and sphere layout "
     "(a pre-Pcg64 version of the scene)"),
    ("globe", "simple_light", None, 1920, 1080, rt.INT_MIS, 5, 256, "globe.png",
     "simple_light() main.rs:636-685 as it stands (earth texture, sphere + rectangle light); the legacy scenes have no `lights` list, so MIS takes its BRDF branch"),
]
ctx = rt.Context(0)
meta = []
for tag, scene, edit, W, H, integ, depth, spp, ref, note in JOBS:
    hs = rt.HostScene.builtin(scene, bvh_seed=3, bvh_axis="min_area")
    if edit:
        text = hs.describe().decode()
        for old, new in edit:
            assert text.count(old) == 1, (tag, old, text.count(old))
            text = text.replace(old, new)
        hs = rt.HostScene.from_description(text)
    gs = ctx.upload(hs)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
    t0 = time.perf_counter(); rays = 0
    per_call = max(1, min(spp, (96 << 20) // (W * H)))
    for b in range(0, spp, per_call):
        rays += gs.render(acc.data_ptr(), W, H, b, min(b + per_call, spp), depth, integ).rays
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    px = torch.empty_like(acc)
    ctx.resolve(acc.data_ptr(), W, H, spp, True, px.data_ptr()); ctx.synchronize()
    img = px.cpu().numpy().astype(np.float64)
    b8 = (np.clip(np.sqrt(img), 0.0, 0.999) * 256.0).astype(np.uint8)[::-1]     # ppm.rs:15,27-32
    Image.fromarray(b8, "RGB").save(os.path.join(out, f"r03_cmp_{tag}.png"), optimize=True)
    # object classes at the pixel centres of sample 0 (what a camera ray hits first): 0 miss, 1 light (DiffuseLight), 2 anything else
    gs.close()
    meta.append({"tag": tag, "scene": scene, "edit": [list(e) for e in edit] if edit else None, "width": W, "height": H, "integrator": rt.INTEGRATORS and [k for k, v in rt.INTEGRATORS.items() if v == integ][0],
                 "max_depth": depth, "spp": spp, "seconds": dt, "mrays_per_s": rays / dt / 1e6, "reference_image": "img/" + ref, "known_differences": note})
    print(tag, f"{dt:.2f} s", flush=True)
json.dump(meta, open(os.path.join(out, "r03_cmp_meta.json"), "w"), indent=1)
