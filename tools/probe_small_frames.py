"""How many 10-spp frames of configs[0] (400x225, depth 50) per second does the GPU render when MORE than two are in flight?
Experiment: S scene handles of the same blob on one context, two lanes each, frames begun round-robin (each scene has its own accumulator,
so nothing orders the sums across scenes: a throughput probe, not a renderer).  usage: probe_small_frames.py [frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as rt
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 48
W, H, SPP, DEPTH = 400, 225, 10, 50
hs = rt.HostScene.builtin("random", bvh_seed=3, bvh_axis="min_area")
ctx = rt.Context(0)
for S in (1, 2, 3, 4):
    scenes = [ctx.upload(hs) for _ in range(S)]
    accs = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(S)]
    def run(n, first):
        pending, rays = [], 0
        for f in range(n):
            k = f % S
            if len(pending) == 2 * S:
                g, j = pending.pop(0); rays += g.render_end(j).as_dict()["rays"]
            pending.append((scenes[k], scenes[k].render_begin(accs[k].data_ptr(), W, H, (first + f) * SPP, (first + f + 1) * SPP, DEPTH, rt.INT_MIS, seed=0x5EED)))
        for g, j in pending:
            rays += g.render_end(j).as_dict()["rays"]
        return rays
    run(2 * S, 1000); torch.cuda.synchronize()
    t0 = time.perf_counter(); rays = run(frames, 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{S} scene(s) x 2 lanes = {2 * S} frames in flight: {rays / dt / 1e6:7.0f} Mrays/s  {dt / frames * 1e3:6.2f} ms per 10-spp frame", flush=True)
    for g in scenes: g.close()
ctx.close()
