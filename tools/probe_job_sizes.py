"""Throughput of the headline frame cut into jobs of different sizes, two in flight (one pass per job): probe_job_sizes.py "50,50" "63,63" "100,60" ...
Each pattern is repeated until ~960 spp are rendered; jobs are begun in order on the scene's two lanes (rt_render_begin picks the free one)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracing_amd as rt
W, H, DEPTH = 1920, 1080, 5
hs = rt.HostScene.builtin("mesh", bvh_seed=3, bvh_axis="min_area")
ctx = rt.Context(0)
for pat in sys.argv[1:]:
    sizes = [int(x) for x in pat.split(",")]
    gs = ctx.upload(hs)
    acc = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    def run(total, first):
        pending, rays, b, k = [], 0, first, 0
        while b - first < total:
            n = sizes[k % len(sizes)]; k += 1
            if len(pending) == 2:
                rays += gs.render_end(pending.pop(0)).as_dict()["rays"]
            pending.append(gs.render_begin(acc.data_ptr(), W, H, b, b + n, DEPTH, rt.INT_MIS, seed=0x5EED, samples_per_pass=n)); b += n
        for j in pending:
            rays += gs.render_end(j).as_dict()["rays"]
        return rays, b - first
    try:
        run(2 * sum(sizes), 100000); torch.cuda.synchronize()
        t0 = time.perf_counter(); rays, spp = run(960, 0); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        free, tot = torch.cuda.mem_get_info()
        print(f"jobs {pat:>12}: {rays / dt / 1e6:7.0f} Mrays/s  {dt / spp * 50 * 1e3:6.1f} ms per 50 spp  ({spp} spp; free HBM {free / 2**30:.0f} GiB)", flush=True)
    except Exception as e:
        print(f"jobs {pat:>12}: {type(e).__name__}: {str(e)[:160]}", flush=True)
    gs.close(); del acc; torch.cuda.empty_cache()
ctx.close()
