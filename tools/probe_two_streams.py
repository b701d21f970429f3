"""Probe: two contexts (own stream, own workspace) on ONE device driven by two host threads, each rendering its own
sample range into its own accumulator, against one context doing the same total work.  Prints Mrays/s for both."""
import sys, threading, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import raytracing_amd as rt

W, H, DEPTH, SPP, STEPS = 1920, 1080, 5, int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4
NCTX = int(sys.argv[2]) if len(sys.argv) > 2 else 2
hs = rt.HostScene.builtin("mesh", bvh_seed=3)
blob = hs.blob()
torch.cuda.set_device(0)


def run(n_ctx):
    ctxs = [rt.Context(0) for _ in range(n_ctx)]
    scs = [c.upload(blob) for c in ctxs]
    accs = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(n_ctx)]
    rays = [0] * n_ctx

    def work(k, warm):
        for i in range(1 if warm else STEPS * 12 // n_ctx // 4):
            b = (k * 1000 + i) * SPP
            st = scs[k].render(accs[k].data_ptr(), W, H, b, b + SPP, DEPTH, rt.INT_BRDF, seed=0x5EED).as_dict()
            if not warm:
                rays[k] += st["rays"]

    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(k, warm)) for k in range(n_ctx)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    for s in scs: s.close()
    for c in ctxs: c.close()
    return sum(rays) / dt / 1e6, dt


for n in (1, NCTX, 1, NCTX):
    r, dt = run(n)
    print(f"contexts={n} spp/pass={SPP}: {r:.1f} Mrays/s ({dt*1e3:.0f} ms)", flush=True)
