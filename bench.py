#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline: Mrays/s (+ paths/s) of the pagoda OBJ scene, 1920x1080, MIS integrator
(`MultipleImportanceSampleIntegrator`, integrator.rs:31-171 -- the one the reference's main() selects), depth 5
(main.rs:944), 1000 spp = configs[2], which fits one GPU.

A "step" is one render call (ray-gen -> [traverse -> shade -> traverse shadow rays] x depth -> unwind -> accumulate)
over --spp-per-step samples of every pixel; the default 50 makes the driver's `--steps 20` exactly the 1000-spp frame.
Steps are queued with rt_render_begin and waited for with rt_render_end, two in flight (one 50-spp pass each, on the scene's
two lanes): the thin deep bounces of one step overlap the next step's kernels; --sync-steps runs blocking rt_render calls
(two lanes of 25-spp passes inside each call) instead.  Inputs (the flattened scene) are resident in HBM before the timed region;
accumulators stay on the device.  N ranks: `--scaling weak` (default) gives every rank its own disjoint range of
steps x spp_per_step absolute sample indices; `--scaling strong` splits each step's samples across the ranks
(distributed.spp_range).  One broadcast of the scene bytes before, one all-reduce of the fp32 sums inside the timed region.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child processes,
before this process touches the GPU); under `torch.distributed.run` it is one of the ranks.  n_gpus in the line is the
live world size and must equal --gpus.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import importlib.util
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (scene, legacy_bvh, width, height, integrator id, max_depth)
    "pagoda_1080p_mis": ("mesh", False, 1920, 1080, 0, 5),         # BASELINE.json metric config = configs[2] on one GPU
    "pagoda_1080p_brdf": ("mesh", False, 1920, 1080, 2, 5),        # configs[1]
    "random_spheres_400x225_mis": ("random", False, 400, 225, 0, 50),  # configs[0]
    "earth_1080p_brdf": ("simple", False, 1920, 1080, 2, 5),       # configs[3]
    "broken_cube_4k_mis": ("simple_triangle", False, 3840, 2160, 0, 5),  # configs[4]
    "cornell_4k_mis": ("cornell", False, 3840, 2160, 0, 5),             # configs[4], its other scene: six instanced cubes (main.rs:92-211)
}
INTEGRATOR_NAMES = ["mis", "light_is", "brdf", "uniform"]
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
# Algorithmic bytes of one k_trace launch ("Roofline"): what the production kernel has to move per ray --
# 4 B queue entry + 64 B ray block read, 16 B hit record written (8 B for a shadow ray: t only), the certificate list of
# the ray's direction cell (8 B of cell offsets + 16 B per listed normal; scene mean) -- plus 136 B of stream window per
# record step of a wave (two s_load_dwordx16 + one dwordx2; counted by the kernel itself, rt_stats.trace_wave_steps).
B_QUEUE, B_RAY, B_HIT, B_HIT_SHADOW, B_WINDOW, B_CERT_CELL, B_CERT_ENTRY = 4, 64, 16, 8, 136, 8, 16
# the reference-order traversal this replaces: node 64 B, triangle 72 B, sphere 40 B per test, + ray 56 B in, hit 16 B out
B_NODE, B_TRI, B_SPH, B_RAY_REF = 64, 72, 40, 56 + 16


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="pagoda_1080p_mis", choices=sorted(WORKLOADS))
    ap.add_argument("--spp-per-step", type=int, default=50)       # x the driver's 20 steps = the 1000 spp of the metric
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--one-lane", action="store_true")            # A/B: the passes of a call one after the other
    ap.add_argument("--sync-steps", action="store_true")          # A/B: every step a blocking rt_render (two lanes inside the call) instead of two steps in flight
    ap.add_argument("--bvh-seed", type=int, default=3)
    # bvh.rs:53 draws every split axis from an unseeded RNG, so the reference's tree differs from run to run and every axis assignment is
    # one of its trees.  min_area (default): per node the axis, of those three, whose median split has the least area x count
    # (include/rt_host.h); seeded: the splitmix64 stream of --bvh-seed, as in the parity suites.  Parity is tested on both.
    ap.add_argument("--bvh-axis", default="min_area", choices=("min_area", "seeded"))
    ap.add_argument("--no-seeded-compare", action="store_true")   # skip the extra (untimed) steps on the seeded-axis tree
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))  # gloo + --renderer: the CPU test of the N-rank plumbing
    ap.add_argument("--renderer", default="")                     # python file with make_renderer(args, rank, world, device, distributed_module): replaces the HIP renderer (tests)
    return ap.parse_args(argv)


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks of this script as child processes.  Runs BEFORE anything here touches
    the GPU (torch is not even imported yet), and nothing is exec'ed in place."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    codes = [p.wait() for p in procs]
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or not lines:
        print(f"bench.py: ranks failed (rank, exit code): {bad}; JSON lines from rank 0: {len(lines)}", file=sys.stderr)
        return 1
    return 0


class HipRenderer:
    """The product path: librt_hip.so through raytracing_amd (raises if the library is missing)."""

    def __init__(self, args, rank, world, local_rank, dev, rtd, bvh_axis=None):
        import raytracing_amd as rt
        self.rt = rt
        scene, legacy_bvh, self.W, self.H, self.integrator, self.depth = WORKLOADS[args.workload]
        blob, self.desc, self.prune = b"", None, None
        if rank == 0:  # built and flattened once, broadcast as bytes, uploaded per rank
            hs = rt.HostScene.builtin(scene, bvh_seed=args.bvh_seed, legacy_bvh=legacy_bvh, bvh_axis=bvh_axis or args.bvh_axis)
            blob, self.desc, self.prune = hs.blob(), hs.describe(), hs.prune_info()
        blob = rtd.broadcast_blob(blob, dev)
        self.ctx = rt.Context(local_rank)
        self.gs = self.ctx.upload(blob)
        self.flags = rt.RENDER_ONE_LANE if args.one_lane else 0

    def render(self, accum, b, e, flags=0, samples_per_pass=0):
        return self.gs.render(accum.data_ptr(), self.W, self.H, b, e, self.depth, self.integrator, seed=0x5EED, samples_per_pass=samples_per_pass,
                              flags=self.flags | flags).as_dict()

    # rt_render_begin / rt_render_end: the call is queued on one of the scene's two lanes; two calls may be in flight
    def begin(self, accum, b, e):
        return self.gs.render_begin(accum.data_ptr(), self.W, self.H, b, e, self.depth, self.integrator, seed=0x5EED, flags=self.flags)

    def end(self, job):
        return self.gs.render_end(job).as_dict()

    def close(self):
        self.gs.close(); self.ctx.close()


def cpu_baseline(desc, integrator, depth, threads):
    """The oracle (a port: the reference is Rust and no toolchain exists here) on bounded samples of the same workload --
    same scene, camera, integrator and depth at a smaller frame (rates per ray do not depend on resolution): once on
    `threads` threads over 16x16 tiles (what the reference's rayon loop does when enabled, main.rs:979) and once on ONE
    thread (what the reference at this commit does, main.rs:980)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as orc
    osc = orc.OracleScene(desc)

    def run(w, h, spp, nthreads):
        _, st = osc.render(w, h, 0, spp, depth, integrator, seed=0x5EED, nthreads=nthreads)
        return st, {"value": st["rays"] / st["seconds"] / 1e6, "unit": "Mrays/s", "cores": nthreads,
                    "sample": f"same scene/camera/integrator/depth at {w}x{h}, {spp} spp: {st['rays']} rays, {st['paths']} paths in {st['seconds']:.2f} s "
                              f"(16x16 tiles over {nthreads} thread{'s' if nthreads > 1 else ''})",
                    "paths_per_s": st["paths"] / st["seconds"]}
    st, out = run(480, 270, 3, threads)
    out["kind"] = "port"
    out["tests_per_ray"] = {"box": st["box"] / max(st["rays"], 1), "tri": st["tri"] / max(st["rays"], 1), "sph": st["sph"] / max(st["rays"], 1)}
    _, one = run(160, 90, 2, 1)
    out["single_thread"] = one
    return out


def main():
    args = parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.spp_per_step < 1:
        print("bench.py: --gpus/--steps/--spp-per-step must be >= 1", file=sys.stderr)
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    from raytracing_amd import distributed as rtd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        return 2
    on_gpu = args.backend == "nccl"
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    if on_gpu:
        torch.cuda.set_device(local_rank)
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)

    scene, legacy_bvh, W, H, integrator, depth = WORKLOADS[args.workload]
    if args.renderer:   # injected per-rank renderer (tests/test_bench_spawn.py): same loop, same collectives, no GPU
        spec = importlib.util.spec_from_file_location("bench_renderer", args.renderer)
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        R = mod.make_renderer(args, rank, world, dev, rtd)
        W, H = R.W, R.H
    else:
        R = HipRenderer(args, rank, world, local_rank, dev, rtd)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    spp_step = args.spp_per_step
    total_steps = args.warmup + args.steps

    def step_range(i):
        if args.scaling == "weak":   # rank r owns the absolute sample indices [r*total_steps*spp_step, (r+1)*total_steps*spp_step)
            b = (rank * total_steps + i) * spp_step
            return b, b + spp_step
        return rtd.spp_range(rank, world, spp_step, i * spp_step)   # strong: step i = samples [i*spp_step, (i+1)*spp_step) split over the ranks

    pipelined = hasattr(R, "begin") and not args.sync_steps and not args.one_lane

    def run_steps(first, count):
        """`count` steps; pipelined: steps are queued two at a time (rt_render_begin x2, then rt_render_end x2), one 50-spp pass each
        on the scene's two lanes, so the thin deep bounces of one step overlap the other step's.  Begun TOGETHER on purpose: two steps
        in the same phase measured 6 % faster than a staggered pair (a thin launch queued behind the other step's full-machine
        launch waits for it).  Every step still adds its sums after the step before it."""
        out, pending = [], []
        for i in range(first, first + count):
            b, e = step_range(i)
            if e <= b:
                continue
            if pipelined:
                pending.append(R.begin(accum, b, e))
                if len(pending) == 2:
                    out += [R.end(pending.pop(0)), R.end(pending.pop(0))]
            else:
                out.append(R.render(accum, b, e))
        while pending:
            out.append(R.end(pending.pop(0)))
        return out

    sync()
    run_steps(0, args.warmup)
    accum.zero_()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    stats = run_steps(args.warmup, args.steps)
    rtd.allreduce_accum(accum)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0

    rays = sum(s["rays"] for s in stats)
    paths = sum(s["paths"] for s in stats)
    tot = torch.tensor([float(rays), float(paths), dt], dtype=torch.float64, device=dev)
    if world > 1:
        sums = tot.clone(); dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        rays_all, paths_all, t_max = float(sums[0]), float(sums[1]), float(mx[2])
    else:
        rays_all, paths_all, t_max = float(rays), float(paths), dt
    spp_total = args.steps * spp_step * (world if args.scaling == "weak" else 1)

    rc = 0
    if rank == 0:
        out = {
            "metric": "Mrays/s", "value": rays_all / t_max / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_max / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (reference pagoda OBJ asset, Lambertian 0.5 albedo; Philox-seeded camera samples)",
            "config": {"workload": args.workload, "scene": scene, "width": W, "height": H, "integrator": INTEGRATOR_NAMES[integrator],
                       "max_depth": depth, "spp_per_step": spp_step, "render_lanes": 1 if args.one_lane else 2, "spp_total": spp_total,
                       "steps_in_flight": 2 if pipelined else 1,
                       "sharding": f"spp-range x{world}" if world > 1 else "none", "bvh_seed": args.bvh_seed, "bvh_axis": args.bvh_axis},
            "paths_per_s": paths_all / t_max, "rays": rays_all, "paths": paths_all,
            "accum_checksum": float(accum.double().sum()),
        }
        if not args.renderer:
            out.update(report_kernels(args, R, accum, stats, step_range, rays, spp_step))
            if world == 1 and not args.no_cpu_baseline:
                threads = args.cpu_threads or min(16, os.cpu_count() or 1)
                out["cpu_baseline"] = cpu_baseline(R.desc, integrator, depth, threads)
    R.close()
    if rank == 0:
        if world == 1 and not args.renderer and args.bvh_axis != "seeded" and not args.no_seeded_compare:
            out["seeded_tree"] = seeded_compare(args, local_rank, dev, rtd, accum, sync, spp_step)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def seeded_compare(args, local_rank, dev, rtd, accum, sync, spp_step):
    """After the timed region and after its scene is closed: the same steps on the tree of the seeded axis stream (the tree of the
    parity suites and of rounds 1-2a), two warm-up and four timed steps, so the line shows what the axis choice is worth."""
    R = HipRenderer(args, 0, 1, local_rank, dev, rtd, bvh_axis="seeded")
    try:
        def pair(b):
            jobs = [R.begin(accum, b, b + spp_step), R.begin(accum, b + spp_step, b + 2 * spp_step)]
            return [R.end(j) for j in jobs]
        pair(0); sync()
        t0 = time.perf_counter()
        st = pair(2 * spp_step) + pair(4 * spp_step)
        sync()
        dt = time.perf_counter() - t0
    finally:
        R.close()
    rays = sum(s["rays"] for s in st)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "steps": 4, "bvh_axis": "seeded", "bvh_seed": args.bvh_seed,
            "note": "same workload on the tree built from the seeded axis stream; untimed extra, not part of `value`"}


def report_kernels(args, R, accum, stats, step_range, rays, spp_step):
    """rank 0, after the timed region: per-kernel times of the timed steps, the roofline of the dominant kernel (k_trace),
    and two untimed extra renders -- one step on ONE lane (launch durations without the other lane's kernels sharing the
    machine) and a 1-spp pass of the counting kernel (reference-order test counts: what the traversal replaces)."""
    rt = R.rt
    b, e = step_range(args.warmup)
    alone = R.render(accum, b, e, flags=rt.RENDER_ONE_LANE, samples_per_pass=stats[0]["samples_per_pass"] if stats else 0)
    # the waves' record steps come from a step of their own: the counter is one same-address atomic per wave (5 % of the frame), so neither the
    # timed steps nor the one-lane step above run with it
    stepped = R.render(accum, b, e, flags=rt.RENDER_COUNT_STEPS, samples_per_pass=stats[0]["samples_per_pass"] if stats else 0)
    counted = R.render(accum, b, b + 1, flags=rt.RENDER_COUNT_TESTS)
    r1 = max(counted["rays"], 1)
    n_box, n_tri, n_sph = counted["box_tests"] / r1, counted["tri_tests"] / r1, counted["sphere_tests"] / r1
    ms_trace = sum(s["ms_trace"] for s in stats)
    ms_shadow = sum(s["ms_trace_shadow"] for s in stats)
    launches = max(sum(s["trace_launches"] for s in stats), 1)
    shadow_launches = sum(s["trace_shadow_launches"] for s in stats)
    shadow_rays = sum(s["shadow_rays"] for s in stats)
    wave_steps = stepped["trace_wave_steps"] * rays / max(stepped["rays"], 1)   # that step's record steps per ray, scaled to the timed rays (same workload and sample indices as the first timed step)
    avg_launch_s = ms_trace / launches / 1e3
    rays_per_launch = rays / launches
    # mean certificate list per direction cell (lists are padded to multiples of four entries), 0 when the scene has no pruned space
    cert_entries = (R.prune["grid_entries"] / (3 * 1024 * 1024)) if R.prune and R.prune.get("on") else 0.0
    per_ray = {"queue_ray_hit": B_QUEUE + B_RAY + (B_HIT * (rays - shadow_rays) + B_HIT_SHADOW * shadow_rays) / max(rays, 1),
               "certificate": (B_CERT_CELL + B_CERT_ENTRY * cert_entries) if cert_entries else 0.0,
               "stream_windows": B_WINDOW * wave_steps / max(rays, 1)}
    bytes_per_ray = sum(per_ray.values())
    achieved_gbs = bytes_per_ray * rays_per_launch / avg_launch_s / 1e9
    ref_bytes_per_ray = n_box * B_NODE + n_tri * B_TRI + n_sph * B_SPH + B_RAY_REF
    traffic, pmc = None, {}
    pmc_path = os.path.join(ROOT, "profiles", "trace_pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pm = json.load(open(pmc_path))
            if (pm.get("workload") == args.workload and pm.get("spp_per_step") == alone["samples_per_pass"]
                    and pm.get("bvh_axis", "seeded") == args.bvh_axis):   # counters of a pass of this size on this tree
                traffic = pm.get("hbm_bytes_per_launch")
                pmc = {k: pm[k] for k in ("valu_issue_utilisation", "scalar_issue_utilisation", "valu_insts_per_ray", "lane_use_by_launch",
                                          "measured_hbm_frac", "spp_per_step", "source") if k in pm}
        except Exception:
            traffic, pmc = None, {}
    return {
        "kernel_ms": {"trace": ms_trace, "trace_shadow_part": ms_shadow, "shade": sum(s["ms_shade"] for s in stats), "other": sum(s["ms_other"] for s in stats),
                      "total": sum(s["ms_total"] for s in stats),
                      "note": "HIP-event sums per launching stream; with two render lanes the launches of both lanes overlap, so the parts can exceed the total -- "
                              "only the one-lane figures below attribute time to a kernel"},
        "samples_per_pass": stats[0]["samples_per_pass"] if stats else None,
        "roofline": {"kernel": "k_trace<false, false, false>", "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_ray": bytes_per_ray, "algorithmic_bytes_per_ray_parts": per_ray,
                     "rays_per_launch": rays_per_launch, "launches": launches, "avg_launch_ms": avg_launch_s * 1e3,
                     "shadow": {"launches": shadow_launches, "rays": shadow_rays, "avg_launch_ms": ms_shadow / max(shadow_launches, 1)},
                     # the same launches with nothing else on the machine (untimed extra step on one lane)
                     "one_lane": {"avg_launch_ms": alone["ms_trace"] / max(alone["trace_launches"], 1), "step_ms": alone["ms_total"],
                                  "trace_ms": alone["ms_trace"], "trace_shadow_ms": alone["ms_trace_shadow"], "shade_ms": alone["ms_shade"],
                                  "other_ms": alone["ms_other"], "rays": alone["rays"], "shadow_rays": alone["shadow_rays"],
                                  "achieved": bytes_per_ray * alone["rays"] / max(alone["ms_trace"], 1e-9) / 1e6,
                                  "frac": bytes_per_ray * alone["rays"] / max(alone["ms_trace"], 1e-9) / 1e6 / HBM_PEAK_GBS},
                     "issue_pmc": pmc,
                     "executed": {"record_steps_per_ray": wave_steps / max(rays, 1), "record_steps_per_wave": 64.0 * wave_steps / max(rays, 1)},
                     "note": "achieved = algorithmic bytes the production kernel moves per launch (queue entry, ray block, hit record, certificate list, 136-B stream "
                             "window per record step of a wave -- the step count comes from the kernel) / average k_trace launch duration by HIP events over the "
                             "timed region; traffic = measured L2-miss bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/); the kernel is bound by "
                             "instruction issue and lane use, not by HBM (issue_pmc)"},
        # what the reference-order traversal (bvh.rs:91-106, per-axis box test) would move for the same rays, from an untimed pass of the counting kernel
        "algorithmic_saving": {"reference_order_bytes_per_ray": ref_bytes_per_ray, "tests_per_ray": {"box": n_box, "tri": n_tri, "sph": n_sph},
                               "ratio": ref_bytes_per_ray / max(bytes_per_ray, 1e-9),
                               "note": "SURVEY.md 8d byte model of the reference's own visited set / the bytes the pruned walk moves; not a bandwidth figure"},
    }


if __name__ == "__main__":
    sys.exit(main())
