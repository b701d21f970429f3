#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline: Mrays/s (+ paths/s) of the pagoda OBJ scene, 1920x1080, MIS integrator
(`MultipleImportanceSampleIntegrator`, integrator.rs:31-171 -- the one the reference's main() selects), depth 5
(main.rs:944), 1000 spp = configs[2], which fits one GPU.

A "step" is --spp-per-step samples of every pixel (ray-gen -> [traverse -> shade -> traverse shadow rays] x depth -> unwind ->
accumulate); the default 50 makes the driver's `--steps 20` exactly the 1000-spp frame, and ms_per_step = time / steps (how the renderer batches the
frame's samples into jobs is its own business: --job-spp).
The samples are rendered as JOBS queued with rt_render_begin and waited for with rt_render_end, two in flight (one pass each, on
the scene's two lanes): the thin deep bounces of one job overlap the other job's kernels; --sync-steps runs blocking rt_render
calls (two lanes of half-size passes inside each call) instead.  Inputs (the flattened scene) are resident in HBM before the timed
region; accumulators stay on the device.

N ranks (main.rs:1009-1011: tonemap after the sum): `--scaling strong` (default) renders THE frame -- the
steps x spp_per_step absolute sample indices [0, S) -- split as contiguous per-rank ranges [floor(S r/N), floor(S (r+1)/N));
every rank cuts its range into jobs of at least spp_per_step samples, as large as one pass per lane allows (--job-spp 0: 72 spp at 1080p, so the
1000-spp frame is 14 jobs of 71-72 on one rank and 2 jobs of 62 and 63 on each of 8), so job size does not shrink with N.  `--scaling weak` gives every rank its own S samples (rank r: [S r, S (r+1))).  One broadcast of the
scene bytes before, one all-reduce of the fp32 sums inside the timed region (reported separately as allreduce_ms).

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child processes,
before this process touches the GPU), polls them all, and on the first rank that exits non-zero terminates the others and
returns non-zero; every rank's stderr is kept and relayed.  Under `torch.distributed.run` it is one of the ranks.  n_gpus in
the line is the live world size and must equal --gpus.

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
    # how a rank cuts its samples into jobs (one pass each, two in flight): 0 = auto -- as large as one pass per lane allows (72 spp at 1080p, depth 5),
    # never below spp_per_step; N = jobs of about N samples (--job-spp 10 with --spp-per-step 10: one 10-spp FRAME per call, configs[0] at its own size)
    ap.add_argument("--job-spp", type=int, default=0)
    # strong (default): the frame of steps x spp_per_step samples, split over the ranks as contiguous sample ranges;
    # weak: every rank renders steps x spp_per_step samples of its own.  The same thing on one rank.
    ap.add_argument("--scaling", default="strong", choices=("weak", "strong"))
    ap.add_argument("--one-lane", action="store_true")            # A/B: the passes of a call one after the other
    ap.add_argument("--sync-steps", action="store_true")          # A/B: every job a blocking rt_render (two lanes inside the call) instead of two jobs in flight
    ap.add_argument("--bvh-seed", type=int, default=3)
    # bvh.rs:53 draws every split axis from an unseeded RNG, so the reference's tree differs from run to run and every axis assignment is
    # one of its trees.  min_area (default): per node the axis, of those three, whose median split has the least area x count
    # (include/rt_host.h); seeded: the splitmix64 stream of --bvh-seed, as in the parity suites.  Parity is tested on both.
    ap.add_argument("--bvh-axis", default="min_area", choices=("min_area", "seeded"))
    ap.add_argument("--no-seeded-compare", action="store_true")   # skip the extra (untimed) steps on the seeded-axis tree
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))  # gloo + --renderer: the CPU test of the N-rank plumbing
    # gloo collectives (staged through host memory) around the REAL renderer, every rank on GPU (local rank mod device count): the N-rank path end to end
    # on a box with fewer GPUs than ranks, where RCCL refuses two ranks on one device (tests/test_gpu_multirank.py); never a measurement
    ap.add_argument("--gloo-gpu", action="store_true")
    ap.add_argument("--renderer", default="")                     # python file with make_renderer(args, rank, world, device, distributed_module): replaces the HIP renderer (tests)
    # rehearsal on ONE GPU of what rank r of an N-rank run does: "r/N" -> this process renders rank r's share of the frame (same jobs,
    # same sample indices), with no process group; the line then carries `rehearsal` and is not a scaling measurement
    ap.add_argument("--rehearse-rank", default="")
    return ap.parse_args(argv)


def plan_jobs(b, e, job_spp, job_max=0):
    """Cuts the sample range [b, e) into contiguous jobs of near-equal size.  Every job has at least job_spp samples (unless the whole range is
    smaller), so the per-launch ray counts do not shrink with the number of ranks.  job_max > job_spp: jobs as large as job_max allows -- the fewest
    jobs of at most job_max samples, an even number of them when there are several (two are in flight: an odd one out would run alone) -- because a
    pass's deeper bounces are the denser, and the cheaper per ray, the more samples it holds (25 / 50 / 100 spp per pass cost
    2.65 / 2.32 / 2.10 ms per sample).  0 < job_max < job_spp (a step does not fit one pass: 4K, 25 spp against 18): jobs of at most job_max, each ONE
    pass, instead of jobs of a step that run as three short passes (r03x: broken cube 4K +5 %).  job_max 0: max(1, (e-b) // job_spp) jobs."""
    n = e - b
    if n <= 0:
        return []
    job_spp = max(job_spp, 1)
    k = max(1, n // job_spp)
    if job_max > 0 and job_max != job_spp:
        kk = -(-n // job_max)
        if kk > 1 and kk % 2:
            kk += 1
        if job_max < job_spp or kk == 1 or n // kk >= job_spp:
            k = min(kk, n)
    return [(b + (i * n) // k, b + ((i + 1) * n) // k) for i in range(k)]


def rank_range(scaling, rank, world, frame_spp):
    """Absolute sample indices of rank `rank`: strong -- its contiguous share of the frame [0, frame_spp) (distributed.spp_range);
    weak -- a frame of its own."""
    if scaling == "weak":
        return rank * frame_spp, (rank + 1) * frame_spp
    return (rank * frame_spp) // world, ((rank + 1) * frame_spp) // world


def spawn_ranks(args):
    """--gpus N without a launcher: start N ranks of this script as child processes.  Runs BEFORE anything here touches
    the GPU (torch is not even imported yet), and nothing is exec'ed in place.  All children are polled: the first one that
    exits non-zero ends the run -- its siblings (which would otherwise sit in a collective until some time limit) are
    terminated, by the PIDs started here, and the exit code is non-zero.  stdout and stderr of every rank are kept."""
    import signal
    import tempfile
    signal.signal(signal.SIGTERM, lambda *a: sys.exit(143))   # so that the `finally` below ends the ranks when this parent is told to stop
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix="rt_bench_ranks_")
    procs, files = [], []
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            fo, fe = open(os.path.join(tmp, f"rank{r}.out"), "w+"), open(os.path.join(tmp, f"rank{r}.err"), "w+")
            files.append((fo, fe))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=fo, stderr=fe, text=True))
        codes = [None] * len(procs)
        first_bad = None
        while any(c is None for c in codes) and first_bad is None:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
                    if codes[r] not in (None, 0) and first_bad is None:
                        first_bad = r
            if first_bad is None:
                time.sleep(0.05)
        if first_bad is not None:   # fail fast: nobody waits for a rank that is gone
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            deadline = time.time() + 10.0
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
    finally:
        for p in procs:   # an exception / signal in this parent must not leave ranks behind
            if p.poll() is None:
                p.kill()
                p.wait()
    outs = []
    for r, (fo, fe) in enumerate(files):
        fo.seek(0); fe.seek(0)
        outs.append(fo.read())
        err = fe.read()
        for ln in err.splitlines():
            print(f"[rank {r}] {ln}", file=sys.stderr)
        for ln in (outs[r].splitlines() if r else []):   # only rank 0's stdout carries the line; anything a sibling printed is kept too
            print(f"[rank {r} stdout] {ln}", file=sys.stderr)
        fo.close(); fe.close()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or not lines:
        what = f"rank {first_bad} exited with code {codes[first_bad]} first; the other ranks were terminated" if first_bad is not None else "no JSON line from rank 0"
        print(f"bench.py: {what} (rank, exit code): {bad}", file=sys.stderr)
        return 1
    print(lines[-1], flush=True)
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
        import ctypes
        n_lights = rt._ffi.BlobHeader.from_buffer_copy(bytes(blob[:ctypes.sizeof(rt._ffi.BlobHeader)])).n_lights
        # no light sample can occur (run_paths in rt_device.hip): 32-byte bounce records, and a records buffer sized for them
        self.compact_records = not (self.integrator == rt.INT_MIS and n_lights > 0)

    def render(self, accum, b, e, flags=0, samples_per_pass=0):
        return self.gs.render(accum.data_ptr(), self.W, self.H, b, e, self.depth, self.integrator, seed=0x5EED, samples_per_pass=samples_per_pass,
                              flags=self.flags | flags).as_dict()

    # rt_render_begin / rt_render_end: the call is queued on one of the scene's two lanes; two calls may be in flight
    def begin(self, accum, b, e, samples_per_pass=0):
        return self.gs.render_begin(accum.data_ptr(), self.W, self.H, b, e, self.depth, self.integrator, seed=0x5EED, samples_per_pass=samples_per_pass,
                                    flags=self.flags)

    LANE_BYTES = 112 << 30   # path state per lane: two lanes + scene + certificate tables stay under the 288 GB

    def bytes_per_sample(self):
        return self.W * self.H * (self.depth * (32 if self.compact_records else 96) + 320)     # depth x 96 (or 32) B of bounce records + 320 B per path

    def pass_size_for(self, n_spp):
        """A job of n_spp samples runs as ONE pass when its path state fits LANE_BYTES; otherwise the library's default pass size (0)."""
        return n_spp if n_spp * self.bytes_per_sample() <= self.LANE_BYTES and n_spp * self.W * self.H < 0x7FFFFF00 else 0

    def max_job_spp(self):
        """The largest job that still runs as one pass per lane (1080p, depth 5: 72 spp with full MIS records, 120 with compact ones)."""
        return max(1, min(self.LANE_BYTES // self.bytes_per_sample(), (0x7FFFFF00 - 1) // (self.W * self.H)))

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
    rehearse = None
    if args.rehearse_rank:
        try:
            rr, rn = (int(x) for x in args.rehearse_rank.split("/"))
            assert 0 <= rr < rn and args.gpus == 1
        except Exception:
            print("bench.py: --rehearse-rank wants r/N with 0 <= r < N, and --gpus 1", file=sys.stderr)
            return 2
        rehearse = (rr, rn)
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
    if args.gloo_gpu:
        args.backend = "gloo"
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    on_gpu = args.backend == "nccl" or args.gloo_gpu
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")    # where the collectives' tensors live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
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
        R = HipRenderer(args, rank, world, local_rank, coll_dev, rtd)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    spp_step = args.spp_per_step
    frame_spp = args.steps * spp_step                      # the frame: absolute sample indices [0, frame_spp)
    shard_rank, shard_world = rehearse if rehearse else (rank, world)
    my_b, my_e = rank_range(args.scaling, shard_rank, shard_world, frame_spp)
    pass_size = getattr(R, "pass_size_for", lambda n: 0)
    job_min = args.job_spp if args.job_spp > 0 else spp_step
    job_max = 0 if args.job_spp > 0 or not hasattr(R, "max_job_spp") else R.max_job_spp()
    jobs = plan_jobs(my_b, my_e, job_min, job_max)         # this rank's share as jobs of >= spp_per_step samples, one pass each where memory allows
    # warm-up: `--warmup` jobs of this rank's largest job size (so the path buffers the timed jobs need exist), on sample indices past
    # every rank's share; their sums are thrown away
    big = max((e - b for b, e in jobs), default=spp_step)
    warm_first = max(shard_world, 1) * frame_spp
    pipelined = hasattr(R, "begin") and not args.sync_steps and not args.one_lane
    n_warm = 2 if (pipelined and args.warmup == 1) else args.warmup      # two jobs in flight: a single warm-up job would leave the second lane's ~100 GB of path buffers to be allocated inside the timed region
    warm_jobs = [(warm_first + i * big, warm_first + (i + 1) * big) for i in range(n_warm)]

    def run_jobs(todo):
        """Renders the sample ranges of `todo` in order.  Pipelined: two jobs in flight (rt_render_begin / rt_render_end, one pass each on the
        scene's two lanes), the next one begun as soon as the oldest has ended, so the thin deep bounces of one job overlap the other job's
        kernels and an odd job count leaves no lane idle for a whole job.  Every job still adds its sums after the job begun before it."""
        out, pending = [], []
        for b, e in todo:
            if pipelined:
                if len(pending) == 2:
                    out.append(R.end(pending.pop(0)))
                pending.append(R.begin(accum, b, e, samples_per_pass=pass_size(e - b)))
            else:
                out.append(R.render(accum, b, e, samples_per_pass=pass_size(e - b)))
        while pending:
            out.append(R.end(pending.pop(0)))
        return out

    sync()
    run_jobs(warm_jobs)
    if world > 1 and args.warmup > 0:
        rtd.allreduce_accum(accum)     # the warm-up covers the collective too: the first all-reduce of a size sets the communicator's rings and buffers up
    accum.zero_()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    stats = run_jobs(jobs)
    sync()
    t_render = time.perf_counter() - t0
    rtd.allreduce_accum(accum)
    sync()
    t_allreduce = time.perf_counter() - t0 - t_render
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0

    rays = sum(s["rays"] for s in stats)
    paths = sum(s["paths"] for s in stats)
    mine = torch.tensor([float(rays), float(paths), dt, t_render, t_allreduce, float(my_e - my_b)], dtype=torch.float64, device=coll_dev)
    if world > 1:
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = torch.stack(every).cpu()
    else:
        every = mine.cpu().reshape(1, -1)
    rays_all, paths_all, t_max = float(every[:, 0].sum()), float(every[:, 1].sum()), float(every[:, 2].max())
    spp_total = int(every[:, 5].sum())
    checksum = float(accum.double().sum())
    if world > 1:   # the ranks part here: what rank 0 measures beside the line (one-lane step, counting pass, CPU baseline) is its own business
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        R.close()
        return 0

    per_rank_rate = [float(every[r, 0] / max(float(every[r, 3]), 1e-12) / 1e6) for r in range(world)]
    out = {
        "metric": "Mrays/s", "value": rays_all / t_max / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": t_max / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic (reference pagoda OBJ asset, Lambertian 0.5 albedo; Philox-seeded camera samples)",
        "config": {"workload": args.workload, "scene": scene, "width": W, "height": H, "integrator": INTEGRATOR_NAMES[integrator],
                   "max_depth": depth, "spp_per_step": spp_step, "render_lanes": 1 if args.one_lane else 2, "spp_total": spp_total,
                   "frame_spp": frame_spp, "jobs_in_flight": 2 if pipelined else 1, "job_spp": "auto" if args.job_spp <= 0 else args.job_spp,
                   "sharding": ("none" if world == 1 else
                                f"spp-range x{world}: rank r renders samples [floor({frame_spp}r/{world}), floor({frame_spp}(r+1)/{world})) of the frame, as jobs of >= {spp_step} spp"
                                if args.scaling == "strong" else
                                f"spp-range x{world}: rank r renders samples [{frame_spp}r, {frame_spp}(r+1)), as jobs of >= {spp_step} spp"),
                   "per_rank_spp": [int(v) for v in every[:, 5]], "rank0_jobs_spp": [e - b for b, e in jobs],
                   "bvh_seed": args.bvh_seed, "bvh_axis": args.bvh_axis},
        "paths_per_s": paths_all / t_max, "rays": rays_all, "paths": paths_all,
        "render_ms": {"min": float(every[:, 3].min()) * 1e3, "max": float(every[:, 3].max()) * 1e3},
        "allreduce_ms": float(every[:, 4].max()) * 1e3,     # slowest rank: sum of the fp32 accumulators (W x H x 3), inside the timed region
        "per_rank_mrays_s": {"min": min(per_rank_rate), "max": max(per_rank_rate)},
        "accum_checksum": checksum,
    }
    if args.gloo_gpu:
        out["gloo_gpu"] = "gloo collectives staged through host memory around the HIP renderer, ranks sharing GPUs: a functional run of the N-rank path, not a measurement"
    if rehearse:
        out["rehearsal"] = {"rank": rehearse[0], "world": rehearse[1],
                            "note": "ONE GPU rendering the share rank r of an N-rank run would render (same jobs, same sample indices); no collective ran"}
    if not args.renderer:
        out.update(report_kernels(args, R, accum, stats, jobs, rays))
        if not args.no_cpu_baseline:   # on rank 0's host cores, after the ranks have parted
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(R.desc, integrator, depth, threads)
    R.close()
    if world == 1 and not rehearse and not args.renderer and args.bvh_axis != "seeded" and not args.no_seeded_compare:
        out["seeded_tree"] = seeded_compare(args, local_rank, dev, rtd, accum, sync, max((e - b for b, e in jobs), default=spp_step))
    print(json.dumps(out), flush=True)
    return 0


def seeded_compare(args, local_rank, dev, rtd, accum, sync, spp_step):
    """After the timed region and after its scene is closed: jobs of the timed region's size on the tree of the seeded axis stream (the tree of
    the parity suites and of rounds 1-2a), two warm-up and four timed jobs, so the line shows what the axis choice is worth."""
    R = HipRenderer(args, 0, 1, local_rank, dev, rtd, bvh_axis="seeded")
    try:
        def pair(b):
            ps = R.pass_size_for(spp_step)
            jobs = [R.begin(accum, b, b + spp_step, samples_per_pass=ps), R.begin(accum, b + spp_step, b + 2 * spp_step, samples_per_pass=ps)]
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


def report_kernels(args, R, accum, stats, jobs, rays):
    """rank 0, after the timed region: per-kernel times of the timed steps, the roofline of the dominant kernel (k_trace),
    and two untimed extra renders -- one step on ONE lane with the shadow traversals on the same stream (launch durations without
    another lane's or stream's kernels sharing the machine: the only figures that attribute time to a kernel) and a 1-spp pass of the counting kernel (reference-order test counts: what the traversal replaces)."""
    rt = R.rt
    b, e = jobs[0]
    alone = R.render(accum, b, e, flags=rt.RENDER_ONE_LANE | rt.RENDER_SERIAL_SHADOW, samples_per_pass=stats[0]["samples_per_pass"] if stats else 0)
    # the waves' record steps come from a step of their own: the counter is one same-address atomic per wave (5 % of the frame), so neither the
    # timed steps nor the one-lane step above run with it
    stepped = R.render(accum, b, e, flags=rt.RENDER_COUNT_STEPS, samples_per_pass=stats[0]["samples_per_pass"] if stats else 0)
    counted = R.render(accum, b, b + 1, flags=rt.RENDER_COUNT_TESTS)
    r1 = max(counted["rays"], 1)
    n_box, n_tri, n_sph = counted["box_tests"] / r1, counted["tri_tests"] / r1, counted["sphere_tests"] / r1
    ms_trace = sum(s["ms_trace"] for s in stats)
    ms_shadow = sum(s["ms_trace_shadow"] for s in stats)
    launches = sum(s["trace_launches"] for s in stats)
    ms_tail = sum(s.get("ms_tail", 0.0) for s in stats)
    tail_bounces = sum(s.get("tail_bounces", 0) for s in stats)
    tail_only = launches == 0   # a frame so small that every bounce ran in the small-queue mode: its launches are timed as ONE span per pass (rt_stats.ms_tail),
    if tail_only:               # shading included, so the traversal rate derived from it is a lower bound
        ms_trace, launches = ms_tail, max(tail_bounces, 1)
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
        "kernel_ms": {"trace": 0.0 if tail_only else ms_trace, "trace_shadow_part": ms_shadow, "shade": sum(s["ms_shade"] for s in stats), "other": sum(s["ms_other"] for s in stats),
                      "tail": ms_tail, "tail_bounces": tail_bounces,   # bounces over small queues: traversal + shading + visibility launches under one span per pass
                      "total": sum(s["ms_total"] for s in stats),
                      "note": "HIP-event sums per launching stream; with two render lanes the launches of both lanes overlap, so the parts can exceed the total -- "
                              "only the one-lane figures below attribute time to a kernel"},
        "samples_per_pass": stats[0]["samples_per_pass"] if stats else None,
        "roofline": {"kernel": "k_trace<false, false, false>", "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                     # `achieved` prices every 136-B stream window as if it came from HBM although most are served by the scalar cache / L2;
                     # measured_frac is the PMC-measured L2-miss traffic of the same launches (one lane) over their duration, / peak -- the HBM rate proper
                     "measured_frac": pmc.get("measured_hbm_frac"), "limiter": "instruction issue x lane use, not bandwidth (issue_pmc)",
                     "algorithmic_bytes_per_ray": bytes_per_ray, "algorithmic_bytes_per_ray_parts": per_ray,
                     "rays_per_launch": rays_per_launch, "launches": launches, "avg_launch_ms": avg_launch_s * 1e3,
                     "shadow": {"launches": shadow_launches, "rays": shadow_rays, "avg_launch_ms": ms_shadow / max(shadow_launches, 1)},
                     # the same launches with nothing else on the machine (untimed extra step on one lane)
                     "one_lane": {"avg_launch_ms": (alone["ms_trace"] or alone.get("ms_tail", 0.0)) / max(alone["trace_launches"] or alone.get("tail_bounces", 0), 1), "step_ms": alone["ms_total"],
                                  "trace_ms": alone["ms_trace"], "tail_ms": alone.get("ms_tail", 0.0), "trace_shadow_ms": alone["ms_trace_shadow"], "shade_ms": alone["ms_shade"],
                                  "other_ms": alone["ms_other"], "rays": alone["rays"], "shadow_rays": alone["shadow_rays"],
                                  "achieved": bytes_per_ray * alone["rays"] / max(alone["ms_trace"] + alone.get("ms_tail", 0.0), 1e-9) / 1e6,
                                  "frac": bytes_per_ray * alone["rays"] / max(alone["ms_trace"] + alone.get("ms_tail", 0.0), 1e-9) / 1e6 / HBM_PEAK_GBS},
                     "issue_pmc": pmc,
                     "executed": {"record_steps_per_ray": wave_steps / max(rays, 1), "record_steps_per_wave": 64.0 * wave_steps / max(rays, 1)},
                     "note": "achieved = algorithmic bytes incl. cache-served re-reads: what the production kernel moves per launch (queue entry, ray block, hit record, certificate list, 136-B stream "
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
