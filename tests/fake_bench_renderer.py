"""Injected per-rank renderer for tests/test_bench_spawn.py: bench.py's rank loop, sample sharding, scene broadcast and
accumulator all-reduce run as they do on the GPU box, but every sample s adds (s + 1) to every channel instead of radiance.
FAKE_BENCH_FAIL_RANK=r makes rank r raise in its first timed job (the fail-fast test); FAKE_BENCH_LOG=dir records the
sample ranges every rank rendered."""
import json
import os

import torch


class FakeRenderer:
    W, H = 8, 4

    def __init__(self, args, rank, world, dev, rtd):
        blob = rtd.broadcast_blob(b"scene bytes from rank 0" if rank == 0 else b"", dev)
        assert blob == b"scene bytes from rank 0", blob
        self.rank, self.rendered = rank, []
        self.fail_rank = int(os.environ.get("FAKE_BENCH_FAIL_RANK", "-1"))
        self.warm_first = world * args.steps * args.spp_per_step   # bench.py's warm-up jobs use sample indices from here on

    def render(self, accum, b, e, flags=0, samples_per_pass=0):
        if self.rank == self.fail_rank and b < self.warm_first:   # a timed job, not the warm-up
            raise RuntimeError(f"fake renderer: rank {self.rank} fails on purpose")
        accum += float(sum(s + 1 for s in range(b, e)))
        self.rendered.append((b, e))
        n = (e - b) * self.W * self.H
        return {"rays": 2 * n, "paths": n, "ms_trace": 0.0, "ms_trace_shadow": 0.0, "trace_launches": 0, "trace_shadow_launches": 0, "shadow_rays": 0,
                "trace_wave_steps": 0, "ms_shade": 0.0, "ms_other": 0.0, "ms_total": 0.0, "samples_per_pass": e - b, "lanes": 1}

    def close(self):
        log = os.environ.get("FAKE_BENCH_LOG")
        if log:
            with open(os.path.join(log, f"rank{self.rank}.json"), "w") as f:
                json.dump(self.rendered, f)


def make_renderer(args, rank, world, dev, rtd):
    return FakeRenderer(args, rank, world, dev, rtd)
