"""Injected per-rank renderer for tests/test_bench_spawn.py: bench.py's rank loop, sample sharding, scene broadcast and
accumulator all-reduce run as they do on the GPU box, but every sample s adds (s + 1) to every channel instead of radiance."""
import torch


class FakeRenderer:
    W, H = 8, 4

    def __init__(self, args, rank, world, dev, rtd):
        blob = rtd.broadcast_blob(b"scene bytes from rank 0" if rank == 0 else b"", dev)
        assert blob == b"scene bytes from rank 0", blob
        self.rendered = []

    def render(self, accum, b, e, flags=0):
        accum += float(sum(s + 1 for s in range(b, e)))
        self.rendered.append((b, e))
        n = (e - b) * self.W * self.H
        return {"rays": 2 * n, "paths": n, "ms_trace": 0.0, "ms_trace_shadow": 0.0, "trace_launches": 0, "trace_shadow_launches": 0, "shadow_rays": 0,
                "trace_wave_steps": 0, "ms_shade": 0.0, "ms_other": 0.0, "ms_total": 0.0, "samples_per_pass": e - b, "lanes": 1}

    def close(self):
        pass


def make_renderer(args, rank, world, dev, rtd):
    return FakeRenderer(args, rank, world, dev, rtd)
