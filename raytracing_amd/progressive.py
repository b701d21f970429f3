"""Progressive / checkpointed rendering: a render as a sequence of absolute sample
ranges added into fp32 radiance sums, persisted after every chunk in the format of include/rt_checkpoint.h.

The reference keeps the frame in memory and writes output.ppm from its ctrl-c handler
(desktop/src/main.rs:970-974); it cannot continue a render.  Here `ProgressiveRender.run(stop=...)` returns
after the chunk in flight when `stop()` turns true (the handler's job: the caller resolves and writes the
partial image), and `ProgressiveRender.resume(...)` continues at `spp_done`: Philox counters are keyed on the
absolute (pixel, sample, bounce), so an interrupted and resumed render adds the same chunk sums in the same
order as an uninterrupted one -- the accumulators are bit-identical.  With R ranks each rank checkpoints its own
range (`path` + ".rank<r>"); the all-reduce happens once at the end (raytracing_amd/distributed.py)."""
import os
import struct
import zlib

import numpy as np

MAGIC = b"RTCKPT1\0"
HEADER = struct.Struct("<8s5I2i3I2Q")       # include/rt_checkpoint.h: rt_checkpoint_header, 64 bytes
assert HEADER.size == 64
FIELDS = ("width", "height", "spp_begin", "spp_done", "spp_end", "max_depth", "integrator", "rank", "world", "chunk", "seed", "scene_id")


class CheckpointError(ValueError):
    pass


def scene_id(blob):
    b = bytes(blob)
    return ((len(b) & 0xFFFFFFFF) << 32) | (zlib.crc32(b) & 0xFFFFFFFF)


def save_checkpoint(path, header, accum):
    """Atomically writes header (dict with FIELDS) + sums: a reader sees the old file or the new one, never a part."""
    acc = np.ascontiguousarray(accum, np.float32)
    if acc.size != header["width"] * header["height"] * 3:
        raise CheckpointError("accumulator size does not match width x height x 3")
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(HEADER.pack(MAGIC, *(int(header.get(k, 0)) for k in FIELDS)))
        f.write(acc.tobytes())
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, path)


def load_checkpoint(path):
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < HEADER.size:
        raise CheckpointError("checkpoint shorter than its header")
    vals = HEADER.unpack_from(raw)
    if vals[0] != MAGIC:
        raise CheckpointError("not a render checkpoint (bad magic)")
    h = dict(zip(FIELDS, vals[1:]))
    n = h["width"] * h["height"] * 3
    if len(raw) != HEADER.size + 4 * n:
        raise CheckpointError(f"checkpoint holds {len(raw) - HEADER.size} bytes of sums, header says {4 * n}")
    if not (h["spp_begin"] <= h["spp_done"] <= h["spp_end"]):
        raise CheckpointError("checkpoint sample counters out of order")
    return h, np.frombuffer(raw, np.float32, n, HEADER.size).reshape(h["height"], h["width"], 3).copy()


class ProgressiveRender:
    """render_fn(spp_a, spp_b) -> float32 [height, width, 3] radiance SUMS of the absolute samples [spp_a, spp_b)
    (e.g. lambda a, b: scene.render_host(w, h, a, b, depth, integrator, seed)[0])."""

    def __init__(self, render_fn, blob, width, height, spp_begin, spp_end, max_depth, integrator, seed, chunk, path=None, rank=0, world=1):
        if chunk < 1 or spp_end < spp_begin:
            raise ValueError("chunk must be >= 1 and spp_end >= spp_begin")
        self.render_fn, self.chunk = render_fn, int(chunk)
        self.path = None if path is None else (path if world == 1 else f"{path}.rank{rank}")
        self.header = dict(width=width, height=height, spp_begin=spp_begin, spp_done=spp_begin, spp_end=spp_end, max_depth=max_depth,
                           integrator=integrator, rank=rank, world=world, chunk=int(chunk), seed=seed, scene_id=scene_id(blob))
        self.accum = np.zeros((height, width, 3), np.float32)

    @classmethod
    def resume(cls, render_fn, blob, path, chunk, rank=0, world=1, expect=None):
        """Continues the render whose checkpoint is at `path`.  `expect` (dict of header fields, e.g. width, height,
        max_depth, integrator, seed, spp_end) must match what the file records; so must the scene."""
        file = path if world == 1 else f"{path}.rank{rank}"
        h, acc = load_checkpoint(file)
        if h["scene_id"] != scene_id(blob):
            raise CheckpointError("checkpoint was written for another scene")
        if (h["rank"], h["world"]) != (rank, world):
            raise CheckpointError(f"checkpoint belongs to rank {h['rank']} of {h['world']}")
        if h["chunk"] not in (0, int(chunk)):   # another chunk size would add the same samples in another grouping: not bit-identical
            raise CheckpointError(f"checkpoint was written with chunk = {h['chunk']}, this render asks for {chunk}")
        for k, v in (expect or {}).items():
            if h[k] != v:
                raise CheckpointError(f"checkpoint has {k} = {h[k]}, this render asks for {v}")
        self = cls(render_fn, blob, h["width"], h["height"], h["spp_begin"], h["spp_end"], h["max_depth"], h["integrator"], h["seed"], chunk, path, rank, world)
        self.header, self.accum = h, acc
        return self

    @property
    def spp_done(self):
        return self.header["spp_done"] - self.header["spp_begin"]

    @property
    def complete(self):
        return self.header["spp_done"] == self.header["spp_end"]

    def run(self, stop=lambda: False, on_chunk=None):
        """Renders chunk after chunk until the range is done or stop() is true after a chunk.  Chunk boundaries sit at
        spp_begin + k * chunk whatever the resume point, so an interrupted render adds the same sums in the same order.
        Returns (sums, samples in them)."""
        h = self.header
        while h["spp_done"] < h["spp_end"]:
            a = h["spp_done"]
            b = min(h["spp_begin"] + ((a - h["spp_begin"]) // self.chunk + 1) * self.chunk, h["spp_end"])
            part = np.asarray(self.render_fn(a, b), np.float32)
            if part.shape != self.accum.shape:
                raise ValueError("render_fn returned a frame of the wrong shape")
            self.accum += part
            h["spp_done"] = b
            if self.path is not None:
                save_checkpoint(self.path, h, self.accum)
            if on_chunk is not None:
                on_chunk(self)
            if stop():
                break
        return self.accum, self.spp_done
