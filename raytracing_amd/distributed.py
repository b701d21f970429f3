"""Sample-index sharding across ranks: every rank holds the whole flattened scene and
the whole frame, renders a disjoint range of absolute sample indices (Philox counters are keyed on
the absolute (pixel, sample, level), so the union over ranks is the same set of paths as one GPU
rendering everything), and the fp32 radiance SUMS are added with one all-reduce.  Tonemapping is
applied to the mean (main.rs:1009-1011), i.e. after the all-reduce.

torch.distributed is plumbing only: broadcast of the scene bytes, all-reduce of the accumulator
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def spp_range(rank, world, spp_total, spp_first=0):
    """Rank r renders samples [floor(r*spp/R), floor((r+1)*spp/R)) (+ spp_first)."""
    return spp_first + (rank * spp_total) // world, spp_first + ((rank + 1) * spp_total) // world


def broadcast_blob(blob, device, src=0):
    """One broadcast of the flattened scene bytes from rank `src`; returns bytes on every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return blob
    rank = dist.get_rank()
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, src)
    if rank == src:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def allreduce_accum(accum):
    """Sum of the per-rank fp32 radiance accumulators (in place).  RCCL reduces device tensors directly; under gloo a device tensor
    is staged through host memory (the functional N-rank run on a box with fewer GPUs than ranks)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if accum.is_cuda and dist.get_backend() == "gloo":
            host = accum.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            accum.copy_(host)
        else:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM)
    return accum


def render_sharded(render_fn, width, height, spp_total, device, spp_first=0):
    """render_fn(accum_tensor, spp_begin, spp_end) adds this rank's radiance sums into accum_tensor
    (H x W x 3 fp32 on `device`).  Returns the all-reduced accumulator (identical on every rank)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    accum = torch.zeros((height, width, 3), dtype=torch.float32, device=device)
    b, e = spp_range(rank, world, spp_total, spp_first)
    if e > b:
        render_fn(accum, b, e)
    return allreduce_accum(accum)
