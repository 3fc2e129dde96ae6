"""Image-per-GPU sharding helpers: images are independent, so ranks never exchange data on the decode path.
The only cross-rank steps are a barrier around the timed region and a MAX-reduce of the elapsed time."""


def rank_seed_base(rank: int) -> int:
    """Distinct synthetic-image seeds per rank (image i of rank r uses rank_seed_base(r) + i)."""
    return 1 + rank * 1000003


def shard_indices(n_items: int, rank: int, world: int):
    """Round-robin shard of a global list: item i goes to rank i % world (image i -> GPU i mod G).
    The same rule as the C ABI's jpgpu_shard (include/jpgpu.h), which it calls."""
    import ctypes as C

    from . import _capi

    first, stride, count = C.c_int(), C.c_int(), C.c_int()
    _capi.lib.jpgpu_shard(n_items, rank, world, C.byref(first), C.byref(stride), C.byref(count))
    return [first.value + k * stride.value for k in range(count.value)]


def max_over_ranks(dist, value: float, device=None) -> float:
    """MAX all-reduce of a host float (nccl needs a device tensor, gloo a CPU one)."""
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(pixels_per_rank_step: int, world: int, steps: int, elapsed_max: float) -> float:
    """Whole-job Mpixels/s: units processed by all ranks / slowest rank's time."""
    return pixels_per_rank_step * world * steps / elapsed_max / 1e6
