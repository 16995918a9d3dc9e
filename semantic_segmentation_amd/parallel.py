"""Batch sharding across the GPUs of one node (one process per GPU).

The path is independent across the batch index (the reference's reshape keeps dim 0,
LocalConvolution.py:80-81), so the forward needs no collective: rank r of R simply owns
images [r*B/R, (r+1)*B/R).  Training needs one all-reduce (sum) over a flat buffer of
potential-parameter gradients per optimiser step - issued once, after local accumulation,
mirroring the reference's gradient accumulation before optimizer.step()
(AdvancedModel.py:298-303).
"""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def shard_range(batch: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of a global batch for `rank`; remainders go to the first ranks."""
    if batch < 0 or world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("invalid shard request")
    base, extra = divmod(batch, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def allreduce_gradients(parameters: Iterable[torch.nn.Parameter], group=None) -> None:
    """Sum the gradients of `parameters` over all ranks with ONE collective on a flat buffer.

    Parameters without a gradient contribute zeros so that every rank reduces the same layout.
    """
    params: List[torch.nn.Parameter] = [p for p in parameters if p.requires_grad]
    if not params or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    if flat.is_cuda and dist.get_backend(group) == "gloo":
        # a CPU-side process group (tests, rehearsals): stage the flat buffer through the host
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    offset = 0
    for p in params:
        n = p.numel()
        g = flat[offset:offset + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        offset += n
