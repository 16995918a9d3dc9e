"""Batch sharding across the GPUs of one node (one process per GPU).

The path is independent across the batch index (the reference's reshape keeps dim 0,
LocalConvolution.py:80-81), so the forward needs no collective: rank r of R simply owns
images [r*B/R, (r+1)*B/R).  Training needs one all-reduce (sum) over a flat buffer of
potential-parameter gradients per optimiser step - issued once, after local accumulation,
mirroring the reference's gradient accumulation before optimizer.step()
(AdvancedModel.py:298-303).

The per-pixel affine parameters (LinearActivation, AdvancedModel.py:38-56, 138, 145: 2*L*H*W and 2*L*M*H*W floats,
44 MB + 220 MB at 21 labels / 512x512) are the one place where the exchange is bandwidth-bound: for them
ShardedUpdate reduce-scatters the gradients (every rank receives the sum of its 1/R slice), lets the caller's
optimiser update that slice only, and all-gathers the updated parameters - 2*(R-1)/R of the buffer per rank in
total, like an all-reduce, but over xGMI's direct links both collectives run on all links at once, the
optimiser state and the update itself shrink to 1/R per rank, and the update overlaps nothing it has to wait for.
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


class ShardedUpdate:
    """Data-parallel update of large parameters with reduce-scatter + all-gather instead of an all-reduce.

        sharded = ShardedUpdate(model.local_activation_function.parameters())
        optimizer = torch.optim.SGD([sharded.shard], lr=...)      # optimiser state for 1/R of the parameters only
        ...
        loss.backward()
        sharded.reduce_scatter_gradients()      # sharded.shard.grad = sum over ranks of this rank's slice
        optimizer.step()
        sharded.all_gather_parameters()         # every rank holds the updated parameters again

    The parameters are viewed as one flat buffer, zero-padded to a multiple of the world size; rank r owns
    elements [r*n, (r+1)*n).  With one rank (or no process group) the slice is the whole buffer and the
    collectives are skipped.  A gloo group (CPU tests, rehearsals) has no reduce-scatter: the sum goes through
    an all-reduce and is sliced, which gives the same values."""

    def __init__(self, parameters: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        dev, dtype = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dtype for p in self.params):
            raise ValueError("parameters must share one device and dtype")
        self.numel = sum(p.numel() for p in self.params)
        self.per_rank = (self.numel + self.world - 1) // self.world
        self._flat = torch.zeros(self.per_rank * self.world, dtype=dtype, device=dev)
        with torch.no_grad():
            self._pack([p.detach() for p in self.params])
            begin = self.rank * self.per_rank
            self.shard = torch.nn.Parameter(self._flat[begin:begin + self.per_rank].clone())

    def _pack(self, tensors) -> None:
        offset = 0
        for t in tensors:
            self._flat[offset:offset + t.numel()].copy_(t.reshape(-1))
            offset += t.numel()
        self._flat[offset:].zero_()

    def _staged(self) -> bool:
        return self.world > 1 and self._flat.is_cuda and dist.get_backend(self.group) == "gloo"

    @torch.no_grad()
    def reduce_scatter_gradients(self) -> None:
        """shard.grad <- sum over ranks of the gradient slice this rank owns."""
        self._pack([p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params])
        begin = self.rank * self.per_rank
        if self.world == 1:
            out = self._flat[begin:begin + self.per_rank].clone()
        elif dist.get_backend(self.group) == "gloo":
            buf = self._flat.cpu() if self._flat.is_cuda else self._flat
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            out = buf[begin:begin + self.per_rank].to(self._flat.device).clone()
        else:
            out = torch.empty(self.per_rank, dtype=self._flat.dtype, device=self._flat.device)
            dist.reduce_scatter_tensor(out, self._flat, op=dist.ReduceOp.SUM, group=self.group)
        self.shard.grad = out

    @torch.no_grad()
    def all_gather_parameters(self) -> None:
        """Every rank's parameters <- the concatenation of all ranks' updated slices."""
        if self.world == 1:
            self._flat[:self.per_rank].copy_(self.shard)
        elif dist.get_backend(self.group) == "gloo":
            pieces = [torch.empty(self.per_rank, dtype=self._flat.dtype) for _ in range(self.world)]
            dist.all_gather(pieces, self.shard.detach().cpu(), group=self.group)
            self._flat.copy_(torch.cat(pieces).to(self._flat.device))
        else:
            dist.all_gather_into_tensor(self._flat, self.shard.detach().contiguous(), group=self.group)
        offset = 0
        for p in self.params:
            p.copy_(self._flat[offset:offset + p.numel()].view_as(p))
            offset += p.numel()
