#!/usr/bin/env python3
"""One rank of the multi-process collective checks: run under `python -m torch.distributed.run`, one process per
rank, started fresh (nothing in the parent's address space, no GPU call before this file runs).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tests/dist_worker.py --backend nccl|gloo [--numel N]

With `nccl` (= RCCL on ROCm) every rank takes GPU LOCAL_RANK and the tensors live there; with `gloo` they are CPU
tensors (the rehearsal the CPU suite runs).  Checks, each against sums this rank computes by itself from the seeds
of all ranks:
  - parallel.allreduce_gradients: one flat all-reduce over parameters of mixed sizes, a frozen one and one without
    a gradient on some ranks;
  - parallel.ShardedUpdate: reduce-scatter of the gradients of two per-pixel affine tensors, SGD on this rank's
    slice, all-gather of the parameters == plain SGD on the summed gradients (a size that the world size does not
    divide).
Prints "rank R ok" and exits 0, or raises."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--numel", type=int, default=21 * 37 * 41)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.backend == "nccl":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        device = torch.device("cpu")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    try:
        from semantic_segmentation_amd.parallel import ShardedUpdate, allreduce_gradients

        def seeded(shape, seed):
            return torch.randn(shape, generator=torch.Generator().manual_seed(seed)).to(device)

        # ---- the flat all-reduce of the potential-parameter gradients
        pw = torch.nn.Parameter(torch.zeros(1, device=device))
        cw = torch.nn.Parameter(torch.zeros(1, device=device))
        compat = torch.nn.Parameter(torch.zeros(21, 21, device=device))
        frozen = torch.nn.Parameter(torch.zeros(3, device=device), requires_grad=False)
        sometimes = torch.nn.Parameter(torch.zeros(2, 2, device=device))           # a gradient on rank 0 only
        pw.grad, cw.grad, compat.grad = seeded((1,), 10 + rank), seeded((1,), 20 + rank), seeded((21, 21), 30 + rank)
        if rank == 0:
            sometimes.grad = torch.ones(2, 2, device=device)
        allreduce_gradients([pw, cw, frozen, compat, sometimes])
        for p, base in ((pw, 10), (cw, 20), (compat, 30)):
            want = sum(seeded(tuple(p.shape), base + r) for r in range(world))
            assert torch.allclose(p.grad, want, atol=1e-5), f"rank {rank}: all-reduce of seed base {base}"
        assert frozen.grad is None and torch.equal(sometimes.grad, torch.ones(2, 2, device=device))

        # ---- reduce-scatter / SGD on the slice / all-gather for the per-pixel affine parameters
        n = args.numel
        wgt = torch.nn.Parameter(seeded((n,), 1))
        bias = torch.nn.Parameter(seeded((n + 3,), 2))
        w0, b0 = wgt.detach().clone(), bias.detach().clone()
        wgt.grad, bias.grad = seeded((n,), 100 + rank), seeded((n + 3,), 200 + rank)
        total_w = sum(seeded((n,), 100 + r) for r in range(world))
        total_b = sum(seeded((n + 3,), 200 + r) for r in range(world))
        sharded = ShardedUpdate([wgt, bias])
        assert sharded.shard.numel() == (2 * n + 3 + world - 1) // world
        opt = torch.optim.SGD([sharded.shard], lr=0.1)
        for _ in range(2):                                   # twice: the buffers are reused between steps
            sharded.reduce_scatter_gradients()
            opt.step()
            sharded.all_gather_parameters()
        assert torch.allclose(wgt.detach(), w0 - 0.2 * total_w, atol=1e-5), f"rank {rank}: sharded update of the weight"
        assert torch.allclose(bias.detach(), b0 - 0.2 * total_b, atol=1e-5), f"rank {rank}: sharded update of the bias"
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        print(f"rank {rank} ok", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
