#!/usr/bin/env python3
"""Throughput of the fused mean-field / local-window path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 21 labels, 512x512, batch 8 per GPU, 7x7 window, 10
mean-field iterations with an LxL label-compatibility matrix, fp32.  One "step" is one call of
mfl_meanfield_forward over the batch (10 iterations); inputs are resident in HBM before the
timed region.  With N GPUs every rank owns its own 8 images (weak scaling, no collective on the
data path); the time is the maximum over ranks between two barriers.

Prints one JSON line: metric = mean-field iter*Mpixel/s over the whole job, plus
  roofline      the fused step kernel against the HBM roofline: algorithmic bytes per launch
                ((12*L + 12) B per pixel-iteration) / its average duration measured with HIP events;
                traffic = memory-side bytes per launch from the committed PMC passes
  cpu_baseline  the CPU oracle (a port of the reference's arithmetic) timed on this host's cores
                on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LABELS, HEIGHT, WIDTH, WINDOW, ITERS, BATCH_PER_GPU = 21, 512, 512, 7, 10, 8
POS_W, COL_W = -0.004, -0.003
HBM_PEAK_GBPS = 8000.0
VALU_PEAK_TFLOPS = 157.0      # fp32 vector peak
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32).reshape(3, 1, 1)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32).reshape(3, 1, 1)


def synthetic_image(index):
    """Unary logits ~ 2*N(0,1) and an ImageNet-normalised U[0,1) image, seeded per global image index."""
    rng = np.random.default_rng(1234 + index)
    unary = (2 * rng.standard_normal((LABELS, HEIGHT, WIDTH))).astype(np.float32)
    img = ((rng.random((3, HEIGHT, WIDTH), dtype=np.float32) - IMAGENET_MEAN) / IMAGENET_STD).astype(np.float32)
    return unary, img


def compat_matrix():
    rng = np.random.default_rng(99)
    return ((1 - np.eye(LABELS)) + 0.01 * rng.standard_normal((LABELS, LABELS))).astype(np.float32)


def cpu_baseline():
    """The oracle on the host cores: 2 images x 10 iterations of the same workload (second call timed)."""
    from oracle import oracle as O
    O.build()
    sample = 2
    data = [synthetic_image(i) for i in range(sample)]
    unary = np.stack([d[0] for d in data])
    img = np.stack([d[1] for d in data])
    compat = compat_matrix()
    O.meanfield_forward(unary[:1], img[:1], POS_W, COL_W, WINDOW, 1, None, None, compat)      # warm-up
    t0 = time.perf_counter()
    O.meanfield_forward(unary, img, POS_W, COL_W, WINDOW, ITERS, None, None, compat)
    dt = time.perf_counter() - t0
    return {"value": round(ITERS * sample * HEIGHT * WIDTH / 1e6 / dt, 4), "unit": "iter*Mpixel/s",
            "cores": O.num_threads(), "kind": "port",
            "sample": f"{sample} of the {BATCH_PER_GPU} images x {ITERS} iterations, C oracle with OpenMP, "
                      f"{dt:.1f} s"}


def measured_traffic():
    """Memory-side bytes per launch of the fused step kernel from the committed PMC passes
    (profiles/traffic.json says how they were collected and corrected); None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        return int(t["traffic_bytes_per_launch"]), t["sources"]
    except (OSError, KeyError, ValueError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import semantic_segmentation_amd as S

    # this rank's shard of the global batch: images [rank*8, rank*8 + 8)
    data = [synthetic_image(rank * BATCH_PER_GPU + i) for i in range(BATCH_PER_GPU)]
    unary = torch.from_numpy(np.stack([d[0] for d in data])).to(device)
    img = torch.from_numpy(np.stack([d[1] for d in data])).to(device)
    compat = torch.from_numpy(compat_matrix()).to(device)
    pw = torch.tensor([POS_W], device=device)
    cw = torch.tensor([COL_W], device=device)

    def step(iters=ITERS):
        return S.ops.meanfield_forward(unary, img, pw, cw, WINDOW, iters, None, None, compat)

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out).all()

    # average duration of one fused step launch, HIP events on the launch stream:
    # (call with ITERS iterations - call with 1 iteration) / (ITERS - 1) removes the one-off
    # padding / table set-up of a call.
    def timed(iters, reps=5):
        best = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step(iters)
            e1.record()
            e1.synchronize()
            best.append(e0.elapsed_time(e1))
        return float(np.median(best))

    t_full, t_one = timed(ITERS), timed(1)
    launch_ms = (t_full - t_one) / (ITERS - 1)
    pixels = BATCH_PER_GPU * HEIGHT * WIDTH
    bytes_per_launch = (12 * LABELS + 12) * pixels
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
    # algorithmic flops per pixel-iteration: gather MACs + weight build + LxL compat + two softmax passes
    flops_per_launch = (2 * LABELS * WINDOW ** 2 + 11 * WINDOW ** 2 + 2 * LABELS ** 2 + 8 * LABELS) * pixels
    tflops = flops_per_launch / (launch_ms * 1e-3) / 1e12

    if rank == 0:
        traffic, traffic_src = measured_traffic()
        total_pixel_iters = ITERS * BATCH_PER_GPU * world * HEIGHT * WIDTH * args.steps
        line = {
            "metric": "mean-field iter*Mpixel/s (node)",
            "value": round(total_pixel_iters / 1e6 / elapsed, 2),
            "unit": "iter*Mpixel/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "AdvancedModel inference: 21 labels, 512x512, batch 8 per GPU, 7x7 window, "
                                   "10 mean-field iterations, LxL compat (BASELINE.json configs[1])",
                       "labels": LABELS, "height": HEIGHT, "width": WIDTH, "window": WINDOW, "iters": ITERS,
                       "batch_per_gpu": BATCH_PER_GPU, "global_batch": BATCH_PER_GPU * world,
                       "parallelism": f"batch-shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "traffic_unit": "bytes per launch (PMC, committed passes)", "traffic_source": traffic_src,
                         "kernel": "diag_kernel (one fused mean-field step: regular tiles + tail workgroups)", "launch_ms": round(launch_ms, 4),
                         "bytes_per_launch": bytes_per_launch,
                         "valu": {"achieved": round(tflops, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(tflops / VALU_PEAK_TFLOPS, 4)}},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
