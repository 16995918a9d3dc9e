#!/usr/bin/env python3
"""Throughput of the fused mean-field / local-window path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own N rank processes
(python -m torch.distributed.run, 127.0.0.1) before anything touches the GPU; launched under
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

Headline workload (BASELINE.json configs[3], the one the metric and the 60 % target are quoted on):
19 labels, 1024x2048, global batch 16, 11x11 window, 10 mean-field iterations with an LxL label
compatibility matrix, fp32.  One "step" is one call of mfl_meanfield_forward over the rank's images
(10 iterations); inputs are resident in HBM before the timed region.  The global batch of 16 is
split over the ranks (16 / 8 / 4 / 2 images per GPU at 1 / 2 / 4 / 8 GPUs: strong scaling, no
collective on the data path); the time is the maximum over ranks between two barriers.

Prints one JSON line: metric = mean-field iter*Mpixel/s over the whole job, plus
  roofline       the fused step kernel against the HBM roofline: algorithmic bytes per launch
                 ((12*L + 12) B per pixel-iteration) / its average duration measured with HIP events on
                 the launch stream; `valu` is the same launch against the fp32 vector peak;
                 traffic = memory-side bytes per launch from the committed PMC passes
  cpu_baseline   the CPU oracle (a port of the reference's arithmetic) timed on this host's cores on a
                 512x1024 crop of image 0 (rank 0, N = 1 only), median of 5
  other_configs  short measurements of BASELINE.json configs[1] and configs[4] (N = 1 only) and of
                 configs[2] (training step: forward, NLL with 255-ignore, backward, one flat all-reduce
                 of the potential-parameter gradients; at N = 1 its parameter gradients for image 0 are checked
                 against the CPU oracle outside the timed region) and, at N > 1, of the reduce-scatter /
                 all-gather exchange of the per-pixel affine parameters
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

POS_W, COL_W = -0.004, -0.003
HBM_PEAK_GBPS = 8000.0
VALU_PEAK_TFLOPS = 157.3      # fp32 vector peak
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

CONFIGS = {
    "c4": dict(ref="BASELINE.json configs[3]", what="Cityscapes-shaped inference", labels=19, height=1024, width=2048,
               window=11, iters=10, batch=16),
    "c2": dict(ref="BASELINE.json configs[1]", what="AdvancedModel inference", labels=21, height=512, width=512,
               window=7, iters=10, batch=8),
    "c5": dict(ref="BASELINE.json configs[4]", what="large-label stress", labels=150, height=512, width=512,
               window=7, iters=10, batch=16),
    "c3": dict(ref="BASELINE.json configs[2]", what="AdvancedModel training fwd+bwd", labels=21, height=512,
               width=512, window=7, iters=1, batch=32),
}


def synthetic_image(cfg, index, device):
    """Unary logits ~ 2*N(0,1) and an ImageNet-normalised U[0,1) image, seeded per GLOBAL image index
    (so any batch shard regenerates its own images identically), generated on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 + index)
    unary = 2 * torch.randn(cfg["labels"], cfg["height"], cfg["width"], generator=g, device=device)
    img = torch.rand(3, cfg["height"], cfg["width"], generator=g, device=device)
    mean = torch.tensor(IMAGENET_MEAN, device=device).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD, device=device).view(3, 1, 1)
    return unary, (img - mean) / std


def compat_matrix(labels):
    rng = np.random.default_rng(99)
    return ((1 - np.eye(labels)) + 0.01 * rng.standard_normal((labels, labels))).astype(np.float32)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, unary0, img0, crop=(512, 1024), repeats=5):
    """The oracle on the host cores: the mean-field loop of the same workload on a crop of image 0
    (second-call timing, median of `repeats`)."""
    from oracle import oracle as O
    O.build()
    h, w = min(crop[0], cfg["height"]), min(crop[1], cfg["width"])
    unary = unary0[:, :h, :w].contiguous().cpu().numpy()[None]
    img = img0[:, :h, :w].contiguous().cpu().numpy()[None]
    compat = compat_matrix(cfg["labels"])
    k, iters = cfg["window"], cfg["iters"]
    O.meanfield_forward(unary, img, POS_W, COL_W, k, 1, None, None, compat)      # warm-up
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        O.meanfield_forward(unary, img, POS_W, COL_W, k, iters, None, None, compat)
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts))
    return {"value": round(iters * h * w / 1e6 / dt, 4), "unit": "iter*Mpixel/s", "cores": O.num_threads(),
            "kind": "port", "cpu_model": cpu_model(), "host_cpus": os.cpu_count(),
            "sample": f"{h}x{w} crop of image 0, {cfg['labels']} labels, {k}x{k} window, {iters} iterations, "
                      f"C oracle with OpenMP on {O.num_threads()} threads, median of {repeats} runs = {dt:.2f} s "
                      f"(total {sum(ts):.0f} s)"}


def measured_traffic(key, pixels):
    """Memory-side bytes per launch of the fused step kernel from the committed PMC passes
    (profiles/traffic.json says how they were collected and corrected), scaled from the pixels per launch of
    that measurement to this rank's; None when absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)[key]
        return int(t["traffic_bytes_per_launch"] * (pixels / t["pixels_per_launch"])), t["sources"]
    except (OSError, KeyError, ValueError, TypeError):
        return None, None


def self_launch(args):
    """Start one rank per GPU; nothing in this process has touched the GPU yet."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.only:
        cmd += ["--only", args.only]
    if args.backend != "nccl":
        cmd += ["--backend", args.backend]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


class Job:
    """Rank, device and the collectives the timing needs."""

    def __init__(self, gpus, backend="nccl"):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != gpus:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={self.world}")
        # one GPU per rank; with fewer devices than ranks (a rehearsal of the launcher on a one-GPU box,
        # --backend gloo) the ranks share devices
        local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        self.backend = backend
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
            else:
                dist.init_process_group(backend, rank=self.rank, world_size=self.world)

    def fence(self):
        torch.cuda.synchronize(self.device)
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize(self.device)

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1:
            dist.destroy_process_group()


def event_ms(fn, reps=5):
    """Median duration of fn() measured with HIP events on torch's current stream (the launch stream)."""
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def run_inference(job, key, steps, warmup, with_cpu):
    """Strong-scaled inference of CONFIGS[key]: returns the fields of one result (rank 0) or None."""
    import semantic_segmentation_amd as S
    cfg = CONFIGS[key]
    L, H, W, k, iters, batch = cfg["labels"], cfg["height"], cfg["width"], cfg["window"], cfg["iters"], cfg["batch"]
    lo, hi = S.parallel.shard_range(batch, job.rank, job.world)
    mine = hi - lo
    dev = job.device
    data = [synthetic_image(cfg, i, dev) for i in range(lo, hi)]
    unary = torch.stack([d[0] for d in data])
    img = torch.stack([d[1] for d in data])
    del data
    compat = torch.from_numpy(compat_matrix(L)).to(dev)
    pw = torch.tensor([POS_W], device=dev)
    cw = torch.tensor([COL_W], device=dev)

    def step(n_iters=iters):
        return S.ops.meanfield_forward(unary, img, pw, cw, k, n_iters, None, None, compat)

    out = None
    for _ in range(warmup):
        out = step()
    job.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    job.fence()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(out).all()
    del out

    # average duration of one fused step launch: (call with `iters` iterations - call with 1 iteration)
    # / (iters - 1) removes the one-off set-up of a call
    t_full, t_one = event_ms(step), event_ms(lambda: step(1))
    launch_ms = (t_full - t_one) / (iters - 1)
    pixels = mine * H * W
    bytes_per_launch = (12 * L + 12) * pixels
    achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
    # algorithmic flops per pixel-iteration: gather MACs + weight build + LxL compat + two softmax passes
    flops_per_launch = (2 * L * k * k + 11 * k * k + 2 * L * L + 8 * L) * pixels
    tflops = flops_per_launch / (launch_ms * 1e-3) / 1e12
    if job.rank != 0:
        return None
    traffic, traffic_src = measured_traffic(key, pixels)
    res = {
        "value": round(iters * batch * H * W * steps / 1e6 / elapsed, 2),
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "config": {"workload": f"{cfg['what']}: {L} labels, {H}x{W}, global batch {batch}, {k}x{k} window, "
                               f"{iters} mean-field iterations, LxL compat ({cfg['ref']})",
                   "labels": L, "height": H, "width": W, "window": k, "iters": iters, "global_batch": batch,
                   "batch_per_gpu": mine, "parallelism": f"batch-shard x{job.world}"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "traffic_unit": "bytes per launch (PMC, committed passes)", "traffic_source": traffic_src,
                     "kernel": "fused mean-field step (one launch per iteration over the rank's images)",
                     "launch_ms": round(launch_ms, 4), "bytes_per_launch": bytes_per_launch,
                     "bytes_per_pixel_iteration": 12 * L + 12, "pixels_per_launch": pixels,
                     "valu": {"achieved": round(tflops, 2), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": round(tflops / VALU_PEAK_TFLOPS, 4)}},
    }
    if with_cpu:
        res["cpu_baseline"] = cpu_baseline(cfg, unary[0], img[0])
    return res


def run_training(job, key, steps, warmup, check=False):
    """CONFIGS[key] as a training step on the rank's shard: forward (mean-field module), NLL with
    255-ignore (AdvancedModel.py:164, 298), backward, ONE flat all-reduce of the potential-parameter
    gradients (AdvancedModel.py:298-303 accumulates before optimizer.step())."""
    import semantic_segmentation_amd as S
    cfg = CONFIGS[key]
    L, H, W, k, iters, batch = cfg["labels"], cfg["height"], cfg["width"], cfg["window"], cfg["iters"], cfg["batch"]
    lo, hi = S.parallel.shard_range(batch, job.rank, job.world)
    dev = job.device
    data = [synthetic_image(cfg, i, dev) for i in range(lo, hi)]
    unary = torch.stack([d[0] for d in data])
    img = torch.stack([d[1] for d in data])
    del data
    g = torch.Generator(device=dev)
    g.manual_seed(4321 + job.rank)
    labels = torch.randint(0, L, (hi - lo, H, W), generator=g, device=dev)
    labels[torch.rand(hi - lo, H, W, generator=g, device=dev) < 0.05] = 255
    model = S.MeanFieldLocal(H, W, L, k, iters=iters, compat=True).to(dev)
    with torch.no_grad():
        model.local_convolution_layer.position_weight.fill_(POS_W)
        model.local_convolution_layer.color_weight.fill_(COL_W)
        model.compat.copy_(torch.from_numpy(compat_matrix(L)).to(dev))
    loss_fn = torch.nn.NLLLoss(ignore_index=255)

    def step():
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model(unary, img), labels) * ((hi - lo) / batch)
        loss.backward()
        S.parallel.allreduce_gradients(model.parameters())
        return loss

    for _ in range(warmup):
        step()
    job.fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    job.fence()
    elapsed = job.max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(loss).all()
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert all(torch.isfinite(gr).all() for gr in grads)
    if job.rank != 0:
        return None
    res = {"workload": f"{cfg['what']}: {L} labels, {H}x{W}, global batch {batch}, {k}x{k} window, "
                       f"{iters} mean-field iteration(s), fwd + NLL(255-ignore) + bwd + flat all-reduce ({cfg['ref']})",
           "value": round(batch * H * W * steps / 1e6 / elapsed, 2), "unit": "Mpixel/s (fwd+bwd)",
           "ms_per_step": round(elapsed / steps * 1e3, 3), "steps": steps, "batch_per_gpu": hi - lo,
           "allreduce_floats": int(sum(gr.numel() for gr in grads))}
    if check:
        res["oracle_check"] = training_oracle_check(model, cfg, unary[:1], img[:1], labels[:1], loss_fn)
    return res


def training_oracle_check(model, cfg, unary, img, labels, loss_fn):
    """Outside the timed region: the gradients of the two potential parameters (and of the label matrix) for the loss
    of image 0 alone, from the library's backward and from the CPU oracle's restatement of the same loop backward
    (AdvancedModel.py:222-244, 298-299).  Raises if they disagree beyond 2e-4."""
    from oracle import oracle as O
    O.build()
    k, iters = cfg["window"], cfg["iters"]
    model.zero_grad(set_to_none=True)
    logq = model(unary, img)
    logq.retain_grad()
    loss_fn(logq, labels).backward()
    local = model.local_convolution_layer
    got = {"dpw": float(local.position_weight.grad), "dcw": float(local.color_weight.grad)}
    ref = O.meanfield_backward(unary.cpu().numpy(), img.cpu().numpy(), POS_W, COL_W, k, iters, logq.grad.cpu().numpy(), None, None,
                               model.compat.detach().cpu().numpy())
    err = {key: abs(got[key] - ref[key]) / (abs(ref[key]) + 1e-30) for key in got}
    dc = model.compat.grad.cpu().numpy().astype(np.float64)
    err["dcompat"] = float(np.abs(dc - ref["dcompat"]).max() / (np.abs(ref["dcompat"]).max() + 1e-30))
    if max(err.values()) > 2e-4:
        raise SystemExit(f"config 3: the library's parameter gradients differ from the oracle's: {err}")
    return {"what": "d position_weight, d color_weight, d compat of image 0's loss against the CPU oracle",
            "d_position_weight": got["dpw"], "d_color_weight": got["dcw"],
            "relative_error": {key: float(f"{v:.2e}") for key, v in err.items()}}


def run_exchange(job, reps=5):
    """The bandwidth-bound exchange of data-parallel training: the per-pixel affine parameters of the reference's
    model at 21 labels / 512x512 (local_activation_function 2*L*H*W floats = 44 MB, global_activation_function
    2*L*M*H*W = 220 MB; AdvancedModel.py:38-56, 138, 145) through parallel.ShardedUpdate: reduce-scatter of the
    gradients + all-gather of the parameters per step.  Reports the time per step (max over ranks) and the bytes
    each rank sends per second."""
    import semantic_segmentation_amd as S
    L, M, H, W = 21, 5, 512, 512
    dev = job.device
    params = [torch.nn.Parameter(torch.zeros(n, device=dev)) for n in (L * H * W, L * H * W, L * M * H * W, L * M * H * W)]
    for p in params:
        p.grad = torch.ones_like(p)
    sharded = S.parallel.ShardedUpdate(params)

    def step():
        sharded.reduce_scatter_gradients()
        with torch.no_grad():
            sharded.shard.add_(sharded.shard.grad, alpha=-1e-3)      # the optimiser's update of this rank's slice
        sharded.all_gather_parameters()

    step()
    job.fence()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    job.fence()
    elapsed = job.max_over_ranks(time.perf_counter() - t0) / reps
    want = -1e-3 * (reps + 1) * job.world
    assert abs(float(params[0].detach()[0]) - want) < 1e-6 * abs(want) + 1e-9, "sharded update gave the wrong parameters"
    if job.rank != 0:
        return None
    nbytes = sharded.numel * 4
    sent = 2 * nbytes * (job.world - 1) / job.world            # reduce-scatter + all-gather, per rank
    return {"workload": "per-pixel affine parameters, 21 labels / 5 mixtures / 512x512: reduce-scatter of the gradients, "
                        f"update of 1/N of the parameters, all-gather ({'RCCL' if job.backend == 'nccl' else job.backend})",
            "buffer_MB": round(nbytes / 1e6, 1), "ms_per_step": round(elapsed * 1e3, 3),
            "sent_GBps_per_rank": round(sent / elapsed / 1e9, 1)}


def run_reference_default(job):
    """The one configuration the reference ships (AdvancedModel.py:135-137): 21 labels, 64x64, 50x50 window,
    one image.  Whole-call times (set-up included), HIP events, median of 5."""
    import semantic_segmentation_amd as S
    dev = job.device
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    unary = 2 * torch.randn(1, 21, 64, 64, generator=g, device=dev)
    img = torch.randn(1, 3, 64, 64, generator=g, device=dev)
    pw = torch.tensor([POS_W], device=dev)
    cw = torch.tensor([COL_W], device=dev)
    fwd = lambda: S.ops.local_forward(unary, pw, cw, 50, image=img)
    mf = lambda: S.ops.meanfield_forward(unary, img, pw, cw, 50, 1)
    fwd(), mf()
    return {"workload": "the reference's shipped model: 21 labels, 64x64, 50x50 window, batch 1 (AdvancedModel.py:135-137)",
            "local_forward_ms": round(event_ms(fwd), 4), "meanfield_1_iteration_ms": round(event_ms(mf), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the default) or gloo (launcher rehearsal)")
    ap.add_argument("--only", default="", help="run just the headline workload as this config (c2, c4, c5): "
                                               "profiling aid, no other_configs")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    job = Job(args.gpus, args.backend)
    head = args.only or "c4"
    main_res = run_inference(job, head, args.steps, args.warmup, job.world == 1 and not args.no_cpu_baseline)
    others = {}
    if not args.only:
        torch.cuda.empty_cache()
        r = run_training(job, "c3", steps=5, warmup=2, check=job.world == 1 and not args.no_cpu_baseline)
        if r:
            others["config3_training"] = r
        if job.world > 1:
            torch.cuda.empty_cache()
            r = run_exchange(job)
            if r:
                others["affine_parameter_exchange"] = r
        if job.world == 1:
            for key, st, wu in (("c2", 5, 2), ("c5", 2, 1)):
                torch.cuda.empty_cache()
                r = run_inference(job, key, st, wu, False)
                others["config2" if key == "c2" else "config5"] = {
                    "workload": r["config"]["workload"], "value": r["value"], "unit": "iter*Mpixel/s",
                    "ms_per_step": r["ms_per_step"], "steps": st, "launch_ms": r["roofline"]["launch_ms"],
                    "hbm_frac": r["roofline"]["frac"], "valu_frac": r["roofline"]["valu"]["frac"]}
            others["reference_default_k50"] = run_reference_default(job)
    if job.rank == 0:
        line = {"metric": "mean-field iter*Mpixel/s (node)", "value": main_res["value"], "unit": "iter*Mpixel/s",
                "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": main_res["config"],
                "roofline": main_res["roofline"]}
        if "cpu_baseline" in main_res:
            line["cpu_baseline"] = main_res["cpu_baseline"]
        if others:
            line["other_configs"] = others
        print(json.dumps(line), flush=True)
    job.close()


if __name__ == "__main__":
    main()
