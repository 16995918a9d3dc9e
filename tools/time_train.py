#!/usr/bin/env python3
"""Times one training step of MeanFieldLocal (fused forward, NLL, loop backward) on 21 labels, 512x512, 7x7,
batch 8, for K = 1 and K = 5: HIP events, median of 5."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
B, C, H, W, k = 8, 21, 512, 512, 7
g = torch.Generator(device="cuda").manual_seed(0)
unary = 2 * torch.randn(B, C, H, W, device="cuda", generator=g)
img = torch.randn(B, 3, H, W, device="cuda", generator=g)
labels = torch.randint(0, C, (B, H, W), device="cuda", generator=g)
def timed(f, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for K in (1, 5):
    for affine in (False, True):
        model = S.MeanFieldLocal(H, W, C, k, iters=K, compat=True, affine=affine).cuda()
        def step():
            model.zero_grad(set_to_none=True)
            loss = torch.nn.functional.nll_loss(model(unary, img), labels)
            loss.backward()
        step()
        print(f"training step K={K} affine={affine}: {timed(step):.2f} ms")
