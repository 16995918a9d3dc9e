#!/usr/bin/env python3
"""Times the reference's one shipped configuration (AdvancedModel.py:135-137): 21 labels, 64x64, 50x50 window, one
image: whole calls, HIP events, median of 5."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
g = torch.Generator(device="cuda").manual_seed(99)
unary = 2 * torch.randn(1, 21, 64, 64, device="cuda", generator=g)
img = torch.randn(1, 3, 64, 64, device="cuda", generator=g)
pw = torch.tensor([-0.004], device="cuda"); cw = torch.tensor([-0.003], device="cuda")
def timed(f, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
dy = torch.randn(1, 21, 64, 64, device="cuda", generator=g)
for name, f in (("local_forward", lambda: S.ops.local_forward(unary, pw, cw, 50, image=img)),
                ("local_backward", lambda: S.ops.local_backward(unary, pw, cw, 50, dy, image=img)),
                ("meanfield K=1", lambda: S.ops.meanfield_forward(unary, img, pw, cw, 50, 1))):
    f(); print("%-14s B1 C21 64x64 k50: %.3f ms" % (name, timed(f)))
