#!/usr/bin/env python3
"""Times the operators on the 19-label 1024x2048 11x11 shape (batch 2), set-up included: HIP events, median of 5."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
C, H, W, k, B = 19, 1024, 2048, 11, 2
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, C, H, W, device="cuda", generator=g)
img = torch.randn(B, 3, H, W, device="cuda", generator=g)
dy = torch.randn(B, C, H, W, device="cuda", generator=g)
compat = (1 - torch.eye(C, device="cuda"))
pw = torch.tensor([-0.004], device="cuda"); cw = torch.tensor([-0.003], device="cuda")
def timed(f, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
for name, f in (("local_forward", lambda: S.ops.local_forward(x, pw, cw, k, image=img)),
                ("local_backward", lambda: S.ops.local_backward(x, pw, cw, k, dy, image=img)),
                ("meanfield K=1", lambda: S.ops.meanfield_forward(x, img, pw, cw, k, 1, None, None, compat)),
                ("meanfield K=10", lambda: S.ops.meanfield_forward(x, img, pw, cw, k, 10, None, None, compat))):
    f(); print("%-16s B2 C19 1024x2048 k11: %.3f ms" % (name, timed(f)))
