#!/usr/bin/env python3
"""Times the single-operator entry points (local forward, local backward, one mean-field step) on the
config 2 shape, set-up included: HIP events, median of 5."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
C, H, W, k, B = 21, 512, 512, 7, 8
rng = np.random.default_rng(0)
x = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).cuda()
img = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
dy = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).cuda()
pw = torch.tensor([-0.004], device="cuda"); cw = torch.tensor([-0.003], device="cuda")
def timed(f, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
f = lambda: S.ops.local_forward(x, pw, cw, k, image=img)
f(); print("local_forward  B8 C21 512^2 k7: %.3f ms" % timed(f))
b = lambda: S.ops.local_backward(x, pw, cw, k, dy, image=img)
b(); print("local_backward B8 C21 512^2 k7: %.3f ms" % timed(b))
m1 = lambda: S.ops.meanfield_forward(x, img, pw, cw, k, 1, None, None, None)
m1(); print("meanfield K=1  B8 C21 512^2 k7: %.3f ms" % timed(m1))
