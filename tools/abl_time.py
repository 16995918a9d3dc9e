#!/usr/bin/env python3
"""Time one fused mean-field launch (21 labels, 512x512, batch 8, 7x7) with the library named by
MFL_LIBRARY: (10-iteration call - 1-iteration call) / 9, HIP events, median of 5."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
C, H, W, k, B = [int(v) for v in os.environ.get("ABL_SHAPE", "21,512,512,7,8").split(",")]
rng = np.random.default_rng(0)
unary = torch.from_numpy((2 * rng.standard_normal((B, C, H, W))).astype(np.float32)).cuda()
img = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
compat = torch.from_numpy((1 - np.eye(C)).astype(np.float32)).cuda() if os.environ.get("ABL_COMPAT", "1") == "1" else None
pw = torch.tensor([-0.004], device="cuda"); cw = torch.tensor([-0.003], device="cuda")
def timed(iters):
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); S.ops.meanfield_forward(unary, img, pw, cw, k, iters, None, None, compat); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
timed(2)
ms = (timed(10) - timed(1)) / 9
print(f"{os.path.basename(os.environ.get('MFL_LIBRARY', 'default')):24s} compat={os.environ.get('ABL_COMPAT','1')} {ms:8.4f} ms/launch  {B*H*W/1e3/ms:9.1f} iter*Mpixel/s")
