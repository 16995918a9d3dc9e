#!/usr/bin/env python3
"""Debug aid: local forward against the oracle with a break-down of the errors by channel, b-block, group, lane."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
from oracle import oracle as O
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1, 21, 128, 512, 7)]
for shape in shapes:
    B, C, H, W, k = shape
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    img = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    pw = torch.tensor([0.3], device="cuda"); cw = torch.tensor([-0.7], device="cuda")
    y = S.ops.local_forward(torch.from_numpy(x).cuda(), pw, cw, k, image=torch.from_numpy(img).cuda()).cpu().numpy()
    ref = O.local_forward(x, 0.3, -0.7, k, img=img)
    err = np.abs(y - ref) / np.abs(ref).max()
    bad = np.argwhere(err > 1e-4)
    print(shape, "max err", err.max(), "bad", len(bad), "of", err.size)
    if len(bad):
        CK = C * k * k
        l = bad[:, 2] * W + bad[:, 3]
        a, b = l % CK, l // CK
        n = a % k
        lane = (b + n) % 64
        beta = (b + n) // 64
        print("  per channel", np.bincount(bad[:, 1], minlength=C))
        print("  per n", np.bincount(n, minlength=k))
        print("  per beta", np.bincount(beta))
        print("  per lane", np.bincount(lane, minlength=64))
        g = a // k
        print("  groups with errors", len(np.unique(g)), "of", C * k, " first", np.unique(g)[:16])
