#!/usr/bin/env python3
"""Debug aid: local forward of the library named by MFL_LIBRARY against the oracle; reports where errors sit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
from oracle import oracle as O
for shape in [(1, 21, 64, 96, 7), (1, 21, 40, 600, 7), (1, 21, 128, 512, 7)]:
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
        print("  channels", np.unique(bad[:, 1])[:30])
        print("  a%k", np.bincount(a % k, minlength=k), " group a//k (first)", np.unique(a // k)[:20])
        print("  b", np.unique(b)[:40])
