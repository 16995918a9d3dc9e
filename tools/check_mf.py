#!/usr/bin/env python3
"""Debug aid: fused mean-field call against the oracle (all four combinations of affine / label matrix):
elementwise relative error of Q = exp(logq) where Q > 1e-6.  Shapes B,C,H,W,k,iters as arguments."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
from oracle import oracle as O
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(1, 21, 96, 512, 7, 2), (2, 19, 64, 640, 11, 2), (1, 5, 64, 200, 3, 3)]
worst = 0.0
for shape in shapes:
    B, C, H, W, k, iters = shape
    rng = np.random.default_rng(2)
    unary = (2 * rng.standard_normal((B, C, H, W))).astype(np.float32)
    img = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    aw = (1 + 0.1 * rng.standard_normal((C, H, W))).astype(np.float32)
    ab = (0.1 * rng.standard_normal((C, H, W))).astype(np.float32)
    compat = (rng.standard_normal((C, C)) * 0.5).astype(np.float32)
    for use_aff in (False, True):
        for use_compat in (False, True):
            t = lambda a: torch.from_numpy(a).cuda()
            got = S.ops.meanfield_forward(t(unary), t(img), torch.tensor([-0.04], device="cuda"), torch.tensor([-0.03], device="cuda"), k, iters,
                                          t(aw) if use_aff else None, t(ab) if use_aff else None, t(compat) if use_compat else None).cpu().numpy()
            ref = O.meanfield_forward(unary, img, -0.04, -0.03, k, iters, aw if use_aff else None, ab if use_aff else None,
                                      compat if use_compat else None)
            q, qr = np.exp(got.astype(np.float64)), np.exp(ref.astype(np.float64))
            m = qr > 1e-6
            rel = np.abs(q - qr)[m] / qr[m]
            print(shape, "affine", use_aff, "compat", use_compat, "max rel err on Q", rel.max(), "nan", int(np.isnan(got).sum()))
            worst = max(worst, float(rel.max()) if not np.isnan(got).any() else 1.0)
print("WORST", worst)
sys.exit(0 if worst < 1e-4 else 1)
