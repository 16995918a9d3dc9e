#!/usr/bin/env python3
"""Measured parity figures of the HIP path against the C oracle, printed (not merely tolerated): the local operator's
forward and backward at the reference's initial weight scale and at the stress weights (0.3, -0.7), and the mean-field
loop's strict elementwise figure on the marginals.  Per case: max |a - e| / max |e| ("to scale") and, where the
reference value is not tiny, the strict elementwise max |a - e| / |e|.  Run on the GPU box:
    python tools/report_parity.py > gpurun_out/parity_figures.txt"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import semantic_segmentation_amd as S
from oracle import oracle as O
from test_gpu_parity import make_inputs, dev, host, scalar, strict_rel_err_on_marginals

O.build()


def figures(a, e):
    a, e = np.asarray(a, np.float64), np.asarray(e, np.float64)
    big = np.abs(e) > 1e-3 * np.abs(e).max()
    return np.abs(a - e).max() / np.abs(e).max(), (np.abs(a - e)[big] / np.abs(e)[big]).max()


print("local operator (Conv2dLocal forward / backward), HIP vs C oracle (double accumulation)")
print("%-28s %-16s %12s %12s %12s %12s %10s %10s" % ("shape B,C,H,W,k", "weights", "y to scale", "y strict", "dx to scale", "dx strict", "d pw rel", "d cw rel"))
for shape in [(2, 21, 96, 112, 7), (1, 19, 96, 128, 11), (1, 21, 64, 64, 50), (2, 21, 512, 512, 7), (1, 19, 1024, 2048, 11), (1, 40, 96, 100, 3)]:
    x, img, dy = make_inputs(shape)
    k = shape[4]
    for pw, cw in ((-0.004, -0.003), (0.3, -0.7)):
        y = host(S.ops.local_forward(dev(x), scalar(pw), scalar(cw), k, image=dev(img)))
        dx, dpar, _ = S.ops.local_backward(dev(x), scalar(pw), scalar(cw), k, dev(dy), image=dev(img))
        ry = O.local_forward(x, pw, cw, k, img=img)
        rdx, rdpw, rdcw = O.local_backward(x, pw, cw, k, dy, img=img)
        fy, fdx = figures(y, ry), figures(host(dx), rdx)
        print("%-28s %-16s %12.2e %12.2e %12.2e %12.2e %10.1e %10.1e" % (
            ",".join(map(str, shape)), f"({pw}, {cw})", fy[0], fy[1], fdx[0], fdx[1],
            abs(float(dpar[0]) - rdpw) / abs(rdpw), abs(float(dpar[1]) - rdcw) / abs(rdcw)), flush=True)

print()
print("mean-field loop, LxL matrix: strict figure max |Q - Q_ref| / Q_ref over the marginals Q_ref > 1e-6")
print("%-28s %-16s %6s %12s" % ("shape B,C,H,W,k", "weights", "iters", "strict"))
for shape in [(2, 21, 96, 112, 7), (1, 19, 96, 128, 11), (1, 21, 512, 512, 7), (1, 19, 512, 1024, 11)]:
    unary, img, _ = make_inputs(shape, 2)
    C, k = shape[1], shape[4]
    compat = ((1 - np.eye(C)) + 0.01 * np.random.default_rng(7).standard_normal((C, C))).astype(np.float32)
    for pw, cw in ((-0.004, -0.003), (-0.04, -0.03)):
        for iters in (1, 5, 10):
            lq = host(S.ops.meanfield_forward(dev(unary), dev(img), scalar(pw), scalar(cw), k, iters, None, None, dev(compat)))
            ref = O.meanfield_forward(unary, img, pw, cw, k, iters, None, None, compat)
            print("%-28s %-16s %6d %12.2e" % (",".join(map(str, shape)), f"({pw}, {cw})", iters, strict_rel_err_on_marginals(lq, ref)), flush=True)
