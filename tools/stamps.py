#!/usr/bin/env python3
"""Per-segment cycle stamps of the ring kernel (diagnostic build libmfl_stamps.so, -DMFL_STAMPS): median over
the first waves of the cycles spent per tap segment (summed over the taps of one launch)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
from semantic_segmentation_amd import _lib
C, H, W, k, B = [int(v) for v in os.environ.get("ABL_SHAPE", "21,512,512,7,8").split(",")]
NW = int(os.environ.get("ABL_NW", "4"))
rng = np.random.default_rng(0)
unary = torch.from_numpy((2 * rng.standard_normal((B, C, H, W))).astype(np.float32)).cuda()
img = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
compat = torch.from_numpy((1 - np.eye(C)).astype(np.float32)).cuda()
pw = torch.tensor([-0.004], device="cuda"); cw = torch.tensor([-0.003], device="cuda")
dbg = torch.zeros(512 * NW * 48, dtype=torch.int64, device="cuda")
S.ops.meanfield_forward(unary, img, pw, cw, k, 2, None, None, compat)
torch.cuda.synchronize()
_lib.lib.mfl_debug_stamps.argtypes = [ctypes.c_void_p]
_lib.lib.mfl_debug_stamps(ctypes.c_void_p(dbg.data_ptr()))
S.ops.meanfield_forward(unary, img, pw, cw, k, 1, None, None, compat)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(512 * NW, 48).astype(np.float64)
d = d[d[:, 6] > 0]
names = ["head (weights read, record, image loads issued, slot 0 ready)", "gather steps", "image wait", "weight store + lgkm",
         "barrier", "between taps", "whole loop", "epilogue"]
K = k * k
for i, nm in enumerate(names):
    print(f"{nm:62s} median {np.median(d[:, i]):10.0f} cycles  per tap {np.median(d[:, i]) / K:8.1f}")
print("waves", len(d), " start spread (cycles)", np.ptp(d[:, 8]))
es = d[:, 16:40]
print("epilogue stamps (x10 ns after loop end; per chunk: start, transposed, barrier passed, pass done):")
print(np.array2string(np.median(es, axis=0), precision=0, max_line_width=150))
