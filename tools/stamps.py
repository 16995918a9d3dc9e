#!/usr/bin/env python3
"""Per-phase cycle stamps of the diagonal kernel (diagnostic build libmfl_stamps.so, -DMFL_STAMPS):
median over the first 256 workgroups of the time between stamps, in microseconds (100 MHz s_memtime... the
counter runs at the shader clock on gfx950; converted with the measured kernel duration)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
from semantic_segmentation_amd import _lib
C, H, W, k, B = [int(v) for v in os.environ.get("ABL_SHAPE", "21,512,512,7,8").split(",")]
rng = np.random.default_rng(0)
unary = torch.from_numpy((2 * rng.standard_normal((B, C, H, W))).astype(np.float32)).cuda()
img = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
compat = torch.from_numpy((1 - np.eye(C)).astype(np.float32)).cuda()
pw = torch.tensor([-0.004], device="cuda"); cw = torch.tensor([-0.003], device="cuda")
dbg = torch.zeros(256 * 64, dtype=torch.int64, device="cuda")
S.ops.meanfield_forward(unary, img, pw, cw, k, 2, None, None, compat)
torch.cuda.synchronize()
_lib.lib.mfl_debug_stamps.argtypes = [ctypes.c_void_p]
_lib.lib.mfl_debug_stamps(ctypes.c_void_p(dbg.data_ptr()))
S.ops.meanfield_forward(unary, img, pw, cw, k, 1, None, None, compat)
torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(256, 64).astype(np.float64)
names = {0: "start", 1: "setup", 2: "W0+bar", 40: "loop end", 41: "end", 42: "c0 write", 43: "c0 bar", 44: "c0 pixels", 45: "c0 bar2", 46: "c1 write", 47: "c1 bar", 48: "c1 pixels", 49: "c1 bar2"}
segs = d[:, 54:60].copy(); d[:, 54:60] = 0
idx = sorted([i for i in range(64) if d[:, i].min() > 0], key=lambda i: np.median(d[:, i] - d[:, 0]))
t0 = d[:, 0:1]
rel = (d - t0)
prev = 0
print("stamp  name        median_cum   median_delta  (ticks)")
for i in idx:
    cum = np.median(rel[:, i]); 
    g = (i - 3) // 3 if 3 <= i < 40 else -1
    nm = names.get(i, f"g{g}:" + ["W(next)", "gather", "barrier"][(i - 3) % 3])
    print(f"{i:4d}  {nm:12s} {cum:12.0f} {cum - prev:12.0f}")
    prev = cum
print("tap segments (median ticks per workgroup, thread 0): w+record %.0f  corrections %.0f  slot loads %.0f  FMAs+reloads %.0f  between taps %.0f" % tuple(np.median(segs[:, [0, 1, 2, 3, 5]], axis=0)))
