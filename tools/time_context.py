#!/usr/bin/env python3
"""Times the fused label-context kernel (9x9 convolution 21 -> 105 channels + affines + min over 5 mixtures +
combine) on 21 labels, 512x512, batch 8: whole call, HIP events, median of 5."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import semantic_segmentation_amd as S
B, L, M, H, W, ks = 8, 21, 5, 512, 512, 9
g = torch.Generator(device="cuda").manual_seed(0)
t = torch.randn(B, L, H, W, device="cuda", generator=g)
unary = torch.randn(B, L, H, W, device="cuda", generator=g)
mod = S.LabelContext(H, W, L, M, ks).cuda()
def timed(f, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
with torch.no_grad():
    f = lambda: mod(t, unary)
    f()
    ms = timed(f)
flops = 2.0 * B * H * W * (L * M) * (L * ks * ks)
print(f"label context B{B} L{L} M{M} {H}x{W} k{ks}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s (fp32 MFMA peak 157.3)")
G = torch.randn(B, L, H, W, device="cuda", generator=g)
conv = mod.global_convolution_layer
bw = lambda: S.ops.label_context_backward(t, conv.weight.detach(), unary, grad_logq=G, conv_b=conv.bias.detach(),
                                          la_w=mod.local_activation_function.weight.detach(), la_b=mod.local_activation_function.bias.detach(),
                                          ga_w=mod.global_activation_function.weight.detach(), ga_b=mod.global_activation_function.bias.detach())
bw()
ms = timed(bw)
print(f"label context backward (forward recomputed + data gradient + weight gradient): {ms:.3f} ms  "
      f"{3 * flops / ms / 1e9:.1f} TFLOP/s over its three convolution-sized GEMMs")
with torch.no_grad():
    lq, (am, cm) = S.ops.label_context_forward(t, conv.weight.detach(), unary, conv_b=conv.bias.detach(),
                                               la_w=mod.local_activation_function.weight.detach(), la_b=mod.local_activation_function.bias.detach(),
                                               ga_w=mod.global_activation_function.weight.detach(), ga_b=mod.global_activation_function.bias.detach(), keep=True)
bk = lambda: S.ops.label_context_backward(t, conv.weight.detach(), unary, grad_logq=G, conv_b=conv.bias.detach(),
                                          la_w=mod.local_activation_function.weight.detach(), la_b=mod.local_activation_function.bias.detach(),
                                          ga_w=mod.global_activation_function.weight.detach(), ga_b=mod.global_activation_function.bias.detach(),
                                          kept=(lq, am, cm))
bk()
print(f"label context backward from a kept forward (log-marginals, arg-min mixture, convolution value handed back): {timed(bk):.3f} ms")
