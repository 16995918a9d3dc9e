"""K-iteration mean-field loop built on the local-window operator.

Per iteration (every step is one of the reference's modules, only the loop is new):
    M    = Conv2dLocal(Q_t, colour distance)                 LocalConvolution.py:59-90
    M    = affine_w * M + affine_b          (optional)        AdvancedModel.py:52-56, 223
    P    = compat . M                       (optional, LxL)   AdvancedModel.py:143-144 with a 1x1 kernel
    logQ = LogSoftmax(log_softmax(U) - P)                     AdvancedModel.py:242-244
    Q_{t+1} = exp(logQ),  Q_0 = U                             AdvancedModel.py:222

Inference (no gradient required) runs the fused C-ABI loop.  When a gradient is required the
same arithmetic is composed per iteration from the HIP local operator (with its HIP backward)
and stock differentiable torch ops for the normalisation.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import nn
from torch.nn import functional
from torch.nn.parameter import Parameter

from . import ops
from .local_convolution import Conv2dLocal, ColorDistanceHandle, _require_square


class MeanFieldLocal(nn.Module):
    def __init__(self, in_height: int, in_width: int, channels: int, kernel_size: Union[int, Tuple[int, int]],
                 iters: int = 1, compat: bool = False, affine: bool = False):
        super().__init__()
        if iters < 1:
            raise ValueError("iters must be >= 1")
        self.iters = iters
        # parameter names follow the reference model's state dict (AdvancedModel.py:136-138)
        self.local_convolution_layer = Conv2dLocal(in_height, in_width, channels, kernel_size)
        if affine:
            self.affine_weight = Parameter(torch.ones(1, channels, in_height, in_width))
            self.affine_bias = Parameter(torch.zeros(1, channels, in_height, in_width))
        else:
            self.affine_weight = self.affine_bias = None
        self.compat = Parameter(1.0 - torch.eye(channels)) if compat else None

    def forward(self, unary: torch.Tensor, image: torch.Tensor) -> torch.Tensor:
        local = self.local_convolution_layer
        _require_square(local.kernel_size)
        k = local.kernel_size[0]
        needs_grad = torch.is_grad_enabled() and (
            unary.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not needs_grad:
            return ops.meanfield_forward(unary, image, local.position_weight, local.color_weight, k, self.iters,
                                         self.affine_weight, self.affine_bias, self.compat)
        handle = ColorDistanceHandle(image, local.kernel_size)
        ls_unary = functional.log_softmax(unary, dim=1)
        q = unary
        logq = None
        for _ in range(self.iters):
            msg = local(q, handle)
            if self.affine_weight is not None:
                msg = self.affine_weight * msg + self.affine_bias
            if self.compat is not None:
                msg = torch.einsum("ij,bjhw->bihw", self.compat, msg)
            logq = functional.log_softmax(ls_unary - msg, dim=1)
            q = torch.exp(logq)
        return logq
