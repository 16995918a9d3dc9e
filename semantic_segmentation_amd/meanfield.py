"""K-iteration mean-field loop built on the local-window operator.

Per iteration (every step is one of the reference's modules, only the loop is new):
    M    = Conv2dLocal(Q_t, colour distance)                 LocalConvolution.py:59-90
    M    = affine_w * M + affine_b          (optional)        AdvancedModel.py:52-56, 223
    P    = compat . M                       (optional, LxL)   AdvancedModel.py:143-144 with a 1x1 kernel
    logQ = LogSoftmax(log_softmax(U) - P)                     AdvancedModel.py:242-244
    Q_{t+1} = exp(logQ),  Q_0 = U                             AdvancedModel.py:222

The forward is always the fused C-ABI loop.  When a gradient is required, the backward is the library's
loop backward (recompute with checkpoints, mfl_meanfield_backward) for up to 160 labels; above that the same
arithmetic is composed per iteration from the HIP local operator (with its HIP backward) and stock
differentiable torch ops for the normalisation.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import nn
from torch.autograd.function import once_differentiable
from torch.nn import functional
from torch.nn.parameter import Parameter

from . import ops
from .local_convolution import Conv2dLocal, ColorDistanceHandle, _require_square


class LinearActivation(nn.Module):
    """Per-pixel, per-label affine w * x + b with the parameter names and shapes of the reference's
    LinearActivation (AdvancedModel.py:38-56), so that its state-dict keys load here; initialised to the
    identity (the reference leaves the tensors uninitialised, AdvancedModel.py:41-43)."""

    def __init__(self, num_channels: int, height: int, width: int):
        super().__init__()
        self.weight = Parameter(torch.ones(1, num_channels, height, width))
        self.bias = Parameter(torch.zeros(1, num_channels, height, width))

    def forward(self, input_tensor: torch.Tensor) -> torch.Tensor:
        return self.weight * input_tensor + self.bias


class _MeanFieldFunction(torch.autograd.Function):
    """logq = K fused mean-field updates; backward by the library's loop backward."""

    @staticmethod
    def forward(ctx, unary, image, pos_w, col_w, affine_w, affine_b, compat, k, iters):
        ctx.k, ctx.iters = k, iters
        # every iterate is kept for the backward (iters tensors of the output's size) instead of recomputing the loop
        iterates = ops.meanfield_forward_iterates(unary, image, pos_w, col_w, k, iters, affine_w, affine_b, compat)
        ctx.save_for_backward(unary, image, pos_w, col_w, affine_w, affine_b, compat, iterates)
        return iterates[iters - 1].clone()        # (the saved iterates stay private to the backward)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_logq):
        unary, image, pos_w, col_w, affine_w, affine_b, compat, iterates = ctx.saved_tensors
        if ctx.needs_input_grad[1]:
            raise RuntimeError("MeanFieldLocal: no gradient for the image; detach it")
        d_unary, dparams, d_aw, d_ab, d_compat = ops.meanfield_backward(
            unary, image, pos_w, col_w, ctx.k, ctx.iters, grad_logq, affine_w, affine_b, compat, logq_iterates=iterates)
        need = ctx.needs_input_grad
        return (d_unary if need[0] else None, None, dparams[0:1] if need[2] else None, dparams[1:2] if need[3] else None,
                d_aw.reshape(affine_w.shape) if (d_aw is not None and need[4]) else None,
                d_ab.reshape(affine_b.shape) if (d_ab is not None and need[5]) else None,
                d_compat if (d_compat is not None and need[6]) else None, None, None)


class MeanFieldLocal(nn.Module):
    def __init__(self, in_height: int, in_width: int, channels: int, kernel_size: Union[int, Tuple[int, int]],
                 iters: int = 1, compat: bool = False, affine: bool = False):
        super().__init__()
        if iters < 1:
            raise ValueError("iters must be >= 1")
        self.iters = iters
        # parameter names follow the reference model's state dict (AdvancedModel.py:136-138)
        self.local_convolution_layer = Conv2dLocal(in_height, in_width, channels, kernel_size)
        # (keys local_activation_function.weight / .bias, AdvancedModel.py:138)
        self.local_activation_function = LinearActivation(channels, in_height, in_width) if affine else None
        self.compat = Parameter(1.0 - torch.eye(channels)) if compat else None

    def forward(self, unary: torch.Tensor, image: torch.Tensor) -> torch.Tensor:
        local = self.local_convolution_layer
        _require_square(local.kernel_size)
        k = local.kernel_size[0]
        needs_grad = torch.is_grad_enabled() and (
            unary.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not needs_grad:
            act = self.local_activation_function
            return ops.meanfield_forward(unary, image, local.position_weight, local.color_weight, k, self.iters,
                                         None if act is None else act.weight, None if act is None else act.bias,
                                         self.compat)
        if unary.shape[1] <= ops.MEANFIELD_BACKWARD_MAX_LABELS:
            act = self.local_activation_function
            return _MeanFieldFunction.apply(unary, image, local.position_weight, local.color_weight,
                                            None if act is None else act.weight, None if act is None else act.bias,
                                            self.compat, k, self.iters)
        handle = ColorDistanceHandle(image, local.kernel_size)
        ls_unary = functional.log_softmax(unary, dim=1)
        q = unary
        logq = None
        for _ in range(self.iters):
            msg = local(q, handle)
            if self.local_activation_function is not None:
                msg = self.local_activation_function(msg)
            if self.compat is not None:
                msg = torch.einsum("ij,bjhw->bihw", self.compat, msg)
            logq = functional.log_softmax(ls_unary - msg, dim=1)
            q = torch.exp(logq)
        return logq


class _LabelContextFunction(torch.autograd.Function):
    """(logq, smooth) = the fused label-context kernel; backward by mfl_label_context_backward_kept: besides the inputs,
    logq, the arg-min mixture (one byte per label and pixel) and the convolution's value there are kept, as autograd
    keeps the reference's intermediates alive (AdvancedModel.py:298-299), so the convolution is not run twice.
    keep_forward = False trades that memory (9 bytes per label and pixel) for a recomputation inside the backward."""
    keep_forward = True

    @staticmethod
    def forward(ctx, t, unary, la_w, la_b, conv_w, conv_b, ga_w, ga_b):
        if _LabelContextFunction.keep_forward:
            logq, smooth, (amin, cmin) = ops.label_context_forward(t, conv_w, unary, conv_b=conv_b, la_w=la_w, la_b=la_b,
                                                                   ga_w=ga_w, ga_b=ga_b, want_smooth=True, keep=True)
            ctx.save_for_backward(t, unary, la_w, la_b, conv_w, conv_b, ga_w, ga_b, logq, amin, cmin)
            return logq, smooth
        ctx.save_for_backward(t, unary, la_w, la_b, conv_w, conv_b, ga_w, ga_b)
        logq, smooth = ops.label_context_forward(t, conv_w, unary, conv_b=conv_b, la_w=la_w, la_b=la_b, ga_w=ga_w, ga_b=ga_b,
                                                 want_smooth=True)
        return logq, smooth

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_logq, grad_smooth):
        saved = ctx.saved_tensors
        t, unary, la_w, la_b, conv_w, conv_b, ga_w, ga_b = saved[:8]
        kept = tuple(saved[8:]) if len(saved) > 8 else None
        # an output nobody used arrives as zeros (materialised): passing it along costs one more read per element
        g = ops.label_context_backward(t, conv_w, unary, grad_logq=grad_logq, grad_smooth=grad_smooth, conv_b=conv_b,
                                       la_w=la_w, la_b=la_b, ga_w=ga_w, ga_b=ga_b, kept=kept)
        order = ("d_t", "d_unary", "d_la_w", "d_la_b", "d_conv_w", "d_conv_b", "d_ga_w", "d_ga_b")
        return tuple(g[key] if ctx.needs_input_grad[i] else None for i, key in enumerate(order))


class LabelContext(nn.Module):
    """The reference's label-context block and combine step (DPNModel.forward phases 3-4, AdvancedModel.py:223,
    232-244) on the library's fused kernel.  Submodule names and parameter shapes are the reference's
    (AdvancedModel.py:138, 143-145), so the matching entries of a DPNModel state dict load here.
    Under autograd the gradients of both outputs (the log-marginals and, for phase 3, the smoothness term,
    AdvancedModel.py:237-238) come from the library's backward."""

    def __init__(self, in_height: int, in_width: int, num_labels: int = 21, num_mixture_filters: int = 5,
                 kernel_size: int = 9):
        super().__init__()
        self.num_labels, self.num_mixture_filters = num_labels, num_mixture_filters
        channels = num_labels * num_mixture_filters
        self.local_activation_function = LinearActivation(num_labels, in_height, in_width)
        self.global_convolution_layer = nn.Conv2d(num_labels, channels, kernel_size=(kernel_size, kernel_size),
                                                  padding=(kernel_size // 2, kernel_size // 2))
        self.global_activation_function = LinearActivation(channels, in_height, in_width)

    def forward(self, local_output: torch.Tensor, unary: torch.Tensor, want_smooth: bool = False):
        conv = self.global_convolution_layer
        if torch.is_grad_enabled() and (local_output.requires_grad or unary.requires_grad
                                        or any(p.requires_grad for p in self.parameters())):
            logq, smooth = _LabelContextFunction.apply(local_output, unary, self.local_activation_function.weight,
                                                       self.local_activation_function.bias, conv.weight, conv.bias,
                                                       self.global_activation_function.weight,
                                                       self.global_activation_function.bias)
            return (logq, smooth) if want_smooth else logq
        return ops.label_context_forward(local_output, conv.weight, unary, conv_b=conv.bias,
                                         la_w=self.local_activation_function.weight,
                                         la_b=self.local_activation_function.bias,
                                         ga_w=self.global_activation_function.weight,
                                         ga_b=self.global_activation_function.bias,
                                         mixtures=self.num_mixture_filters, want_smooth=want_smooth)
