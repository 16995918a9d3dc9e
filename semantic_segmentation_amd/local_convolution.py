"""Drop-in replacements for the reference's local-window operator.

    Conv2dLocal        <- LocalConvolution.py:30-90 (same constructor, attributes, parameter and
                          buffer names, so an existing state dict loads unchanged)
    color_distance     <- DPNModel._color_distance, AdvancedModel.py:519-551 (same factory signature)

The reference materialises the colour-distance tensor [B, H*W, k*k] and hands it to
Conv2dLocal.forward.  Here the factory returns a ColorDistanceHandle that only carries the
image; Conv2dLocal recognises it and runs the fused kernel, which recomputes the distances
on chip.  A real tensor (from any source) is still accepted and read from memory.
"""
from __future__ import annotations

from itertools import repeat
from typing import Callable, Tuple, Union

import torch
from torch.autograd.function import once_differentiable
from torch import nn
from torch.nn.parameter import Parameter

from . import ops


def _pair(x, name: str = "") -> Tuple[int, int]:
    # LocalConvolution.py:11-22: a tuple must have length 2, a scalar is repeated
    if isinstance(x, tuple):
        if len(x) != 2:
            raise ValueError(f"Argument {name} must be of length 2")
        return x
    return tuple(repeat(x, 2))


class ColorDistanceHandle:
    """Stands for the tensor wrapped_color_distance(image) would return, without computing it."""

    def __init__(self, image: torch.Tensor, kernel_size: Tuple[int, int]):
        self.image = image
        self.kernel_size = kernel_size

    @property
    def shape(self):
        B, _, H, W = self.image.shape
        return torch.Size((B, H * W, self.kernel_size[0] * self.kernel_size[1]))

    def materialize(self) -> torch.Tensor:
        _require_square(self.kernel_size)
        return ops.color_distance(self.image, self.kernel_size[0])


def color_distance(kernel_size: Tuple[int, int], channels: int = 3) -> Callable[[torch.Tensor], ColorDistanceHandle]:
    """Same call shape as DPNModel._color_distance(kernel_size, channels) (AdvancedModel.py:520)."""
    kernel_size = _pair(kernel_size, "kernel_size")
    if channels != 3:
        raise ValueError("the fused colour distance handles 3-channel images")

    def wrapped_color_distance(image_tensor: torch.Tensor) -> ColorDistanceHandle:
        return ColorDistanceHandle(image_tensor, kernel_size)

    return wrapped_color_distance


def _require_square(kernel_size):
    if kernel_size[0] != kernel_size[1]:
        # the reference's forward fails for non-square windows as well (its pad arguments are
        # swapped, LocalConvolution.py:74-76, so the unfolded size no longer matches the view)
        raise RuntimeError(f"Conv2dLocal supports square windows only, got {kernel_size}")


class _LocalConvFunction(torch.autograd.Function):
    """y = x * sum_t (pd[t]*pw + cd*cw) * S;  backward gives dx, d pw, d cw (and d cd for a tensor)."""

    @staticmethod
    def forward(ctx, x, appearance, pos_w, col_w, k, from_cd):
        ctx.k, ctx.from_cd = k, from_cd
        ctx.save_for_backward(x, appearance, pos_w, col_w)
        if from_cd:
            return ops.local_forward(x, pos_w, col_w, k, cd=appearance)
        return ops.local_forward(x, pos_w, col_w, k, image=appearance)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, appearance, pos_w, col_w = ctx.saved_tensors
        if not ctx.from_cd and ctx.needs_input_grad[1]:
            # the reference's autograd would reach the image through its colour distance
            # (AdvancedModel.py:520-551); the fused kernels do not compute that gradient
            raise RuntimeError("Conv2dLocal: no gradient for the image behind a ColorDistanceHandle; "
                               "detach the image or pass a materialised colour-distance tensor")
        want_dcd = ctx.from_cd and ctx.needs_input_grad[1]
        kw = {"cd": appearance} if ctx.from_cd else {"image": appearance}
        dx, dparams, dcd = ops.local_backward(x, pos_w, col_w, ctx.k, dy, want_dcd=want_dcd, **kw)
        return (dx if ctx.needs_input_grad[0] else None, dcd,
                dparams[0:1] if ctx.needs_input_grad[2] else None,
                dparams[1:2] if ctx.needs_input_grad[3] else None, None, None)


class Conv2dLocal(nn.Module):
    """MI355X implementation of the reference's Conv2dLocal (LocalConvolution.py:30-90)."""

    def __init__(self, in_height: int, in_width: int, channels: int, kernel_size: Union[int, Tuple[int, int]]):
        super().__init__()
        self.channels = channels
        self.kernel_size = _pair(kernel_size, "kernel_size")
        self.in_height = in_height
        self.in_width = in_width

        self.color_weight = Parameter(torch.zeros(1, dtype=torch.float))
        self.position_weight = Parameter(torch.zeros(1, dtype=torch.float))

        # squared tap offsets, same values and int64 dtype as LocalConvolution.py:45-51 (kept for
        # state-dict compatibility; the kernels derive them from the tap index)
        kh, kw = self.kernel_size
        height_distance = (torch.arange(kh).reshape(1, -1) - kh // 2) ** 2
        width_distance = (torch.arange(kw).reshape(1, -1) - kw // 2) ** 2
        self.register_buffer("position_distances", (height_distance + width_distance.T).reshape(-1))

        self.reset_parameters()

    def reset_parameters(self):
        self.position_weight.data.uniform_(-0.005, 0.005)
        self.color_weight.data.uniform_(-0.005, 0.005)

    def forward(self, input_tensor: torch.Tensor, color_distance_tensor=0):
        _require_square(self.kernel_size)
        k = self.kernel_size[0]
        if input_tensor.dim() != 4 or input_tensor.shape[1] != self.channels or \
                tuple(input_tensor.shape[2:]) != (self.in_height, self.in_width):
            raise RuntimeError(f"expected input [B, {self.channels}, {self.in_height}, {self.in_width}], "
                               f"got {tuple(input_tensor.shape)}")
        if isinstance(color_distance_tensor, ColorDistanceHandle):
            if color_distance_tensor.kernel_size != self.kernel_size:
                raise RuntimeError("colour distance was built for a different window size")
            return _LocalConvFunction.apply(input_tensor, color_distance_tensor.image, self.position_weight,
                                            self.color_weight, k, False)
        if isinstance(color_distance_tensor, torch.Tensor):
            return _LocalConvFunction.apply(input_tensor, color_distance_tensor, self.position_weight,
                                            self.color_weight, k, True)
        # the reference's default (0) only broadcasts when H*W == k*k; it is a usage error here
        raise RuntimeError("Conv2dLocal.forward needs the colour distance (a ColorDistanceHandle or a "
                           "[B, H*W, k*k] tensor)")
