"""The part of the reference's DPNModel behind its unary CNN (AdvancedModel.py:134-152, 198-244), on the library's
kernels, with the reference's submodule names so that the matching entries of a DPNModel state dict load unchanged:

    color_distance_calculator      a ColorDistanceHandle factory               AdvancedModel.py:135
    local_convolution_layer        Conv2dLocal                                 AdvancedModel.py:136-137
    local_activation_function      LinearActivation(L, H, W)                   AdvancedModel.py:138
    global_convolution_layer       nn.Conv2d(L, L*M, 9x9, padding 4)           AdvancedModel.py:143-144
    global_activation_function     LinearActivation(L*M, H, W)                 AdvancedModel.py:145

`forward(input_tensor, phase)` follows DPNModel.forward phase by phase (the incremental training of
AdvancedModel.py:364-381 trains the model through phases 1 to 4).  The unary CNN itself (the VGG-like stack,
AdvancedModel.py:88-130) is outside this package: pass any module producing [B, L, H, W] scores as
`unary_terms_layers`, or call `forward_from_unary` with scores computed elsewhere.
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
from torch import nn
from torch.nn import functional

from .local_convolution import Conv2dLocal, color_distance
from .meanfield import LinearActivation, _LabelContextFunction
from . import ops


class DPNHead(nn.Module):
    NUM_PHASES = 4

    def __init__(self, image_height: int, image_width: int, num_labels: int = 21, num_mixture_filters: int = 5,
                 kernel_size: Union[int, Tuple[int, int]] = (50, 50), context_kernel_size: int = 9,
                 unary_terms_layers: Optional[nn.Module] = None):
        super().__init__()
        self.unary_terms_layers = unary_terms_layers
        self.local_convolution_layer = Conv2dLocal(in_height=image_height, in_width=image_width, channels=num_labels,
                                                   kernel_size=kernel_size)
        self.color_distance_calculator = color_distance(channels=3, kernel_size=self.local_convolution_layer.kernel_size)
        self.local_activation_function = LinearActivation(num_labels, image_height, image_width)
        channels = num_labels * num_mixture_filters
        self.global_convolution_layer = nn.Conv2d(num_labels, channels, kernel_size=(context_kernel_size, context_kernel_size),
                                                  padding=(context_kernel_size // 2, context_kernel_size // 2))
        self.global_activation_function = LinearActivation(channels, image_height, image_width)
        self.num_labels, self.num_mixture_filters = num_labels, num_mixture_filters
        self.image_size = (image_height, image_width)

    def forward(self, input_tensor: torch.Tensor, phase: int = NUM_PHASES) -> torch.Tensor:
        """DPNModel.forward (AdvancedModel.py:198-244): the image in, log-probabilities of the given phase out."""
        if self.unary_terms_layers is None:
            raise RuntimeError("DPNHead was built without a unary network: use forward_from_unary(unary, image, phase)")
        return self.forward_from_unary(self.unary_terms_layers(input_tensor), input_tensor, phase)

    def forward_from_unary(self, unary_output: torch.Tensor, input_tensor: torch.Tensor, phase: int = NUM_PHASES) -> torch.Tensor:
        if not 1 <= phase <= self.NUM_PHASES:
            raise ValueError(f"phase must be between 1 and {self.NUM_PHASES}")
        if phase == 1:                                                               # AdvancedModel.py:210-211
            return functional.log_softmax(unary_output, dim=1)
        intermediate = self.local_convolution_layer(unary_output, self.color_distance_calculator(input_tensor))
        if phase == 2:                                                               # AdvancedModel.py:222-226
            return functional.log_softmax(self.local_activation_function(intermediate), dim=1)
        # phases 3 and 4: the label-context block and the combine step in one kernel (AdvancedModel.py:223, 232-244)
        act, conv, gact = self.local_activation_function, self.global_convolution_layer, self.global_activation_function
        needs_grad = torch.is_grad_enabled() and (intermediate.requires_grad or unary_output.requires_grad or any(
            p.requires_grad for m in (act, conv, gact) for p in m.parameters()))
        if needs_grad:
            logq, smooth = _LabelContextFunction.apply(intermediate, unary_output, act.weight, act.bias, conv.weight, conv.bias,
                                                       gact.weight, gact.bias)
        else:
            logq, smooth = ops.label_context_forward(intermediate, conv.weight, unary_output, conv_b=conv.bias, la_w=act.weight,
                                                     la_b=act.bias, ga_w=gact.weight, ga_b=gact.bias, want_smooth=True)
        if phase == 3:                                                               # AdvancedModel.py:237-238
            return functional.log_softmax(smooth, dim=1)
        return logq

    def set_phase_trainable(self, phase_index: int) -> None:
        """The freezing schedule of the reference's incremental training (AdvancedModel.py:364-377), applied before
        training phase `phase_index + 1`: from the second phase on the unary network is frozen, in the third the local
        layers as well, and the last phase trains everything again."""
        if phase_index == 1 and self.unary_terms_layers is not None:
            self.unary_terms_layers.requires_grad_(False)
        if phase_index == 2:
            self.local_convolution_layer.requires_grad_(False)
            self.local_activation_function.requires_grad_(False)
        if phase_index == 3:
            if self.unary_terms_layers is not None:
                self.unary_terms_layers.requires_grad_(True)
            self.local_convolution_layer.requires_grad_(True)
            self.local_activation_function.requires_grad_(True)
