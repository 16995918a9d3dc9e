"""MI355X-native local-window message passing (the hot path of yannay-alon/Semantic_Segmentation).

Importing this package loads libmfl.so; it fails if the HIP extension has not been built.
"""
from . import _lib, ops, parallel, preprocess, training  # noqa: F401
from .local_convolution import ColorDistanceHandle, Conv2dLocal, color_distance  # noqa: F401
from .meanfield import LabelContext, LinearActivation, MeanFieldLocal  # noqa: F401
from .dpn import DPNHead  # noqa: F401

__all__ = ["Conv2dLocal", "ColorDistanceHandle", "color_distance", "MeanFieldLocal", "LabelContext", "LinearActivation", "DPNHead", "ops", "parallel", "preprocess", "training"]
