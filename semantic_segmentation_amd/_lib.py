"""ctypes binding of libmfl.so (the HIP kernels behind the C ABI declared in include/mfl.h).

There is no CPU fallback: if the shared library is missing or an entry point
is absent the import of this module fails, and every call checks the library's
return code and raises.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MFL_LIBRARY selects another build of the same ABI (kernel experiments); default: the in-tree library
LIB_PATH = os.environ.get("MFL_LIBRARY") or os.path.join(_HERE, "libmfl.so")

OP_LOCAL_FORWARD = 1
OP_LOCAL_BACKWARD = 2
OP_MEANFIELD_FORWARD = 3
OP_MEANFIELD_BACKWARD = 4

_c_int, _c_size, _c_void = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/mfl.h one to one
SIGNATURES = {
    "mfl_version": (_c_int, []),
    "mfl_error_string": (ctypes.c_char_p, [_c_int]),
    "mfl_workspace_bytes": (_c_size, [_c_int] * 7),
    "mfl_color_distance": (_c_int, [_c_void, _c_void] + [_c_int] * 4 + [_c_void]),
    "mfl_local_forward": (_c_int, [_c_void] * 6 + [_c_int] * 5 + [_c_void, _c_size, _c_void]),
    "mfl_local_backward": (_c_int, [_c_void] * 9 + [_c_int] * 5 + [_c_void, _c_size, _c_void]),
    "mfl_combine_forward": (_c_int, [_c_void, _c_void, _c_int, _c_void, _c_void] + [_c_int] * 4 + [_c_void]),
    "mfl_meanfield_forward": (_c_int, [_c_void] * 8 + [_c_int] * 6 + [_c_void, _c_size, _c_void]),
    "mfl_meanfield_forward_iterates": (_c_int, [_c_void] * 8 + [_c_int] * 6 + [_c_void, _c_size, _c_void]),
    "mfl_meanfield_backward": (_c_int, [_c_void] * 14 + [_c_int] * 6 + [_c_void, _c_size, _c_void]),
    "mfl_meanfield_forward_noalloc": (_c_int, [_c_void] * 8 + [_c_int] * 6 + [_c_void]),
    "mfl_label_context_workspace_bytes": (_c_size, [_c_int] * 6),
    "mfl_label_context_forward": (_c_int, [_c_void] * 10 + [_c_int] * 6 + [_c_void, _c_size, _c_void]),
    "mfl_label_context_backward_workspace_bytes": (_c_size, [_c_int] * 6),
    "mfl_label_context_backward": (_c_int, [_c_void] * 18 + [_c_int] * 6 + [_c_void, _c_size, _c_void]),
    "mfl_label_context_forward_keep": (_c_int, [_c_void] * 12 + [_c_int] * 6 + [_c_void, _c_size, _c_void]),
    "mfl_label_context_backward_kept": (_c_int, [_c_void] * 21 + [_c_int] * 6 + [_c_void, _c_size, _c_void]),
    "mfl_preprocess_workspace_bytes": (_c_size, [_c_int] * 4),
    "mfl_preprocess_rgb8": (_c_int, [_c_void, _c_int, _c_int, _c_void, _c_void, _c_int, _c_void, _c_void, _c_int,
                                     ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _c_void, _c_int, _c_int,
                                     _c_void, _c_size, _c_void]),
    "mfl_resize_labels": (_c_int, [_c_void, _c_int, _c_int, _c_void, _c_void, _c_void, _c_int, _c_int, _c_void]),
    "mfl_iou_counts": (_c_int, [_c_void, _c_void, _c_void, _c_size, _c_int, _c_void]),
    "mfl_argmax_labels": (_c_int, [_c_void, _c_void] + [_c_int] * 4 + [_c_void]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C semantic_segmentation_amd/csrc)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


class MflError(RuntimeError):
    pass


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib.mfl_error_string(code).decode()
        if code == -1:
            raise ValueError(f"{what}: {msg}")
        raise MflError(f"{what}: {msg} (code {code})")


def version() -> int:
    return int(lib.mfl_version())


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_of(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_device(*tensors: torch.Tensor) -> None:
    """The kernels only run on a ROCm device; host tensors are rejected rather than served by a CPU path."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("semantic_segmentation_amd runs on a ROCm GPU only: got a tensor on "
                               f"'{t.device}' (there is no CPU fallback)")
        if t.dtype != torch.float32:
            raise RuntimeError(f"expected float32 tensors, got {t.dtype}")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} and {t.device}")


_PLANS = {}          # (op, B, C, H, W, k, iters, device) -> workspace kept between calls (keep_plans(True))
_KEEP_PLANS = False


def keep_plans(enable: bool = True) -> None:
    """Keep one workspace per (operation, problem size, device) alive between calls instead of taking a fresh
    one from torch's allocator each time.  The library tags a workspace with the plan (address tables) it
    holds and skips rebuilding them when it finds its own tag, which is only safe while nobody else writes
    to that memory: a kept workspace guarantees it, a block handed back to the caching allocator does not
    (so fresh workspaces get their tag cleared).  keep_plans(False) drops the kept workspaces."""
    global _KEEP_PLANS
    _KEEP_PLANS = bool(enable)
    if not enable:
        _PLANS.clear()


def workspace(op: int, B: int, C: int, H: int, W: int, k: int, iters: int, device) -> torch.Tensor:
    n = int(lib.mfl_workspace_bytes(op, B, C, H, W, k, iters))
    if n == 0:
        raise ValueError(f"invalid problem size B={B} C={C} H={H} W={W} k={k}")
    key = (op, B, C, H, W, k, iters, str(device))
    if _KEEP_PLANS and key in _PLANS:
        return _PLANS[key]
    ws = torch.empty(n, dtype=torch.uint8, device=device)
    ws[:32].zero_()          # no plan tag: whatever this block held before is not trusted
    if _KEEP_PLANS:
        _PLANS[key] = ws
    return ws