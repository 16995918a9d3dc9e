"""Tensor-level wrappers of the C ABI (include/mfl.h): argument checking, workspace and
output allocation through torch's caching allocator, launch on torch's current stream."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import lib, ptr, stream_of


def _c(t):
    return None if t is None else t.contiguous()


def color_distance(image: torch.Tensor, k: int) -> torch.Tensor:
    """[B,3,H,W] -> [B, H*W, k*k] squared colour distances (AdvancedModel.py:529-549)."""
    _lib.require_device(image)
    image = _c(image)
    B, ch, H, W = image.shape
    if ch != 3:
        raise ValueError("the colour distance is defined for 3-channel images")
    cd = torch.empty(B, H * W, k * k, dtype=torch.float32, device=image.device)
    with torch.cuda.device(image.device):
        _lib.check(lib.mfl_color_distance(ptr(image), ptr(cd), B, H, W, k, stream_of(image)), "mfl_color_distance")
    return cd


def local_forward(x, pos_w, col_w, k, image=None, cd=None) -> torch.Tensor:
    """Conv2dLocal.forward (LocalConvolution.py:59-90); appearance term from `image` or a `cd` tensor."""
    _lib.require_device(x, pos_w, col_w, image, cd)
    x, image, cd = _c(x), _c(image), _c(cd)
    B, C, H, W = x.shape
    _check_appearance(B, H, W, k, image, cd)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        ws = _lib.workspace(_lib.OP_LOCAL_FORWARD, B, C, H, W, k, 1, x.device)
        _lib.check(lib.mfl_local_forward(ptr(x), ptr(image), ptr(cd), ptr(pos_w), ptr(col_w), ptr(y), B, C, H, W, k,
                                         ptr(ws), ws.numel(), stream_of(x)), "mfl_local_forward")
    return y


def local_backward(x, pos_w, col_w, k, dy, image=None, cd=None, want_dcd=False):
    """Returns (dx, dparams[2] = {d position_weight, d color_weight}, dcd or None)."""
    _lib.require_device(x, pos_w, col_w, dy, image, cd)
    x, dy, image, cd = _c(x), _c(dy), _c(image), _c(cd)
    B, C, H, W = x.shape
    _check_appearance(B, H, W, k, image, cd)
    if dy.shape != x.shape:
        raise RuntimeError(f"grad shape {tuple(dy.shape)} does not match input shape {tuple(x.shape)}")
    dx = torch.empty_like(x)
    dparams = torch.empty(2, dtype=torch.float32, device=x.device)
    dcd = torch.empty_like(cd) if (want_dcd and cd is not None) else None
    with torch.cuda.device(x.device):
        ws = _lib.workspace(_lib.OP_LOCAL_BACKWARD, B, C, H, W, k, 1, x.device)
        _lib.check(lib.mfl_local_backward(ptr(x), ptr(image), ptr(cd), ptr(pos_w), ptr(col_w), ptr(dy), ptr(dx),
                                          ptr(dparams), ptr(dcd), B, C, H, W, k, ptr(ws), ws.numel(), stream_of(x)),
                   "mfl_local_backward")
    return dx, dparams, dcd


def combine_forward(unary, smooth, mixtures=1, want_q=False):
    """LogSoftmax(log_softmax(unary) - min over mixtures of smooth) (AdvancedModel.py:32-35, 242-244)."""
    _lib.require_device(unary, smooth)
    unary, smooth = _c(unary), _c(smooth)
    B, C, H, W = unary.shape
    if smooth.shape != (B, C * mixtures, H, W):
        raise RuntimeError(f"smooth must be {(B, C * mixtures, H, W)}, got {tuple(smooth.shape)}")
    logq = torch.empty_like(unary)
    q = torch.empty_like(unary) if want_q else None
    with torch.cuda.device(unary.device):
        _lib.check(lib.mfl_combine_forward(ptr(unary), ptr(smooth), mixtures, ptr(logq), ptr(q), B, C, H, W,
                                           stream_of(unary)), "mfl_combine_forward")
    return (logq, q) if want_q else logq


def meanfield_forward(unary, image, pos_w, col_w, k, iters, affine_w=None, affine_b=None, compat=None):
    """`iters` fused mean-field updates; returns the final log-marginals [B,C,H,W]."""
    _lib.require_device(unary, image, pos_w, col_w, affine_w, affine_b, compat)
    unary, image, affine_w, affine_b, compat = _c(unary), _c(image), _c(affine_w), _c(affine_b), _c(compat)
    B, C, H, W = unary.shape
    _check_appearance(B, H, W, k, image, None)
    if (affine_w is None) != (affine_b is None):
        raise ValueError("affine_w and affine_b must be given together")
    if affine_w is not None and (affine_w.numel() != C * H * W or affine_b.numel() != C * H * W):
        raise RuntimeError("affine parameters must hold C*H*W elements")
    if compat is not None and compat.shape != (C, C):
        raise RuntimeError(f"compat must be {(C, C)}, got {tuple(compat.shape)}")
    logq = torch.empty_like(unary)
    with torch.cuda.device(unary.device):
        ws = _lib.workspace(_lib.OP_MEANFIELD_FORWARD, B, C, H, W, k, iters, unary.device)
        _lib.check(lib.mfl_meanfield_forward(ptr(unary), ptr(image), ptr(pos_w), ptr(col_w), ptr(affine_w),
                                             ptr(affine_b), ptr(compat), ptr(logq), B, C, H, W, k, iters, ptr(ws),
                                             ws.numel(), stream_of(unary)), "mfl_meanfield_forward")
    return logq


MEANFIELD_BACKWARD_MAX_LABELS = 160


def meanfield_forward_iterates(unary, image, pos_w, col_w, k, iters, affine_w=None, affine_b=None, compat=None):
    """meanfield_forward that returns every iterate: [iters, B, C, H, W] log-marginals (the last one is the
    result of meanfield_forward); meanfield_backward takes them instead of recomputing the loop."""
    _lib.require_device(unary, image, pos_w, col_w, affine_w, affine_b, compat)
    unary, image, affine_w, affine_b, compat = _c(unary), _c(image), _c(affine_w), _c(affine_b), _c(compat)
    B, C, H, W = unary.shape
    _check_appearance(B, H, W, k, image, None)
    if (affine_w is None) != (affine_b is None):
        raise ValueError("affine_w and affine_b must be given together")
    if affine_w is not None and (affine_w.numel() != C * H * W or affine_b.numel() != C * H * W):
        raise RuntimeError("affine parameters must hold C*H*W elements")
    if compat is not None and compat.shape != (C, C):
        raise RuntimeError(f"compat must be {(C, C)}, got {tuple(compat.shape)}")
    out = torch.empty((iters,) + tuple(unary.shape), dtype=torch.float32, device=unary.device)
    with torch.cuda.device(unary.device):
        ws = _lib.workspace(_lib.OP_MEANFIELD_FORWARD, B, C, H, W, k, iters, unary.device)
        _lib.check(lib.mfl_meanfield_forward_iterates(ptr(unary), ptr(image), ptr(pos_w), ptr(col_w), ptr(affine_w),
                                                      ptr(affine_b), ptr(compat), ptr(out), B, C, H, W, k, iters, ptr(ws),
                                                      ws.numel(), stream_of(unary)), "mfl_meanfield_forward_iterates")
    return out


def meanfield_backward(unary, image, pos_w, col_w, k, iters, grad_logq, affine_w=None, affine_b=None, compat=None,
                       logq_iterates=None):
    """Gradients of meanfield_forward for grad_logq = dL/dlogq (autograd through AdvancedModel.py:222-244
    iterated; loss.backward(), AdvancedModel.py:299).  Returns (d_unary, dparams[2] = {d position_weight,
    d color_weight}, d_affine_w, d_affine_b, d_compat); the optional ones are None when the input is."""
    _lib.require_device(unary, image, pos_w, col_w, grad_logq, affine_w, affine_b, compat, logq_iterates)
    unary, image, grad_logq, logq_iterates = _c(unary), _c(image), _c(grad_logq), _c(logq_iterates)
    affine_w, affine_b, compat = _c(affine_w), _c(affine_b), _c(compat)
    B, C, H, W = unary.shape
    _check_appearance(B, H, W, k, image, None)
    if C > MEANFIELD_BACKWARD_MAX_LABELS:
        raise ValueError(f"meanfield_backward handles at most {MEANFIELD_BACKWARD_MAX_LABELS} labels, got {C}")
    if grad_logq.shape != unary.shape:
        raise RuntimeError(f"grad shape {tuple(grad_logq.shape)} does not match {tuple(unary.shape)}")
    if logq_iterates is not None and logq_iterates.shape != (iters,) + tuple(unary.shape):
        raise RuntimeError(f"logq_iterates must be {(iters,) + tuple(unary.shape)}, got {tuple(logq_iterates.shape)}")
    if (affine_w is None) != (affine_b is None):
        raise ValueError("affine_w and affine_b must be given together")
    if affine_w is not None and (affine_w.numel() != C * H * W or affine_b.numel() != C * H * W):
        raise RuntimeError("affine parameters must hold C*H*W elements")
    if compat is not None and compat.shape != (C, C):
        raise RuntimeError(f"compat must be {(C, C)}, got {tuple(compat.shape)}")
    d_unary = torch.empty_like(unary)
    dparams = torch.empty(2, dtype=torch.float32, device=unary.device)
    d_aw = torch.empty_like(affine_w) if affine_w is not None else None
    d_ab = torch.empty_like(affine_b) if affine_b is not None else None
    d_compat = torch.empty_like(compat) if compat is not None else None
    with torch.cuda.device(unary.device):
        ws = _lib.workspace(_lib.OP_MEANFIELD_BACKWARD, B, C, H, W, k, iters, unary.device)
        _lib.check(lib.mfl_meanfield_backward(ptr(unary), ptr(image), ptr(pos_w), ptr(col_w), ptr(affine_w), ptr(affine_b),
                                              ptr(compat), ptr(logq_iterates), ptr(grad_logq), ptr(d_unary), ptr(dparams),
                                              ptr(d_aw), ptr(d_ab),
                                              ptr(d_compat), B, C, H, W, k, iters, ptr(ws), ws.numel(), stream_of(unary)),
                   "mfl_meanfield_backward")
    return d_unary, dparams, d_aw, d_ab, d_compat


def argmax_labels(logq: torch.Tensor) -> torch.Tensor:
    """uint8 label map [B,H,W] = argmax over channels (AdvancedModel.py:420-421)."""
    _lib.require_device(logq)
    logq = _c(logq)
    B, C, H, W = logq.shape
    labels = torch.empty(B, H, W, dtype=torch.uint8, device=logq.device)
    with torch.cuda.device(logq.device):
        _lib.check(lib.mfl_argmax_labels(ptr(logq), ptr(labels), B, C, H, W, stream_of(logq)), "mfl_argmax_labels")
    return labels


def label_context_forward(t, conv_w, unary, conv_b=None, la_w=None, la_b=None, ga_w=None, ga_b=None, mixtures=None,
                          want_smooth=False, keep=False):
    """DPNModel.forward phases 3-4 behind the local convolution (AdvancedModel.py:223, 232-244): per-pixel affine,
    Conv2d(L -> L*M, ks x ks, pad ks//2), per-pixel affine, min over the M mixtures, then
    LogSoftmax(log_softmax(unary) - smooth).  Returns logq, or (logq, smooth) with want_smooth; with keep also the pair
    (arg-min mixture as uint8, the convolution's value there) that label_context_backward(kept=...) takes back instead of
    running the forward again."""
    _lib.require_device(t, conv_w, unary, conv_b, la_w, la_b, ga_w, ga_b)
    t, conv_w, unary, conv_b = _c(t), _c(conv_w), _c(unary), _c(conv_b)
    la_w, la_b, ga_w, ga_b = _c(la_w), _c(la_b), _c(ga_w), _c(ga_b)
    B, L, H, W = t.shape
    if conv_w.dim() != 4 or conv_w.shape[1] != L or conv_w.shape[2] != conv_w.shape[3] or conv_w.shape[0] % L:
        raise RuntimeError(f"conv_w must be [L*M, {L}, ks, ks], got {tuple(conv_w.shape)}")
    M, ks = conv_w.shape[0] // L, conv_w.shape[2]
    if mixtures is not None and mixtures != M:
        raise RuntimeError(f"conv_w holds {M} mixtures, expected {mixtures}")
    if unary.shape != t.shape:
        raise RuntimeError(f"unary must be {tuple(t.shape)}, got {tuple(unary.shape)}")
    if (la_w is None) != (la_b is None) or (ga_w is None) != (ga_b is None):
        raise ValueError("affine weight and bias must be given together")
    if la_w is not None and (la_w.numel() != L * H * W or la_b.numel() != L * H * W):
        raise RuntimeError("la_w / la_b must hold L*H*W elements")
    if ga_w is not None and (ga_w.numel() != L * M * H * W or ga_b.numel() != L * M * H * W):
        raise RuntimeError("ga_w / ga_b must hold L*M*H*W elements")
    if conv_b is not None and conv_b.numel() != L * M:
        raise RuntimeError("conv_b must hold L*M elements")
    n = int(lib.mfl_label_context_workspace_bytes(B, L, M, H, W, ks))
    if n == 0:
        raise ValueError(f"unsupported label-context problem: L={L} (<= 32), M={M} (<= 5), ks={ks} (odd, <= 15)")
    logq = torch.empty_like(t)
    smooth = torch.empty_like(t) if want_smooth else None
    with torch.cuda.device(t.device):
        ws = torch.empty(n, dtype=torch.uint8, device=t.device)
        if keep:
            amin = torch.empty(t.shape, dtype=torch.uint8, device=t.device)
            cmin = torch.empty_like(t)
            _lib.check(lib.mfl_label_context_forward_keep(ptr(t), ptr(la_w), ptr(la_b), ptr(conv_w), ptr(conv_b), ptr(ga_w),
                                                          ptr(ga_b), ptr(unary), ptr(smooth), ptr(logq), ptr(amin), ptr(cmin),
                                                          B, L, M, H, W, ks, ptr(ws), n, stream_of(t)),
                       "mfl_label_context_forward_keep")
            return (logq, smooth, (amin, cmin)) if want_smooth else (logq, (amin, cmin))
        _lib.check(lib.mfl_label_context_forward(ptr(t), ptr(la_w), ptr(la_b), ptr(conv_w), ptr(conv_b), ptr(ga_w), ptr(ga_b),
                                                 ptr(unary), ptr(smooth), ptr(logq), B, L, M, H, W, ks, ptr(ws), n,
                                                 stream_of(t)), "mfl_label_context_forward")
    return (logq, smooth) if want_smooth else logq


def label_context_backward(t, conv_w, unary, grad_logq=None, grad_smooth=None, conv_b=None, la_w=None, la_b=None,
                           ga_w=None, ga_b=None, kept=None):
    """Gradients of label_context_forward (autograd through AdvancedModel.py:223, 232-244; loss.backward(),
    AdvancedModel.py:299) for dL/dlogq and / or dL/dsmooth.  Returns a dict with d_t, d_unary, d_conv_w and, for the
    optional inputs that were given, d_conv_b, d_la_w, d_la_b, d_ga_w, d_ga_b (shaped like the inputs).
    kept: (logq, amin, cmin) of label_context_forward(..., keep=True) on the same inputs: the forward is not run again."""
    _lib.require_device(t, conv_w, unary, grad_logq, grad_smooth, conv_b, la_w, la_b, ga_w, ga_b)
    t, conv_w, unary, conv_b = _c(t), _c(conv_w), _c(unary), _c(conv_b)
    la_w, la_b, ga_w, ga_b, grad_logq, grad_smooth = _c(la_w), _c(la_b), _c(ga_w), _c(ga_b), _c(grad_logq), _c(grad_smooth)
    B, L, H, W = t.shape
    if conv_w.dim() != 4 or conv_w.shape[1] != L or conv_w.shape[2] != conv_w.shape[3] or conv_w.shape[0] % L:
        raise RuntimeError(f"conv_w must be [L*M, {L}, ks, ks], got {tuple(conv_w.shape)}")
    M, ks = conv_w.shape[0] // L, conv_w.shape[2]
    if grad_logq is None and grad_smooth is None:
        raise ValueError("a gradient of logq or of smooth is needed")
    for g in (grad_logq, grad_smooth, unary):
        if g is not None and g.shape != t.shape:
            raise RuntimeError(f"expected {tuple(t.shape)}, got {tuple(g.shape)}")
    if (la_w is None) != (la_b is None) or (ga_w is None) != (ga_b is None):
        raise ValueError("affine weight and bias must be given together")
    if la_w is not None and (la_w.numel() != L * H * W or la_b.numel() != L * H * W):
        raise RuntimeError("la_w / la_b must hold L*H*W elements")
    if ga_w is not None and (ga_w.numel() != L * M * H * W or ga_b.numel() != L * M * H * W):
        raise RuntimeError("ga_w / ga_b must hold L*M*H*W elements")
    n = int(lib.mfl_label_context_backward_workspace_bytes(B, L, M, H, W, ks))
    if n == 0:
        raise ValueError(f"unsupported label-context problem: L={L} (<= 32), M={M} (<= 5), ks={ks} (odd, <= 15), or too large")
    like = lambda v: None if v is None else torch.empty_like(v)
    out = {"d_t": torch.empty_like(t), "d_unary": torch.empty_like(t), "d_conv_w": torch.empty_like(conv_w),
           "d_conv_b": like(conv_b), "d_la_w": like(la_w), "d_la_b": like(la_b), "d_ga_w": like(ga_w), "d_ga_b": like(ga_b)}
    if kept is not None:
        k_logq, k_amin, k_cmin = kept
        _lib.require_device(k_logq, k_cmin)
        if not k_amin.is_cuda or k_amin.device != t.device:
            raise RuntimeError("kept tensors must live on the device of t")
        if k_logq.shape != t.shape or k_cmin.shape != t.shape or k_amin.shape != t.shape or k_amin.dtype != torch.uint8:
            raise RuntimeError("kept must be (logq float32, amin uint8, cmin float32), each shaped like t")
        k_logq, k_amin, k_cmin = _c(k_logq), k_amin.contiguous(), _c(k_cmin)
        with torch.cuda.device(t.device):
            ws = torch.empty(n, dtype=torch.uint8, device=t.device)
            _lib.check(lib.mfl_label_context_backward_kept(
                ptr(t), ptr(la_w), ptr(la_b), ptr(conv_w), ptr(conv_b), ptr(ga_w), ptr(ga_b), ptr(unary), ptr(k_logq), ptr(k_amin),
                ptr(k_cmin), ptr(grad_logq), ptr(grad_smooth), ptr(out["d_t"]), ptr(out["d_unary"]), ptr(out["d_la_w"]),
                ptr(out["d_la_b"]), ptr(out["d_conv_w"]), ptr(out["d_conv_b"]), ptr(out["d_ga_w"]), ptr(out["d_ga_b"]), B, L, M, H, W,
                ks, ptr(ws), n, stream_of(t)), "mfl_label_context_backward_kept")
        return out
    with torch.cuda.device(t.device):
        ws = torch.empty(n, dtype=torch.uint8, device=t.device)
        _lib.check(lib.mfl_label_context_backward(
            ptr(t), ptr(la_w), ptr(la_b), ptr(conv_w), ptr(conv_b), ptr(ga_w), ptr(ga_b), ptr(unary), ptr(grad_logq),
            ptr(grad_smooth), ptr(out["d_t"]), ptr(out["d_unary"]), ptr(out["d_la_w"]), ptr(out["d_la_b"]), ptr(out["d_conv_w"]),
            ptr(out["d_conv_b"]), ptr(out["d_ga_w"]), ptr(out["d_ga_b"]), B, L, M, H, W, ks, ptr(ws), n, stream_of(t)),
            "mfl_label_context_backward")
    return out


def iou_counts(pred: torch.Tensor, target: torch.Tensor, num_labels: int) -> torch.Tensor:
    """int64 [3, num_labels]: per class the pixels in both, in the prediction, in the target
    (the counts behind DPNModel.calc_intersection_over_union, AdvancedModel.py:448-485).
    pred / target: uint8 label maps of equal shape; labels >= num_labels count for no class."""
    if pred.dtype != torch.uint8 or target.dtype != torch.uint8:
        raise TypeError("label maps must be uint8")
    if not (pred.is_cuda and target.is_cuda) or pred.device != target.device:
        raise RuntimeError("semantic_segmentation_amd runs on a ROCm GPU only: label maps must be on one GPU "
                           "(there is no CPU fallback)")
    if pred.shape != target.shape:
        raise RuntimeError(f"prediction {tuple(pred.shape)} and target {tuple(target.shape)} differ in shape")
    pred, target = _c(pred), _c(target)
    counts = torch.empty(3, num_labels, dtype=torch.int64, device=pred.device)
    with torch.cuda.device(pred.device):
        _lib.check(lib.mfl_iou_counts(ptr(pred), ptr(target), ptr(counts), pred.numel(), num_labels, stream_of(pred)),
                   "mfl_iou_counts")
    return counts


def intersection_over_union(pred, target, num_labels: int, include_background: bool = False) -> torch.Tensor:
    """Per-class IoU with the reference's conventions (AdvancedModel.py:469-483): float32 [num_labels]; a class
    absent from both maps gets -1; without include_background class 0 is left at 0."""
    counts = iou_counts(pred, target, num_labels).to(torch.float64)
    inter, union = counts[0], counts[1] + counts[2] - counts[0]
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.full_like(inter, -1.0)).to(torch.float32)
    if not include_background:
        iou[0] = 0
    return iou


def mean_intersection_over_union(pred, target, num_labels: int, include_background: bool = False) -> float:
    """Mean over the classes with IoU > 0 (AdvancedModel.py:436-446); 0 if there is none."""
    iou = intersection_over_union(pred, target, num_labels, include_background)
    mask = iou > 0
    n = int(mask.sum())
    return float(iou[mask].sum() / n) if n else 0


def _check_appearance(B, H, W, k, image, cd):
    if (image is None) == (cd is None):
        raise ValueError("exactly one of image / cd must be given")
    if image is not None and image.shape != (B, 3, H, W):
        raise RuntimeError(f"image must be {(B, 3, H, W)}, got {tuple(image.shape)}")
    if cd is not None and cd.shape != (B, H * W, k * k):
        raise RuntimeError(f"color distance tensor must be {(B, H * W, k * k)}, got {tuple(cd.shape)}")
