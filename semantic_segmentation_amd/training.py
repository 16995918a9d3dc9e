"""Batched training loop and checkpoint format for the modules of this package.

What DPNModel.fit does around the hot path (AdvancedModel.py:246-393), for callers that hold the unary
scores: the reference feeds one image per forward and steps the optimiser every `batch_size` images with each
loss divided by `batch_size` (AdvancedModel.py:295-303); here the `batch_size` images are ONE batched tensor
and one fused call, the loss is the same mean of per-image NLL losses, gradients are summed over the ranks of
a process group before the step, and checkpoints carry the reference's keys (AdvancedModel.py:317-321, 388-391).
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional, Tuple

import torch
from torch import nn

from . import parallel


def class_weights(labels: torch.Tensor, num_labels: int) -> torch.Tensor:
    """Inverse pixel counts per class over the given label maps, zero for absent classes
    (DPNModel._calc_weights, AdvancedModel.py:500-517); labels >= num_labels (255 = void) are not counted.
    Computed where `labels` lives (no host round trip)."""
    flat = labels.reshape(-1).to(torch.long)
    counts = torch.bincount(flat[flat < num_labels], minlength=num_labels).to(torch.float32)
    return torch.where(counts > 0, 1.0 / counts.clamp(min=1), torch.zeros_like(counts))


def per_image_nll(log_probs: torch.Tensor, labels: torch.Tensor, weight: Optional[torch.Tensor] = None,
                  ignore_index: int = 255) -> torch.Tensor:
    """Mean over the batch of nn.NLLLoss(weight, ignore_index)(log_probs[b:b+1], labels[b:b+1]): what the
    reference accumulates with its `loss / batch_size` over `batch_size` single-image forwards
    (AdvancedModel.py:295-299).  An image without a valid pixel contributes 0 (the reference would produce NaN)."""
    losses = nn.functional.nll_loss(log_probs, labels, weight=weight, ignore_index=ignore_index, reduction="none")
    valid = labels != ignore_index
    if weight is None:
        denom = valid.flatten(1).sum(1).to(losses.dtype)
    else:
        denom = (weight[labels.clamp(max=weight.numel() - 1)] * valid).flatten(1).sum(1)
    per_image = losses.flatten(1).sum(1) / denom.clamp(min=torch.finfo(losses.dtype).tiny)
    return torch.where(denom > 0, per_image, torch.zeros_like(per_image)).mean()


def fit(model: nn.Module, optimizer: torch.optim.Optimizer,
        batches: Callable[[], Iterable[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]], epochs: int,
        weight: Optional[torch.Tensor] = None, global_batch: Optional[int] = None, group=None,
        on_epoch: Optional[Callable[[int, float], None]] = None) -> list:
    """Trains `model(unary, image) -> log-marginals` (MeanFieldLocal).  `batches()` yields (unary [B,L,H,W],
    image [B,3,H,W], labels [B,H,W] long) device tensors, this rank's shard of each global batch.
    One optimiser step per batch; with a process group the loss is scaled by shard/global_batch and the
    gradients are summed with one flat all-reduce before the step (parallel.allreduce_gradients).
    Returns the list of epoch losses (sum of the batch losses, as the reference reports)."""
    history = []
    for epoch in range(epochs):
        epoch_loss = 0.0
        for unary, image, labels in batches():
            optimizer.zero_grad(set_to_none=True)
            share = 1.0 if global_batch is None else unary.shape[0] / float(global_batch)
            loss = per_image_nll(model(unary, image), labels, weight) * share
            loss.backward()
            parallel.allreduce_gradients(model.parameters(), group)
            optimizer.step()
            epoch_loss += float(loss.detach())
        history.append(epoch_loss)
        if on_epoch is not None:
            on_epoch(epoch, epoch_loss)
    return history


def fit_incremental(model: nn.Module, optimizer: torch.optim.Optimizer,
                    batches: Callable[[], Iterable[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]], epochs: int,
                    weight: Optional[torch.Tensor] = None, incremental: bool = True, global_batch: Optional[int] = None,
                    group=None, on_epoch: Optional[Callable[[int, int, float], None]] = None) -> list:
    """The reference's incremental training (AdvancedModel.py:364-381) for a model with `forward_from_unary(unary,
    image, phase)` and `set_phase_trainable(phase_index)` (DPNHead): `epochs` epochs per phase, phases 1 .. 4, the
    layers frozen and released as the reference does, ONE optimiser for all phases; without `incremental` a single
    run of the last phase.  `batches()` yields (unary or None, image, labels): with None the model's own unary
    network produces the scores from the image.  Returns one list of epoch losses per phase."""
    phases = range(model.NUM_PHASES) if incremental else [model.NUM_PHASES - 1]
    histories = []
    for phase_index in phases:
        if incremental:
            model.set_phase_trainable(phase_index)
        phase = phase_index + 1
        history = []
        for epoch in range(epochs):
            epoch_loss = 0.0
            for unary, image, labels in batches():
                optimizer.zero_grad(set_to_none=True)
                share = 1.0 if global_batch is None else image.shape[0] / float(global_batch)
                out = model(image, phase) if unary is None else model.forward_from_unary(unary, image, phase)
                loss = per_image_nll(out, labels, weight) * share
                if loss.requires_grad:                 # (a phase whose layers are all frozen has nothing to train)
                    loss.backward()
                    parallel.allreduce_gradients(model.parameters(), group)
                    optimizer.step()
                epoch_loss += float(loss.detach())
            history.append(epoch_loss)
            if on_epoch is not None:
                on_epoch(phase, epoch, epoch_loss)
        histories.append(history)
    return histories


def save_checkpoint(path: str, model: nn.Module, optimizer: torch.optim.Optimizer, epoch: Optional[int] = None,
                    phase: Optional[int] = None, loss: Optional[float] = None) -> None:
    """The reference's checkpoint dictionary: "model_state_dict" and "optimizer_state_dict", plus "epoch", "phase",
    "loss" for a mid-training checkpoint (AdvancedModel.py:317-321, 388-391)."""
    payload = {"model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict()}
    for key, value in (("epoch", epoch), ("phase", phase), ("loss", loss)):
        if value is not None:
            payload[key] = value
    torch.save(payload, path)


def load_checkpoint(path: str, model: nn.Module, optimizer: Optional[torch.optim.Optimizer] = None,
                    map_location=None) -> dict:
    """Loads a checkpoint in the reference's format the way DPNModel.fit does (AdvancedModel.py:350-360): model keys
    non-strictly (a DPNModel checkpoint holds many layers this package does not model), an optimiser state that
    does not fit is skipped.  Tensors only are unpickled.  Returns {"missing", "unexpected", "optimizer_loaded"} plus
    the checkpoint's epoch / phase / loss when present."""
    weights = torch.load(path, map_location=map_location, weights_only=True)
    result = model.load_state_dict(weights["model_state_dict"], strict=False)
    info = {"missing": list(result.missing_keys), "unexpected": list(result.unexpected_keys), "optimizer_loaded": False}
    if optimizer is not None and "optimizer_state_dict" in weights:
        try:
            optimizer.load_state_dict(weights["optimizer_state_dict"])
            info["optimizer_loaded"] = True
        except ValueError:
            pass
    for key in ("epoch", "phase", "loss"):
        if key in weights:
            info[key] = weights[key]
    return info
