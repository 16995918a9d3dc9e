"""Geometry sweep of the diagonal gather kernels on the GPU: for seeded random shapes that the fast
path accepts, the image-mode local forward and backward (diagonal kernels, possibly with tail
workgroups and overflow records) must agree with the tensor-mode ones (generic kernels fed a
materialised colour-distance tensor), and one mean-field step must agree with its composition from the separate
entry points.  The generic kernels are pinned against the oracle in test_gpu_parity.py; this sweep
covers many more alignments (L % k, tails, row-start patterns) than the oracle could in seconds."""
import numpy as np
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import semantic_segmentation_amd as S
    return S


def sweep_shapes():
    rng = np.random.default_rng(20240611)
    shapes = []
    while len(shapes) < 28:
        k = int(rng.choice([3, 5, 7, 9, 11]))
        C = int(rng.integers(1, 34 if k <= 7 else 21))
        W = int(rng.integers(64 + k, 330))
        H = int(rng.integers(8, 260))
        if H * W < 4 * C * k * k or H * W * C * k * k > 1.2e9:
            continue
        shapes.append((int(rng.integers(1, 3)), C, H, W, k))
    return shapes


@pytest.mark.parametrize("shape", sweep_shapes())
def test_diagonal_kernels_agree_with_generic_kernels(shape, S):
    B, C, H, W, k = shape
    rng = np.random.default_rng(B + 10 * C + 1000 * H + W)
    x = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).cuda()
    img = torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype(np.float32)).cuda()
    pw = torch.tensor([0.3], device="cuda")
    cw = torch.tensor([-0.7], device="cuda")
    y_fast = S.ops.local_forward(x, pw, cw, k, image=img)
    cd = S.ops.color_distance(img, k)
    y_gen = S.ops.local_forward(x, pw, cw, k, cd=cd)
    assert_close(y_fast.cpu().numpy(), y_gen.cpu().numpy(), 1e-4, f"local forward {shape}")
    # backward: scatter on the diagonal kernels against the generic kernels' atomics
    dy = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).cuda()
    dx_fast, dpar_fast, _ = S.ops.local_backward(x, pw, cw, k, dy, image=img)
    dx_gen, dpar_gen, _ = S.ops.local_backward(x, pw, cw, k, dy, cd=cd)
    assert_close(dx_fast.cpu().numpy(), dx_gen.cpu().numpy(), 1e-4, f"local backward dx {shape}")
    a, b = dpar_fast.cpu().numpy().astype(np.float64), dpar_gen.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(a - b) <= 2e-4 * np.abs(b) + 1e-5 * np.abs(b).max()), (a, b)
    # one fused mean-field step = local forward of the unary + combine
    compat = torch.from_numpy(((1 - np.eye(C)) + 0.01 * rng.standard_normal((C, C))).astype(np.float32)).cuda()
    lq = S.ops.meanfield_forward(x, img, pw * 0.01, cw * 0.01, k, 1, None, None, compat)
    msg = S.ops.local_forward(x, pw * 0.01, cw * 0.01, k, cd=cd)
    smooth = torch.einsum("ij,bjhw->bihw", compat, msg)
    ref = S.ops.combine_forward(x, smooth.contiguous())
    assert_close(lq.cpu().numpy(), ref.cpu().numpy(), 1e-4, f"mean-field step {shape}")
