"""mfl_meanfield_forward_noalloc (the entry point with the scratch buffer built into the library) against
mfl_meanfield_forward (caller's workspace), through the C ABI: same kernels, so the results are compared
bit for bit; two calls in a row, the second on a smaller problem, then the size limit."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SCRATCH_LIMIT = -4


def _inputs(B, C, H, W, seed):
    rng = np.random.default_rng(seed)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).cuda()
    return dict(unary=t(2 * rng.standard_normal((B, C, H, W))), img=t(rng.standard_normal((B, 3, H, W))),
                aw=t(1 + 0.1 * rng.standard_normal((C, H, W))), ab=t(0.1 * rng.standard_normal((C, H, W))),
                compat=t(0.5 * rng.standard_normal((C, C))), pw=torch.tensor([-0.004], device="cuda"),
                cw=torch.tensor([-0.003], device="cuda"))


def _both(shape, k, iters, seed):
    import semantic_segmentation_amd as S
    from semantic_segmentation_amd import _lib
    lib, ptr = _lib.lib, _lib.ptr
    B, C, H, W = shape
    d = _inputs(B, C, H, W, seed)
    want = S.ops.meanfield_forward(d["unary"], d["img"], d["pw"], d["cw"], k, iters, d["aw"], d["ab"], d["compat"])
    got = torch.full_like(d["unary"], float("nan"))
    rc = lib.mfl_meanfield_forward_noalloc(ptr(d["unary"]), ptr(d["img"]), ptr(d["pw"]), ptr(d["cw"]), ptr(d["aw"]),
                                           ptr(d["ab"]), ptr(d["compat"]), ptr(got), B, C, H, W, k, iters,
                                           _lib.stream_of(got))
    assert rc == 0, lib.mfl_error_string(rc).decode()
    torch.cuda.synchronize()
    return got.cpu().numpy(), want.cpu().numpy()


def test_noalloc_matches_workspace_entry_point_twice():
    assert torch.cuda.is_available()
    # first call: 11x11 stream kernels (weight table, two padded buffers, records); second, smaller: the
    # diagonal kernels on what the first call left in the buffer; last: the first shape again, whose plan tag the
    # calls in between replaced (the tables are rebuilt, not trusted)
    for shape, k, iters, seed in (((1, 8, 64, 128), 11, 2, 5), ((2, 21, 96, 112), 7, 3, 6), ((1, 5, 40, 72), 3, 2, 7),
                                  ((1, 8, 64, 128), 11, 2, 5)):
        got, want = _both(shape, k, iters, seed)
        assert np.isfinite(got).all()
        assert np.array_equal(got, want), f"{shape} k={k}: max diff {np.abs(got - want).max()}"


def test_noalloc_rejects_problems_past_the_built_in_buffer():
    from semantic_segmentation_amd import _lib
    lib = _lib.lib
    B, C, H, W, k = 8, 21, 512, 512, 7
    assert lib.mfl_workspace_bytes(_lib.OP_MEANFIELD_FORWARD, B, C, H, W, k, 1) > (64 << 20)
    x = torch.zeros(1, device="cuda")
    p = _lib.ptr(x)
    rc = lib.mfl_meanfield_forward_noalloc(p, p, p, p, None, None, None, p, B, C, H, W, k, 1, None)
    assert rc == SCRATCH_LIMIT
    assert b"scratch" in lib.mfl_error_string(rc)
