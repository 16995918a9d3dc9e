"""Address table and ownership map of the diagonal gather (mfl_diag.h), replayed on the CPU against
the oracle: every output pixel produced exactly once, the A/B split per tap column, flagged steps at
image-row starts, channel splits with dummy channels, partial b-blocks and group blocks."""
import ctypes

import numpy as np
import pytest

from conftest import assert_close

F = ctypes.POINTER(ctypes.c_float)
I = ctypes.POINTER(ctypes.c_int)

CASES = [  # B, C, H, W, k
    (1, 3, 8, 70, 3),          # narrowest supported width for k = 3
    (2, 5, 20, 80, 5),
    (1, 21, 64, 96, 7),        # 21 labels, 7x7: two channel splits of 11 (one dummy channel)
    (1, 19, 80, 128, 11),      # 19 labels, 11x11: splits of 10
    (1, 8, 33, 75, 3),
    (1, 9, 40, 71, 3),         # two splits of 8
    (1, 23, 70, 90, 5),        # three splits
    (1, 2, 128, 128, 3),       # BASELINE config 1 shape
    (1, 4, 30, 200, 9),
    (1, 5, 35, 81, 3),         # a last b-block of 1 live lane: its pixels go to the tail kernel
    (1, 8, 130, 200, 5),       # ... of 6 live lanes
]


@pytest.mark.parametrize("case", CASES)
def test_diag_matches_oracle(case, emu_diag, oracle):
    B, C, H, W, k = case
    rng = np.random.default_rng(C * 1000 + H * 10 + k)
    x = rng.standard_normal((B, C, H, W), dtype=np.float32)
    img = rng.standard_normal((B, 3, H, W), dtype=np.float32)
    y = np.empty_like(x)
    cnt = np.empty((B, H * W), np.int32)
    rc = emu_diag.emu_diag_local_forward(x.ctypes.data_as(F), img.ctypes.data_as(F), ctypes.c_float(0.3),
                                         ctypes.c_float(-0.7), y.ctypes.data_as(F), cnt.ctypes.data_as(I), B, C, H, W, k)
    assert rc == 0
    assert (cnt == 1).all(), "every output pixel must be owned by exactly one (lane, pixel) slot"
    assert_close(y, oracle.local_forward(x, 0.3, -0.7, k, img=img), rtol=2e-5, what=str(case))


def test_unsupported_shapes_are_rejected(emu_diag):
    x = np.zeros((1, 3, 8, 60), np.float32)
    img = np.zeros((1, 3, 8, 60), np.float32)
    cnt = np.empty((1, 480), np.int32)
    for k in (3, 4):   # too narrow / even window: the generic kernels handle these
        rc = emu_diag.emu_diag_local_forward(x.ctypes.data_as(F), img.ctypes.data_as(F), ctypes.c_float(0.3),
                                             ctypes.c_float(-0.7), x.ctypes.data_as(F), cnt.ctypes.data_as(I), 1, 3, 8, 60, k)
        assert rc == -1


@pytest.mark.parametrize("shape", [(2, 128, 128, 3), (21, 96, 112, 7), (23, 70, 90, 5), (9, 50, 75, 3), (12, 100, 120, 9),
                                   (19, 128, 160, 7), (6, 200, 100, 7), (8, 130, 96, 11)])
def test_backward_scatter_targets_have_one_writer_or_two_edge_lanes(shape, emu_diag):
    """The diagonal backward stores a lane's sum into the copy of the gradient planes named by (side, window row of the
    record): replayed from the table records for every (b-block, group, tap, channel, lane), a target has one writer, or
    two that are both edge lanes by the kernel's rule (those add atomically, two addends onto zero), on shapes with at
    least k * C * k * k pixels (the entry point's condition for this path)."""
    C, H, W, k = shape
    assert H * W >= k * C * k * k
    out = (ctypes.c_longlong * 4)()
    assert emu_diag.emu_diag_scatter_targets(C, H, W, k, out) == 0
    multi, multi_non_edge, most, written = out[0], out[1], out[2], out[3]
    assert written > 0
    assert most <= 2, f"{most} writers on one target"
    assert multi_non_edge == 0, f"{multi_non_edge} of {multi} shared targets have a writer that stores instead of adding"
