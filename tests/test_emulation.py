"""Index arithmetic of the device gather core (mfl_core.h), emulated thread by thread on the CPU,
against the oracle: geometry, offset table, strided column mapping, incremental decode, row
crossing of M-wide runs, channel chunks, grid-tail duplicates."""
import ctypes

import numpy as np
import pytest

from conftest import assert_close

F = ctypes.POINTER(ctypes.c_float)


def _p(a):
    return None if a is None else a.ctypes.data_as(F)


CASES = [  # B, C, H, W, k, M, CC, use_cd
    (2, 2, 5, 7, 3, 1, 8, False),
    (2, 2, 5, 7, 3, 1, 8, True),
    (1, 5, 9, 11, 3, 2, 8, False),
    (1, 5, 9, 11, 3, 4, 8, False),
    (1, 19, 24, 20, 3, 4, 21, False),     # C < CC
    (1, 21, 32, 32, 3, 4, 21, False),
    (1, 21, 16, 16, 7, 1, 21, False),
    (2, 3, 40, 37, 3, 4, 3, False),       # odd width: many runs cross a row end
    (1, 13, 30, 31, 4, 2, 8, False),      # two channel chunks, even window
    (1, 5, 4, 4, 6, 1, 8, False),         # C*K > L
    (1, 7, 64, 64, 3, 4, 8, False),
    (1, 21, 12, 9, 11, 1, 21, False),     # window larger than the image
    (1, 3, 50, 70, 5, 4, 3, True),
]


@pytest.mark.parametrize("case", CASES)
def test_core_matches_oracle(case, emu, oracle):
    B, C, H, W, k, M, CC, use_cd = case
    rng = np.random.default_rng(hash(case) % 2**32)
    x = rng.standard_normal((B, C, H, W), dtype=np.float32)
    img = rng.standard_normal((B, 3, H, W), dtype=np.float32)
    cd = oracle.color_distance(img, k) if use_cd else None
    y = np.empty_like(x)
    rc = emu.emu_local_forward(_p(x), _p(None if use_cd else img), _p(cd), ctypes.c_float(0.3), ctypes.c_float(-0.7),
                               _p(y), B, C, H, W, k, M, CC)
    assert rc == 0
    assert_close(y, oracle.local_forward(x, 0.3, -0.7, k, img=img), rtol=2e-5, what=str(case))


def test_choose_m(emu):
    emu.emu_choose_m.argtypes = [ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int]
    assert emu.emu_choose_m(512 * 512, 21 * 49, 4) == 4        # 21 labels, 512^2, 7x7
    assert emu.emu_choose_m(1024 * 2048, 19 * 121, 4) == 4     # 19 labels, 1024x2048, 11x11
    assert emu.emu_choose_m(64 * 64, 21 * 2500, 4) == 1        # the model's own 50x50 window at 64^2
