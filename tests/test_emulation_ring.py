"""Records and ownership map of the stream gather kernels (mfl_ring.h), replayed on the CPU against the oracle:
every output pixel produced exactly once by the 65 - k owning lanes of a wave (plus the tail pixels of a nearly
empty last b-block), the stream order of the taps, the side split per alignment class, the thresholds at
image-row starts and the rule that the element a lane needs there is the one lane + gap loaded."""
import ctypes

import numpy as np
import pytest

from conftest import assert_close

F = ctypes.POINTER(ctypes.c_float)
I = ctypes.POINTER(ctypes.c_int)

CASES = [  # B, C, H, W, k
    (1, 19, 80, 128, 11),      # 19 labels, 11x11: two channel splits of 10
    (1, 21, 100, 120, 11),     # 21 labels: three splits of 7
    (1, 8, 40, 100, 11),       # two splits (the minimum), one dummy channel... 8 = 7 + 1
    (2, 21, 80, 96, 9),        # 21 labels, 9x9: three splits of 8
    (1, 30, 110, 90, 9),       # three splits of 10
    (1, 31, 110, 100, 9),      # four splits of 8, one group per workgroup
    (1, 5, 60, 75, 11),        # narrowest supported width for k = 11
    (1, 12, 100, 120, 9),
    (1, 28, 90, 160, 11),      # three splits of 10
]


@pytest.mark.parametrize("case", CASES)
def test_ring_matches_oracle(case, emu_ring, oracle):
    B, C, H, W, k = case
    assert emu_ring.emu_ring_supported(C, H, W, k) == 1
    rng = np.random.default_rng(C * 1000 + H * 10 + k)
    x = rng.standard_normal((B, C, H, W), dtype=np.float32)
    img = rng.standard_normal((B, 3, H, W), dtype=np.float32)
    y = np.empty_like(x)
    cnt = np.empty((B, H * W), np.int32)
    rc = emu_ring.emu_ring_local_forward(x.ctypes.data_as(F), img.ctypes.data_as(F), ctypes.c_float(0.3),
                                         ctypes.c_float(-0.7), y.ctypes.data_as(F), cnt.ctypes.data_as(I), B, C, H, W, k)
    assert rc == 0
    assert (cnt == 1).all(), "every output pixel must be owned by exactly one (lane, pixel) slot"
    assert_close(y, oracle.local_forward(x, 0.3, -0.7, k, img=img), rtol=2e-5, what=str(case))


def test_ring_geometry_and_eligibility(emu_ring):
    out = (ctypes.c_int * 8)()
    # BASELINE config 4: 19 labels, 1024x2048, 11x11
    assert emu_ring.emu_ring_geom(16, 19, 1024, 2048, 11, out) == 0
    nbb, ng, ngw, ncs, ch, bl, tail, recw = list(out)
    assert (ng, ngw, ncs, ch, bl) == (19 * 11, 2, 2, 10, 54)        # two waves of 10 channels instead of three of 7
    assert nbb * bl + tail >= (1024 * 2048 + 19 * 121 - 1) // (19 * 121) + 10      # all b values plus the k - 1 shifted ones
    assert recw % 4 == 0 and recw >= 2 * ch + (ch + 1) // 2
    # 21 labels: the wide slices would not save a wave per group, so 3 x 7 stays
    assert emu_ring.emu_ring_geom(8, 21, 512, 512, 11, out) == 0
    assert tuple(out)[2:5] == (2, 3, 7)
    # windows the stream kernels are not built for, narrow images, too many labels
    assert emu_ring.emu_ring_supported(21, 512, 512, 7) == 0
    assert emu_ring.emu_ring_supported(19, 1024, 70, 11) == 0
    assert emu_ring.emu_ring_supported(40, 512, 512, 9) == 0
    assert emu_ring.emu_ring_supported(19, 1024, 2048, 10) == 0
