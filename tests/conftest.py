import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run with -m gpu on the MI355X box)")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "_*.npz")))


def load(path):
    return dict(np.load(path, allow_pickle=False))


def rel_err(a, b):
    """max |a-b| relative to the largest reference magnitude."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def assert_close(actual, expected, rtol=1e-4, what=""):
    """|actual - expected| <= rtol * (|expected| + max|expected|) elementwise: 1e-4 relative, with
    the tensor's own scale as the floor for elements that cancel to (almost) zero."""
    a = np.asarray(actual, dtype=np.float64)
    e = np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, f"{what}: shape {a.shape} vs {e.shape}"
    assert np.isfinite(a).all(), f"{what}: non-finite values"
    tol = rtol * (np.abs(e) + np.abs(e).max())
    bad = np.abs(a - e) > tol
    assert not bad.any(), (f"{what}: {int(bad.sum())} of {a.size} elements differ, "
                           f"max rel-to-scale {rel_err(a, e):.3e}")


@pytest.fixture(scope="session")
def oracle():
    """The C oracle (built on demand with gcc)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def emu():
    """Host emulation of the device gather core, built on demand with clang++."""
    import ctypes
    src = os.path.join(ROOT, "tests", "emu", "emu_core.cpp")
    so = os.path.join(ROOT, "tests", "emu", "libmfl_emu.so")
    core = os.path.join(ROOT, "semantic_segmentation_amd", "csrc", "mfl_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core)):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
        subprocess.check_call([clang, "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def emu_diag():
    """Host emulation of the diagonal gather (address table, ownership, flagged steps)."""
    import ctypes
    src = os.path.join(ROOT, "tests", "emu", "emu_diag.cpp")
    so = os.path.join(ROOT, "tests", "emu", "libmfl_emu_diag.so")
    hdrs = [os.path.join(ROOT, "semantic_segmentation_amd", "csrc", h) for h in ("mfl_core.h", "mfl_diag.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
        subprocess.check_call([clang, "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def emu_ring():
    """Host emulation of the stream gather kernels (records, thresholds, lane permute rule, ownership)."""
    import ctypes
    src = os.path.join(ROOT, "tests", "emu", "emu_ring.cpp")
    so = os.path.join(ROOT, "tests", "emu", "libmfl_emu_ring.so")
    hdrs = [os.path.join(ROOT, "semantic_segmentation_amd", "csrc", h) for h in ("mfl_core.h", "mfl_diag.h", "mfl_ring.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        clang = "/opt/rocm/lib/llvm/bin/clang++"
        subprocess.check_call([clang, "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, src])
    return ctypes.CDLL(so)
