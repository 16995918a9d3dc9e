"""ctypes front-end of the C oracle (oracle/mfl_oracle.c).

TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker / reported CPU baseline.  The product package
(semantic_segmentation_amd) must never import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmfl_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_D = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (a few seconds)."""
    src = os.path.join(_HERE, "mfl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.mflo_num_threads.restype = ctypes.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(_F)


def num_threads() -> int:
    return int(lib().mflo_num_threads())


def color_distance(img, k):
    img = _f32(img)
    B, _, H, W = img.shape
    cd = np.empty((B, H * W, k * k), np.float32)
    lib().mflo_color_distance(_p(img), _p(cd), B, H, W, k)
    return cd


def local_forward(x, pw, cw, k, cd=None, img=None):
    """Exactly one of cd ([B,HW,K]) / img ([B,3,H,W]) supplies the appearance term."""
    assert (cd is None) != (img is None)
    x = _f32(x)
    cd = None if cd is None else _f32(cd)
    img = None if img is None else _f32(img)
    B, C, H, W = x.shape
    y = np.empty_like(x)
    lib().mflo_local_forward(_p(x), _p(cd), _p(img), ctypes.c_float(pw), ctypes.c_float(cw), _p(y), B, C, H, W, k)
    return y


def local_backward(x, pw, cw, k, dy, cd=None, img=None):
    assert (cd is None) != (img is None)
    x, dy = _f32(x), _f32(dy)
    cd = None if cd is None else _f32(cd)
    img = None if img is None else _f32(img)
    B, C, H, W = x.shape
    dx = np.empty_like(x)
    dpar = np.zeros(2, np.float64)
    lib().mflo_local_backward(_p(x), _p(cd), _p(img), ctypes.c_float(pw), ctypes.c_float(cw), _p(dy), _p(dx),
                              dpar.ctypes.data_as(_D), B, C, H, W, k)
    return dx, float(dpar[0]), float(dpar[1])


def combine(unary, smooth, mixtures=1, want_q=False):
    unary, smooth = _f32(unary), _f32(smooth)
    B, C, H, W = unary.shape
    assert smooth.shape == (B, C * mixtures, H, W)
    logq = np.empty_like(unary)
    q = np.empty_like(unary) if want_q else None
    lib().mflo_combine(_p(unary), _p(smooth), _p(logq), _p(q), B, C, H, W, mixtures)
    return (logq, q) if want_q else logq


def meanfield_forward(unary, img, pw, cw, k, iters, affine_w=None, affine_b=None, compat=None):
    unary, img = _f32(unary), _f32(img)
    B, C, H, W = unary.shape
    aw = None if affine_w is None else _f32(affine_w).reshape(C, H, W)
    ab = None if affine_b is None else _f32(affine_b).reshape(C, H, W)
    cm = None if compat is None else _f32(compat).reshape(C, C)
    logq = np.empty_like(unary)
    lib().mflo_meanfield_forward(_p(unary), _p(img), ctypes.c_float(pw), ctypes.c_float(cw), _p(aw), _p(ab), _p(cm),
                                 _p(logq), B, C, H, W, k, iters)
    return logq


def _log_softmax(z):
    m = z.max(axis=1, keepdims=True)
    return z - m - np.log(np.exp(z - m).sum(axis=1, keepdims=True))


def meanfield_backward(unary, img, pw, cw, k, iters, grad_logq, affine_w=None, affine_b=None, compat=None):
    """Gradients of sum(logq_K * grad_logq) through the K-iteration loop, i.e. what autograd derives
    from the composition AdvancedModel.py:222-244 iterated (loss.backward(), AdvancedModel.py:299).

    Restated by hand, iteration t = K-1 .. 0, with z_t = log_softmax(U) - P_t:
        logQ_{t+1} = z_t - lse(z_t)        =>  dz = g - Q_{t+1} * sum_c g
        P_t = compat . A_t                  =>  dA = compat^T . dP (dP = -dz),  dcompat += sum dP (x) A_t
        A_t = aw * M_t + ab                 =>  dM = aw * dA,  daw += sum_b dA * M_t,  dab += sum_b dA
        M_t = Conv2dLocal(Q_t)              =>  (dQ_t, dpw, dcw) by the local backward (LocalConvolution.py:59-90)
        Q_t = exp(logQ_t)  (t > 0)          =>  g <- dQ_t * Q_t;      Q_0 = U: dU += dQ_0
    and dU += d log_softmax(U) for the summed dz.  Elementwise steps in float64, the two local operators by
    the C oracle (fp32 storage, double accumulation).  Returns a dict of float64 arrays / floats."""
    U = np.asarray(unary, np.float64)
    B, C, H, W = U.shape
    aw = None if affine_w is None else np.asarray(affine_w, np.float64).reshape(1, C, H, W)
    ab = None if affine_b is None else np.asarray(affine_b, np.float64).reshape(1, C, H, W)
    cm = None if compat is None else np.asarray(compat, np.float64).reshape(C, C)
    ls_u = _log_softmax(U)
    # forward, keeping every Q_t and M_t
    qs, ms = [U], []
    for _ in range(iters):
        m = local_forward(qs[-1], pw, cw, k, img=img).astype(np.float64)
        ms.append(m)
        a = m if aw is None else aw * m + ab
        p = a if cm is None else np.einsum("ij,bjhw->bihw", cm, a)
        qs.append(np.exp(_log_softmax(ls_u - p)))
    g = np.asarray(grad_logq, np.float64)
    d_ls = np.zeros_like(U)
    out = {"dpw": 0.0, "dcw": 0.0, "dcompat": np.zeros((C, C)), "daw": np.zeros((C, H, W)), "dab": np.zeros((C, H, W))}
    for t in range(iters - 1, -1, -1):
        dz = g - qs[t + 1] * g.sum(axis=1, keepdims=True)
        d_ls += dz
        dp = -dz
        a = ms[t] if aw is None else aw * ms[t] + ab
        if cm is None:
            da = dp
        else:
            out["dcompat"] += np.einsum("bihw,bjhw->ij", dp, a)
            da = np.einsum("ij,bihw->bjhw", cm, dp)
        if aw is None:
            dm = da
        else:
            out["daw"] += (da * ms[t]).sum(axis=0)
            out["dab"] += da.sum(axis=0)
            dm = aw * da
        dq, dpw, dcw = local_backward(qs[t], pw, cw, k, dm, img=img)
        out["dpw"] += dpw
        out["dcw"] += dcw
        dq = dq.astype(np.float64)
        if t > 0:
            g = dq * qs[t]
        else:
            d_u0 = dq
    out["dU"] = d_u0 + d_ls - np.exp(ls_u) * d_ls.sum(axis=1, keepdims=True)
    out["logq"] = np.log(qs[-1])
    return out
