"""Parity of the HIP path (through the C ABI) with the oracle and the reference's golden vectors.

Bar: fp32 results within 1e-4 relative (conftest.assert_close: |a-e| <= 1e-4 * (|e| + max|e|)),
and argmax label maps identical wherever the top-2 margin exceeds that tolerance.
"""
import numpy as np
import pytest
import torch

from conftest import assert_close, golden_files, load, rel_err

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope="module")
def S():
    import semantic_segmentation_amd as S
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return S


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def scalar(v):
    return torch.tensor([float(v)], dtype=torch.float32, device="cuda")


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ golden vectors
@pytest.mark.parametrize("path", golden_files("cd"))
def test_color_distance_golden(path, S):
    d = load(path)
    cd = host(S.ops.color_distance(dev(d["img"]), int(d["k"])))
    assert_close(cd, d["cd"], RTOL, "cd")


@pytest.mark.parametrize("path", golden_files("local"))
def test_local_golden_tensor_mode(path, S):
    """Conv2dLocal with the golden's own (random) colour-distance tensor: forward and all gradients."""
    d = load(path)
    k = int(d["k"])
    x, cd, dy = dev(d["x"]), dev(d["cd"]), dev(d["dy"])
    pw, cw = scalar(d["pw"]), scalar(d["cw"])
    y = S.ops.local_forward(x, pw, cw, k, cd=cd)
    assert_close(host(y), d["y"], RTOL, "y")
    dx, dpar, _ = S.ops.local_backward(x, pw, cw, k, dy, cd=cd)
    assert_close(host(dx), d["dx"], RTOL, "dx")
    dpar = host(dpar)
    for got, key in ((dpar[0], "dpw64"), (dpar[1], "dcw64")):
        ref = float(d[key][0])
        assert abs(got - ref) <= RTOL * abs(ref) + 1e-6, (key, got, ref)


@pytest.mark.parametrize("path", golden_files("meanfield"))
def test_meanfield_golden(path, S):
    d = load(path)
    k = int(d["k"])
    aw = dev(d["la_w"]) if "la_w" in d else None
    ab = dev(d["la_b"]) if "la_b" in d else None
    for it in (1, 2, 5, 10):
        lq = S.ops.meanfield_forward(dev(d["unary"]), dev(d["img"]), scalar(d["pw"]), scalar(d["cw"]), k, it,
                                     aw, ab, dev(d["compat"]))
        assert_close(host(lq), d[f"logq_it{it}"], RTOL, f"logq it{it}")


def test_dpn_phase2_and_combine_golden(S):
    """DPNModel.forward phase 2 / phase 4 data flow with the reference's own 50x50 window."""
    d = load(golden_files("dpn")[0])
    k, M = int(d["k"]), int(d["mixtures"])
    unary, img = dev(d["unary"]), dev(d["img"])
    msg = S.ops.local_forward(unary, scalar(d["pw"]), scalar(d["cw"]), k, image=img)
    a = dev(d["la_w"]) * msg + dev(d["la_b"])
    assert_close(host(torch.log_softmax(a, dim=1)), d["phase2"], RTOL, "phase2")
    c = dev(d["ga_w"]) * torch.nn.functional.conv2d(a, dev(d["conv_w"]), dev(d["conv_b"]), padding=4) + dev(d["ga_b"])
    out = S.ops.combine_forward(unary, c, mixtures=M)
    assert_close(host(out), d["phase4"], 2e-4, "phase4")   # includes MIOpen's 9x9 convolution


def test_label_context_tail_golden(S):
    """a5 -> a8 -> a6 -> a7 (AdvancedModel.py:223-244) in the library's fused kernel (9x9 convolution on the
    fp32 matrix instructions, no vendor convolution): the reference's tail outputs `smooth` and `out`."""
    d = load(golden_files("tail")[0])
    out, smooth = S.ops.label_context_forward(dev(d["t"]), dev(d["conv_w"]), dev(d["unary"]), conv_b=dev(d["conv_b"]),
                                              la_w=dev(d["la_w"]), la_b=dev(d["la_b"]), ga_w=dev(d["ga_w"]),
                                              ga_b=dev(d["ga_b"]), mixtures=int(d["mixtures"]), want_smooth=True)
    assert_close(host(smooth), d["smooth"], RTOL, "smooth")
    assert_close(host(out), d["out"], RTOL, "out")


def test_dpn_phases_2_3_4_on_the_library_alone(S):
    """DPNModel.forward phases 2, 3 and 4 (AdvancedModel.py:216-244) from the unary on, with the reference's own
    50x50 window: local operator, then the fused label-context kernel; no stock convolution involved."""
    d = load(golden_files("dpn")[0])
    k, M = int(d["k"]), int(d["mixtures"])
    unary, img = dev(d["unary"]), dev(d["img"])
    msg = S.ops.local_forward(unary, scalar(d["pw"]), scalar(d["cw"]), k, image=img)
    out, smooth = S.ops.label_context_forward(msg, dev(d["conv_w"]), unary, conv_b=dev(d["conv_b"]), la_w=dev(d["la_w"]),
                                              la_b=dev(d["la_b"]), ga_w=dev(d["ga_w"]), ga_b=dev(d["ga_b"]), mixtures=M,
                                              want_smooth=True)
    assert_close(host(torch.log_softmax(smooth, dim=1)), d["phase3"], RTOL, "phase3")
    assert_close(host(out), d["phase4"], RTOL, "phase4")


@pytest.mark.parametrize("B,L,M,H,W,ks", [(2, 21, 5, 40, 70, 9), (1, 7, 2, 33, 31, 3), (1, 32, 1, 20, 65, 5)])
def test_label_context_vs_float64_composition(B, L, M, H, W, ks, S):
    """The fused kernel against the same chain composed in float64 on the host (conv2d, affine, min, softmax),
    on sizes with ragged row tiles, with and without the optional affines and bias."""
    g = torch.Generator().manual_seed(77)
    t = torch.randn(B, L, H, W, generator=g)
    unary = 2 * torch.randn(B, L, H, W, generator=g)
    conv_w = 0.05 * torch.randn(L * M, L, ks, ks, generator=g)
    conv_b = 0.1 * torch.randn(L * M, generator=g)
    la_w, la_b = 1 + 0.2 * torch.randn(1, L, H, W, generator=g), 0.1 * torch.randn(1, L, H, W, generator=g)
    ga_w, ga_b = 1 + 0.2 * torch.randn(1, L * M, H, W, generator=g), 0.1 * torch.randn(1, L * M, H, W, generator=g)
    for full in (True, False):
        kw = dict(conv_b=conv_b, la_w=la_w, la_b=la_b, ga_w=ga_w, ga_b=ga_b) if full else {}
        a = (la_w * t + la_b) if full else t
        c = torch.nn.functional.conv2d(a.double(), conv_w.double(), conv_b.double() if full else None, padding=ks // 2)
        if full:
            c = ga_w.double() * c + ga_b.double()
        smooth = c.reshape(B, M, L, H, W).min(dim=1).values
        want = torch.log_softmax(torch.log_softmax(unary.double(), dim=1) - smooth, dim=1)
        out, sm = S.ops.label_context_forward(t.cuda(), conv_w.cuda(), unary.cuda(), want_smooth=True,
                                              **{n: v.cuda() for n, v in kw.items()})
        assert_close(host(sm), smooth.numpy(), RTOL, f"smooth full={full}")
        assert_close(host(out), want.numpy(), RTOL, f"logq full={full}")


def test_tail_golden(S):
    d = load(golden_files("tail")[0])
    out, q = S.ops.combine_forward(dev(d["unary"]), dev(d["c"]), mixtures=int(d["mixtures"]), want_q=True)
    assert_close(host(out), d["out"], RTOL, "combine")
    assert_close(host(q), np.exp(d["out"]), RTOL, "q")


# ------------------------------------------------------------------ seeded inputs vs the oracle
SHAPES = [  # B, C, H, W, k
    (1, 2, 128, 128, 3),       # BASELINE config 1 shape (2 labels, 128^2)
    (2, 21, 48, 64, 7),        # 21 labels, 7x7 (configs 2/3, reduced image)
    (1, 19, 40, 96, 11),       # 19 labels, 11x11 (config 4, reduced image)
    (1, 40, 24, 24, 7),        # channel chunks: 40 = 32 + 8
    (1, 150, 16, 16, 7),       # 150 labels (config 5, reduced image)
    (1, 5, 33, 17, 4),         # even window, odd sizes
    (2, 3, 1, 1, 3),           # a single pixel
    (1, 7, 9, 300, 5),         # wide and short
    (1, 21, 64, 64, 50),       # the model's own configuration (offset table in global memory)
    (1, 4, 31, 29, 1),         # 1x1 window
    # wide enough for the diagonal gather kernels (W >= 64 + k, H*W >= 4*C*k*k)
    (2, 21, 96, 112, 7),       # two channel splits of 11
    (1, 19, 96, 128, 11),      # wide window, stream kernels: two channel splits of 10
    (1, 23, 70, 90, 5),        # three channel splits
    (2, 5, 40, 80, 5),
    (1, 9, 50, 75, 3),
    (1, 32, 90, 100, 3),
    (1, 12, 100, 120, 9),      # 9x9, stream kernels: two channel splits of 8
    (1, 19, 80, 100, 7),       # two channel splits of 10
    (1, 6, 60, 100, 7),        # one split of 8, eight groups per workgroup
    (1, 14, 64, 96, 5),        # two splits of 8
    (1, 16, 64, 96, 7),
    (1, 28, 80, 96, 5),        # three splits of 10
    (1, 4, 60, 100, 11),       # wide windows: one live split of 7 (and an idle one)
    (1, 8, 64, 100, 9),        # ... of 8
    (1, 19, 90, 120, 9),       # 9x9: two splits of 10
    (1, 25, 96, 128, 11),      # 11x11: three splits of 10
    (1, 31, 110, 140, 11),     # 11x11: four splits of 10, one group per workgroup
    (1, 21, 100, 120, 11),     # 11x11: three splits of 7
    (1, 5, 35, 81, 3),         # a last b-block of 1 live lane: its pixels go to the tail kernel
    (2, 8, 130, 200, 5),       # ... of 6 live lanes
    (1, 17, 64, 70, 3),        # barely wide enough: most steps meet a row start (records with more flagged steps than slots)
    (1, 40, 96, 100, 3),       # more than 33 labels on the diagonal kernels: two blocks of 20 output channels
    (2, 70, 100, 90, 5),       # ... blocks of 18, 18, 18 and 16
]
DIAG_SHAPES = SHAPES[10:]


def make_inputs(shape, seed=0):
    B, C, H, W, k = shape
    rng = np.random.default_rng(1234 + seed)
    x = (2 * rng.standard_normal((B, C, H, W))).astype(np.float32)
    mean = np.array([0.485, 0.456, 0.406], np.float32).reshape(1, 3, 1, 1)
    std = np.array([0.229, 0.224, 0.225], np.float32).reshape(1, 3, 1, 1)
    img = ((rng.random((B, 3, H, W), dtype=np.float32) - mean) / std).astype(np.float32)
    dy = rng.standard_normal((B, C, H, W)).astype(np.float32)
    return x, img, dy


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("weights", [(-0.004, -0.003), (0.3, -0.7)])
def test_local_forward_backward_vs_oracle(shape, weights, S, oracle):
    B, C, H, W, k = shape
    pw, cw = weights
    x, img, dy = make_inputs(shape)
    y = S.ops.local_forward(dev(x), scalar(pw), scalar(cw), k, image=dev(img))
    assert_close(host(y), oracle.local_forward(x, pw, cw, k, img=img), RTOL, "y")
    dx, dpar, _ = S.ops.local_backward(dev(x), scalar(pw), scalar(cw), k, dev(dy), image=dev(img))
    rdx, rdpw, rdcw = oracle.local_backward(x, pw, cw, k, dy, img=img)
    assert_close(host(dx), rdx, RTOL, "dx")
    dpar = host(dpar)
    assert abs(dpar[0] - rdpw) <= 2e-4 * abs(rdpw) + 1e-5, (dpar[0], rdpw)
    assert abs(dpar[1] - rdcw) <= 2e-4 * abs(rdcw) + 1e-5, (dpar[1], rdcw)


@pytest.mark.parametrize("shape", SHAPES[:6])
def test_tensor_mode_equals_image_mode(shape, S):
    """A materialised colour-distance tensor gives the same result as the fused recomputation."""
    B, C, H, W, k = shape
    x, img, dy = make_inputs(shape, 1)
    pw, cw = scalar(0.3), scalar(-0.7)
    cd = S.ops.color_distance(dev(img), k)
    a = S.ops.local_forward(dev(x), pw, cw, k, image=dev(img))
    b = S.ops.local_forward(dev(x), pw, cw, k, cd=cd)
    assert_close(host(b), host(a), 1e-5, "tensor vs image mode")
    dxa, dpa, _ = S.ops.local_backward(dev(x), pw, cw, k, dev(dy), image=dev(img))
    dxb, dpb, dcd = S.ops.local_backward(dev(x), pw, cw, k, dev(dy), cd=cd, want_dcd=True)
    assert_close(host(dxb), host(dxa), 1e-5, "dx")
    assert_close(host(dpb), host(dpa), 1e-4, "dparams")
    # d cd = cw * sum_c D * S, so sum(cd * dcd) / cw = d cw
    got = float((cd.double() * dcd.double()).sum() / -0.7)
    assert abs(got - float(dpb[1])) <= 1e-4 * abs(float(dpb[1])) + 1e-5


def strict_rel_err_on_marginals(logq, ref_logq):
    """max |Q - Q_ref| / Q_ref over the marginals Q = exp(logq) with Q_ref > 1e-6: the elementwise
    relative figure (no tensor-scale floor).  It equals the absolute error of the logit, so it grows with
    the size of the messages: fp32 sums of K*C terms hold it below 1e-4 for potentials of the reference's
    initial scale (|weights| <= 0.005, LocalConvolution.py:55-57)."""
    q, qr = np.exp(np.asarray(logq, np.float64)), np.exp(np.asarray(ref_logq, np.float64))
    m = qr > 1e-6
    return float((np.abs(q - qr)[m] / qr[m]).max())


@pytest.mark.parametrize("shape", [(1, 21, 64, 64, 50), (2, 5, 40, 48, 20), (1, 32, 64, 64, 17)])
@pytest.mark.parametrize("mode", ["plain", "compat", "affine+compat"])
def test_meanfield_large_window_vs_oracle(shape, mode, S, oracle):
    """Large windows on small images (the reference model's own 64 x 64 / 50 x 50 configuration, AdvancedModel.py:135-137):
    the tap-split forward kernel and the unfused label step.  Weights small enough that 2 500 taps keep the logits
    in a range where the elementwise figure on the marginals means something."""
    B, C, H, W, k = shape
    unary, img, _ = make_inputs(shape, 5)
    rng = np.random.default_rng(11)
    compat = ((1 - np.eye(C)) + 0.01 * rng.standard_normal((C, C))).astype(np.float32) if "compat" in mode else None
    aw = (1 + 0.2 * rng.standard_normal((1, C, H, W))).astype(np.float32) if "affine" in mode else None
    ab = (0.1 * rng.standard_normal((1, C, H, W))).astype(np.float32) if "affine" in mode else None
    opt = lambda a: None if a is None else dev(a)
    pw, cw = -4e-5 * (49.0 / (k * k)), -3e-4 * (49.0 / (k * k))
    for iters in (1, 2):
        lq = host(S.ops.meanfield_forward(dev(unary), dev(img), scalar(pw), scalar(cw), k, iters, opt(aw), opt(ab), opt(compat)))
        ref = oracle.meanfield_forward(unary, img, pw, cw, k, iters, aw, ab, compat)
        assert_close(lq, ref, RTOL, f"logq {mode} iters={iters}")
        assert strict_rel_err_on_marginals(lq, ref) < 1e-4


@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[1], SHAPES[2], SHAPES[3], SHAPES[5]] + DIAG_SHAPES)
@pytest.mark.parametrize("mode", ["plain", "compat", "affine+compat"])
def test_meanfield_vs_oracle(shape, mode, S, oracle):
    B, C, H, W, k = shape
    unary, img, _ = make_inputs(shape, 2)
    rng = np.random.default_rng(7)
    compat = ((1 - np.eye(C)) + 0.01 * rng.standard_normal((C, C))).astype(np.float32) if "compat" in mode else None
    aw = (1 + 0.2 * rng.standard_normal((1, C, H, W))).astype(np.float32) if "affine" in mode else None
    ab = (0.1 * rng.standard_normal((1, C, H, W))).astype(np.float32) if "affine" in mode else None
    opt = lambda a: None if a is None else dev(a)
    for iters in (1, 5):
        lq = host(S.ops.meanfield_forward(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), k, iters,
                                          opt(aw), opt(ab), opt(compat)))
        ref = oracle.meanfield_forward(unary, img, -0.004, -0.003, k, iters, aw, ab, compat)
        assert_close(lq, ref, RTOL, f"logq {mode} iters={iters}")
        assert np.allclose(np.exp(lq.astype(np.float64)).sum(1), 1.0, atol=1e-5)
        assert strict_rel_err_on_marginals(lq, ref) < RTOL, f"marginals {mode} iters={iters}"
        # label maps: identical wherever the top-2 margin exceeds the tolerance
        labels = host(S.ops.argmax_labels(dev(lq)))
        srt = np.sort(ref, axis=1)
        decided = (srt[:, -1] - srt[:, -2]) > RTOL * (1 + np.abs(srt[:, -1]))
        assert (labels[decided] == ref.argmax(1)[decided]).all()
        assert decided.mean() > 0.5


@pytest.mark.parametrize("mode", ["compat", "affine+compat"])
def test_150_labels_128x128_vs_oracle(mode, S, oracle):
    """150 labels at 128x128 (H*W > C*K, the regime of the 512x512 configuration): gather in channel chunks,
    then the fused label step with the 150x150 product on the matrix cores, against the C oracle."""
    shape = (1, 150, 128, 128, 7)
    C = 150
    unary, img, _ = make_inputs(shape, 41)
    rng = np.random.default_rng(42)
    compat = ((1 - np.eye(C)) + 0.01 * rng.standard_normal((C, C))).astype(np.float32) / 8
    aw = (1 + 0.2 * rng.standard_normal((1, C, 128, 128))).astype(np.float32) if "affine" in mode else None
    ab = (0.1 * rng.standard_normal((1, C, 128, 128))).astype(np.float32) if "affine" in mode else None
    opt = lambda a: None if a is None else dev(a)
    for iters in (1, 3):
        lq = host(S.ops.meanfield_forward(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), 7, iters, opt(aw), opt(ab), dev(compat)))
        ref = oracle.meanfield_forward(unary, img, -0.004, -0.003, 7, iters, aw, ab, compat)
        assert_close(lq, ref, RTOL, f"logq iters={iters}")
        assert strict_rel_err_on_marginals(lq, ref) < RTOL


@pytest.mark.parametrize("C", [33, 70, 100, 160])
def test_label_step_channel_block_counts_vs_oracle(C, S, oracle):
    """The fused label step is built for 2 .. 5 blocks of 32 output channels: one label count per build
    (33: two blocks with one channel in the second; 160: all five full), affine + matrix, against the oracle."""
    shape = (2, C, 20, 23, 3)          # 460 pixels per image: pixel tiles straddle the image boundary
    unary, img, _ = make_inputs(shape, 61)
    rng = np.random.default_rng(62)
    compat = ((1 - np.eye(C)) + 0.01 * rng.standard_normal((C, C))).astype(np.float32) / 8
    aw = (1 + 0.2 * rng.standard_normal((1, C, 20, 23))).astype(np.float32)
    ab = (0.1 * rng.standard_normal((1, C, 20, 23))).astype(np.float32)
    lq = host(S.ops.meanfield_forward(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), 3, 2, dev(aw), dev(ab), dev(compat)))
    ref = oracle.meanfield_forward(unary, img, -0.004, -0.003, 3, 2, aw, ab, compat)
    assert_close(lq, ref, RTOL, "logq")
    assert strict_rel_err_on_marginals(lq, ref) < RTOL


@pytest.mark.parametrize("shape", [(1, 150, 176, 180, 7),      # seven blocks of output channels (six of 22, one of 18)
                                   (2, 70, 100, 90, 5),        # blocks of 18, 18, 18 and 16 channels
                                   (1, 40, 96, 100, 3)])       # two blocks of 20
def test_many_labels_channel_blocks_vs_oracle(shape, S, oracle):
    """More than 33 labels on images wide and large enough for the diagonal kernels (W >= 64 + k, H*W >= 4*C*k*k):
    the window sums are computed block by block of output channels, then the fused label step; against the C oracle."""
    B, C, H, W, k = shape
    unary, img, _ = make_inputs(shape, 71)
    rng = np.random.default_rng(72)
    compat = ((1 - np.eye(C)) + 0.01 * rng.standard_normal((C, C))).astype(np.float32) / 8
    aw = (1 + 0.2 * rng.standard_normal((1, C, H, W))).astype(np.float32)
    ab = (0.1 * rng.standard_normal((1, C, H, W))).astype(np.float32)
    for iters, affine in ((1, False), (2, True)):
        lq = host(S.ops.meanfield_forward(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), k, iters,
                                          dev(aw) if affine else None, dev(ab) if affine else None, dev(compat)))
        ref = oracle.meanfield_forward(unary, img, -0.004, -0.003, k, iters, aw if affine else None, ab if affine else None, compat)
        assert_close(lq, ref, RTOL, f"logq iters={iters}")
        assert strict_rel_err_on_marginals(lq, ref) < RTOL


def test_150_labels_full_size_properties(S):
    """150 labels, 512x512 (BASELINE config 5 image size), batch 2: marginals sum to one, and an image's result
    does not depend on what else is in the batch."""
    shape = (2, 150, 512, 512, 7)
    C = 150
    g = torch.Generator(device="cuda").manual_seed(5)
    unary = 2 * torch.randn(shape[:4], device="cuda", generator=g)
    img = torch.randn(2, 3, 512, 512, device="cuda", generator=g)
    compat = (1 - torch.eye(C, device="cuda")) / 8
    pw, cw = scalar(-0.004), scalar(-0.003)
    lq = S.ops.meanfield_forward(unary, img, pw, cw, 7, 2, None, None, compat)
    assert torch.isfinite(lq).all()
    assert (torch.exp(lq.double()).sum(1) - 1).abs().max() < 1e-5
    alone = S.ops.meanfield_forward(unary[1:2], img[1:2], pw, cw, 7, 2, None, None, compat)
    assert torch.equal(alone[0], lq[1])


def test_config4_full_image_meanfield_step_vs_oracle(S, oracle):
    """19 labels, 1024x2048, 11x11 (BASELINE config 4): fused mean-field steps of one full image (stream
    kernels, weight table, MFMA label product) against the C oracle, logits and marginals."""
    shape = (1, 19, 1024, 2048, 11)
    unary, img, _ = make_inputs(shape, 31)
    rng = np.random.default_rng(32)
    compat = ((1 - np.eye(19)) + 0.01 * rng.standard_normal((19, 19))).astype(np.float32)
    lq = host(S.ops.meanfield_forward(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), 11, 2, None, None, dev(compat)))
    ref = oracle.meanfield_forward(unary, img, -0.004, -0.003, 11, 2, None, None, compat)
    assert_close(lq, ref, RTOL, "logq")
    assert strict_rel_err_on_marginals(lq, ref) < RTOL
    labels = host(S.ops.argmax_labels(dev(lq)))
    srt = np.sort(ref, axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > RTOL * (1 + np.abs(srt[:, -1]))
    assert (labels[decided] == ref.argmax(1)[decided]).all()


def test_backward_full_512_image_vs_oracle_and_bitwise_reproducible(S, oracle):
    """21 labels, 512x512, 7x7, batch 4 (one GPU's share of the training configuration): image 2 of the batch
    against the C oracle, and the whole dx bit for bit equal between two runs (every element of the scatter's 2k copies
    of the gradient planes has one writer, or two that add onto zero; the copies are summed in a fixed order).  The
    second run finds the first run's contents in the workspace."""
    shape = (4, 21, 512, 512, 7)
    x, img, dy = make_inputs(shape, 21)
    pw, cw = scalar(-0.004), scalar(-0.003)
    dx1, dp1, _ = S.ops.local_backward(dev(x), pw, cw, 7, dev(dy), image=dev(img))
    dx2, dp2, _ = S.ops.local_backward(dev(x), pw, cw, 7, dev(dy), image=dev(img))
    assert torch.equal(dx1, dx2) and torch.equal(dp1, dp2)
    rdx, rdpw, rdcw = oracle.local_backward(x[2:3], -0.004, -0.003, 7, dy[2:3], img=img[2:3])
    assert_close(host(dx1[2:3]), rdx, RTOL, "dx of image 2")
    alone = S.ops.local_backward(dev(x[2:3]), pw, cw, 7, dev(dy[2:3]), image=dev(img[2:3]))
    assert abs(float(alone[1][0]) - rdpw) <= 2e-4 * abs(rdpw) and abs(float(alone[1][1]) - rdcw) <= 2e-4 * abs(rdcw)
    # strict elementwise relative error where the gradient is not tiny
    big = np.abs(rdx) > 1e-3 * np.abs(rdx).max()
    assert (np.abs(host(dx1[2:3]) - rdx)[big] / np.abs(rdx)[big]).max() < 1e-4


def test_backward_fixed_point_scale_follows_the_data(S, oracle):
    """Gradients 1e6 times larger or smaller keep the same relative accuracy (the diagonal kernels' scatter sums plain
    fp32; the tap-split backward of large windows derives a fixed-point scale from max|dy*x| and max|img|)."""
    shape = (1, 8, 70, 96, 5)
    x, img, dy = make_inputs(shape, 22)
    for f in (1e-6, 1.0, 1e6):
        dx, _, _ = S.ops.local_backward(dev(x), scalar(0.3), scalar(-0.7), 5, dev(dy * np.float32(f)), image=dev(img))
        rdx, _, _ = oracle.local_backward(x, 0.3, -0.7, 5, dy * np.float32(f), img=img)
        assert_close(host(dx), rdx, RTOL, f"dx, dy scaled by {f}")


def test_fit_reduces_the_loss_on_batched_tensors(S):
    """training.fit on true batched tensors: fused forward, per-image NLL with class weights and 255-ignore, the
    library's loop backward, one optimiser step per batch; the loss goes down and the checkpoint reloads."""
    from semantic_segmentation_amd import training
    B, C, H, W, k = 4, 6, 40, 72, 3
    unary, img, _ = make_inputs((B, C, H, W, k), 71)
    g = torch.Generator(device="cuda").manual_seed(7)
    labels = torch.randint(0, C, (B, H, W), device="cuda", generator=g)
    labels[torch.rand(B, H, W, device="cuda", generator=g) < 0.05] = 255
    U, I = dev(unary), dev(img)
    model = S.MeanFieldLocal(H, W, C, k, iters=2, compat=True, affine=True).cuda()
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    weight = training.class_weights(labels, C) * float((labels != 255).sum())        # (scaled to O(1))
    history = training.fit(model, opt, lambda: [(U[:2], I[:2], labels[:2]), (U[2:], I[2:], labels[2:])], epochs=6, weight=weight)
    assert len(history) == 6 and all(np.isfinite(history))
    assert history[-1] < history[0]
    for p in model.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()


def test_plan_tag_reuse_and_invalidation(S, oracle):
    """A workspace keeps its address tables between calls of the same operation and size (the table kernels find
    their tag and return); a call of another operation on the same memory invalidates the plan.  Through the C
    ABI on one caller-owned buffer: forward, forward again (plan reused), backward (other layout), forward."""
    from semantic_segmentation_amd import _lib
    lib, ptr = _lib.lib, _lib.ptr
    shape = (1, 8, 70, 96, 5)
    B, C, H, W, k = shape
    x, img, dy = make_inputs(shape, 51)
    X, I, DY = dev(x), dev(img), dev(dy)
    pw, cw = scalar(0.3), scalar(-0.7)
    n = max(int(lib.mfl_workspace_bytes(op, B, C, H, W, k, 1)) for op in (_lib.OP_LOCAL_FORWARD, _lib.OP_LOCAL_BACKWARD))
    ws = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ref = oracle.local_forward(x, 0.3, -0.7, k, img=img)
    st = _lib.stream_of(X)

    def forward():
        y = torch.full_like(X, float("nan"))
        assert lib.mfl_local_forward(ptr(X), ptr(I), None, ptr(pw), ptr(cw), ptr(y), B, C, H, W, k, ptr(ws), n, st) == 0
        return host(y)

    assert_close(forward(), ref, RTOL, "first call")
    tag = ws[:32].clone()
    assert tag.any()
    assert_close(forward(), ref, RTOL, "second call, plan reused")
    assert torch.equal(ws[:32], tag)
    dx, dpar = torch.empty_like(X), torch.empty(2, device="cuda")
    assert lib.mfl_local_backward(ptr(X), ptr(I), None, ptr(pw), ptr(cw), ptr(DY), ptr(dx), ptr(dpar), None, B, C, H, W, k,
                                  ptr(ws), n, st) == 0
    assert not torch.equal(ws[:32], tag)                       # another operation's plan now
    rdx, _, _ = oracle.local_backward(x, 0.3, -0.7, k, dy, img=img)
    assert_close(host(dx), rdx, RTOL, "backward on the shared buffer")
    assert_close(forward(), ref, RTOL, "forward after the backward: plan rebuilt")
    # the host side's kept workspaces
    _lib.keep_plans(True)
    try:
        a = host(S.ops.local_forward(X, pw, cw, k, image=I))
        b = host(S.ops.local_forward(X, pw, cw, k, image=I))
        assert np.array_equal(a, b)
        assert_close(a, ref, RTOL, "kept plan")
    finally:
        _lib.keep_plans(False)


def test_iou_counts_and_reference_conventions(S):
    """Per-class counters on device against the reference's loop (AdvancedModel.py:469-483) restated with
    numpy: exact integers; IoU -1 for an absent class, class 0 skipped unless asked for, 255 = void."""
    rng = np.random.default_rng(3)
    C = 21
    target = rng.integers(0, 19, size=(3, 96, 130)).astype(np.uint8)          # classes 19, 20 never appear
    target[rng.random(target.shape) < 0.05] = 255
    pred = np.where(rng.random(target.shape) < 0.7, target, rng.integers(0, 19, size=target.shape)).astype(np.uint8)
    pred[pred == 255] = 3
    counts = host(S.ops.iou_counts(dev(pred), dev(target), C))
    for c in range(C):
        assert counts[0, c] == np.sum((pred == c) & (target == c))
        assert counts[1, c] == np.sum(pred == c) and counts[2, c] == np.sum(target == c)
    for bg in (False, True):
        want = np.zeros(C, np.float32)
        for c in range(0 if bg else 1, C):
            inter = np.sum((pred == c) & (target == c))
            union = np.sum(pred == c) + np.sum(target == c) - inter
            want[c] = -1 if union == 0 else inter / union
        got = host(S.ops.intersection_over_union(dev(pred), dev(target), C, include_background=bg))
        assert np.allclose(got, want, rtol=1e-6, atol=0)
        m = want[want > 0]
        assert abs(S.ops.mean_intersection_over_union(dev(pred), dev(target), C, bg) - (m.mean() if len(m) else 0)) < 1e-6
    assert np.array_equal(host(S.ops.iou_counts(dev(pred), dev(target), C)), counts)       # order-free integer sums


def _loop_backward(S, d, K, affine):
    opt = lambda a: None if a is None else dev(a)
    return S.ops.meanfield_backward(dev(d["unary"]), dev(d["img"]), scalar(d["pw"]), scalar(d["cw"]), int(d["k"]), K,
                                    dev(d["G"]), opt(d.get("la_w") if affine else None),
                                    opt(d.get("la_b") if affine else None), opt(d["compat"]))


@pytest.mark.parametrize("path", golden_files("mfgrad"))
def test_meanfield_backward_golden(path, S):
    """mfl_meanfield_backward against the reference's autograd gradients through the iterated composition
    (tests/golden/make_golden.py: gen_meanfield_grad), K = 1, 2, 5, at 1e-4."""
    d = load(path)
    affine = "la_w" in d
    for K in (1, 2, 5):
        du, dpar, daw, dab, dcompat = _loop_backward(S, d, K, affine)
        assert_close(host(du), d[f"dU64_K{K}"], RTOL, f"dU K={K}")
        assert_close(host(dcompat), d[f"dcompat64_K{K}"], RTOL, f"dcompat K={K}")
        for got, key in ((dpar[0], "dpw"), (dpar[1], "dcw")):
            want = float(d[f"{key}64_K{K}"][0])
            assert abs(float(got) - want) <= 2e-4 * abs(want) + 1e-6, (key, K, float(got), want)
        if affine:
            assert_close(host(daw).reshape(d["la_w"].shape), d[f"dla_w64_K{K}"], RTOL, f"d affine w K={K}")
            assert_close(host(dab).reshape(d["la_b"].shape), d[f"dla_b64_K{K}"], RTOL, f"d affine b K={K}")


@pytest.mark.parametrize("shape,iters", [((2, 21, 64, 96, 7), 3), ((1, 19, 96, 128, 11), 2), ((2, 8, 40, 44, 5), 4)])
def test_meanfield_backward_vs_oracle(shape, iters, S, oracle):
    """Loop backward on shapes that take the diagonal and the stream kernels, against the oracle's restatement."""
    B, C, H, W, k = shape
    unary, img, _ = make_inputs(shape, 11)
    rng = np.random.default_rng(12)
    compat = ((1 - np.eye(C)) + 0.01 * rng.standard_normal((C, C))).astype(np.float32)
    aw = (1 + 0.2 * rng.standard_normal((1, C, H, W))).astype(np.float32)
    ab = (0.1 * rng.standard_normal((1, C, H, W))).astype(np.float32)
    G = rng.standard_normal((B, C, H, W)).astype(np.float32)
    du, dpar, daw, dab, dcompat = S.ops.meanfield_backward(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), k, iters,
                                                           dev(G), dev(aw), dev(ab), dev(compat))
    ref = oracle.meanfield_backward(unary, img, -0.004, -0.003, k, iters, G, aw, ab, compat)
    assert_close(host(du), ref["dU"], RTOL, "dU")
    assert_close(host(dcompat), ref["dcompat"], RTOL, "dcompat")
    assert_close(host(daw).reshape(C, H, W), ref["daw"], RTOL, "d affine w")
    assert_close(host(dab).reshape(C, H, W), ref["dab"], RTOL, "d affine b")
    assert abs(float(dpar[0]) - ref["dpw"]) <= 2e-4 * abs(ref["dpw"]) + 1e-6
    assert abs(float(dpar[1]) - ref["dcw"]) <= 2e-4 * abs(ref["dcw"]) + 1e-6
    # with the iterates of the forward handed over (no recomputation of the loop): the same gradients (to the last
    # bits only where the local backward is the order-free one)
    its = S.ops.meanfield_forward_iterates(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), k, iters, dev(aw), dev(ab),
                                           dev(compat))
    plain = S.ops.meanfield_forward(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), k, iters, dev(aw), dev(ab), dev(compat))
    assert torch.equal(its[iters - 1], plain)
    kept = S.ops.meanfield_backward(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), k, iters, dev(G), dev(aw), dev(ab),
                                    dev(compat), logq_iterates=its)
    for a, b, name in zip(kept, (du, dpar, daw, dab, dcompat), ("dU", "dparams", "daw", "dab", "dcompat")):
        assert_close(host(a), host(b), 1e-5, name + " with kept iterates")
    # the deterministic parts: everything but the local operator's scatter
    again = S.ops.meanfield_backward(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), k, iters, dev(G), dev(aw), dev(ab),
                                     dev(compat))
    assert_close(host(again[0]), host(du), 1e-6, "dU, second run")


def test_150_labels_meanfield_vs_oracle(S, oracle):
    """Large-label configuration: unfused path (gather in channel chunks, LxL transform, normalise)."""
    shape = (1, 150, 12, 14, 7)
    unary, img, _ = make_inputs(shape, 3)
    rng = np.random.default_rng(8)
    compat = ((1 - np.eye(150)) * 0.05 + 0.01 * rng.standard_normal((150, 150))).astype(np.float32)
    lq = host(S.ops.meanfield_forward(dev(unary), dev(img), scalar(-0.004), scalar(-0.003), 7, 3, None, None,
                                      dev(compat)))
    assert_close(lq, oracle.meanfield_forward(unary, img, -0.004, -0.003, 7, 3, None, None, compat), RTOL, "logq")


# ------------------------------------------------------------------ full BASELINE sizes: properties + spot checks
def test_config2_full_size_properties(S, oracle):
    """21 labels, 512x512, batch 8, 7x7 (BASELINE config 2), at full size:
    - one image against the oracle (forward of the local operator and 2 mean-field iterations),
    - batch independence: image b of the batch equals the same image run alone (the sharding premise),
    - the operator is bilinear in (weights, x^2): y(a*pw, a*cw) = a*y and y(s*x) = s^2 * y,
    - marginals sum to one."""
    shape = (8, 21, 512, 512, 7)
    unary, img, _ = make_inputs(shape, 4)
    pw, cw = scalar(-0.004), scalar(-0.003)
    U, I = dev(unary), dev(img)
    y = S.ops.local_forward(U, pw, cw, 7, image=I)
    ref = oracle.local_forward(unary[3:4], -0.004, -0.003, 7, img=img[3:4])
    assert_close(host(y[3:4]), ref, RTOL, "y[3]")
    alone = S.ops.local_forward(U[5:6].contiguous(), pw, cw, 7, image=I[5:6].contiguous())
    assert torch.equal(alone, y[5:6])
    y2 = S.ops.local_forward(U, scalar(-0.008), scalar(-0.006), 7, image=I)
    assert_close(host(y2[0]), 2 * host(y[0]), 1e-5, "linear in the weights")
    y3 = S.ops.local_forward(U * 0.5, pw, cw, 7, image=I)
    assert_close(host(y3[1]), 0.25 * host(y[1]), 1e-5, "quadratic in x")
    compat = dev(((1 - np.eye(21)) + 0.01 * np.random.default_rng(5).standard_normal((21, 21))).astype(np.float32))
    lq = S.ops.meanfield_forward(U, I, pw, cw, 7, 2, None, None, compat)
    refq = oracle.meanfield_forward(unary[6:7], img[6:7], -0.004, -0.003, 7, 2, None, None, host(compat))
    assert_close(host(lq[6:7]), refq, RTOL, "logq[6]")
    assert torch.allclose(torch.exp(lq.double()).sum(1), torch.ones(8, 512, 512, dtype=torch.float64, device="cuda"),
                          atol=1e-5)


def test_config4_shape_properties(S, oracle):
    """19 labels, 1024x2048, 11x11 (BASELINE config 4) at full image size, batch 2 (one GPU's share
    at 8 GPUs): a 64-row band of the first image against... the whole image is needed by the
    scrambled gather, so the oracle runs one full image once (about 20 s of CPU)."""
    shape = (2, 19, 1024, 2048, 11)
    unary, img, _ = make_inputs(shape, 5)
    pw, cw = scalar(-0.004), scalar(-0.003)
    y = S.ops.local_forward(dev(unary), pw, cw, 11, image=dev(img))
    ref = oracle.local_forward(unary[1:2], -0.004, -0.003, 11, img=img[1:2])
    assert_close(host(y[1:2]), ref, RTOL, "y[1]")
    alone = S.ops.local_forward(dev(unary[0:1]), pw, cw, 11, image=dev(img[0:1]))
    assert torch.equal(alone, y[0:1])


# ------------------------------------------------------------------ the drop-in module
def test_module_forward_backward_like_reference(S, oracle):
    """Conv2dLocal + color_distance used exactly as DPNModel uses them (AdvancedModel.py:135-137, 216-222)."""
    shape = (2, 21, 32, 40, 7)
    B, C, H, W, k = shape
    x, img, dy = make_inputs(shape, 6)
    calc = S.color_distance(kernel_size=(k, k), channels=3)
    layer = S.Conv2dLocal(in_height=H, in_width=W, channels=C, kernel_size=(k, k)).cuda()
    with torch.no_grad():
        layer.position_weight.fill_(0.3)
        layer.color_weight.fill_(-0.7)
    xt = dev(x).requires_grad_(True)
    out = layer(xt, calc(dev(img)))
    out.backward(dev(dy))
    rdx, rdpw, rdcw = oracle.local_backward(x, 0.3, -0.7, k, dy, img=img)
    assert_close(host(out), oracle.local_forward(x, 0.3, -0.7, k, img=img), RTOL, "module forward")
    assert_close(host(xt.grad), rdx, RTOL, "module dx")
    assert abs(layer.position_weight.grad.item() - rdpw) <= 2e-4 * abs(rdpw)
    assert abs(layer.color_weight.grad.item() - rdcw) <= 2e-4 * abs(rdcw)
    # tensor form, as the reference passes it
    out2 = layer(dev(x), calc(dev(img)).materialize())
    assert_close(host(out2), host(out), 1e-5, "tensor form")


def test_meanfield_module_gradients_match_composed_torch_ops(S):
    """MeanFieldLocal under autograd (fused forward + the library's loop backward) against the same loop
    composed per iteration from the HIP local operator and differentiable torch ops."""
    B, C, H, W, k = 2, 5, 20, 24, 3
    unary, img, _ = make_inputs((B, C, H, W, k), 7)
    mod = S.MeanFieldLocal(H, W, C, k, iters=3, compat=True, affine=True).cuda()
    with torch.no_grad():
        mod.local_convolution_layer.position_weight.fill_(-0.004)
        mod.local_convolution_layer.color_weight.fill_(-0.003)
        mod.local_activation_function.weight.add_(0.1 * torch.randn_like(mod.local_activation_function.weight))
        fused = mod(dev(unary), dev(img))
    target = torch.randint(0, C, (B, H, W), device="cuda")
    U = dev(unary).requires_grad_(True)
    out = mod(U, dev(img))
    assert_close(host(out), host(fused), 1e-6, "forward under autograd vs inference")
    torch.nn.functional.nll_loss(out, target).backward()
    got = {n: p.grad.clone() for n, p in mod.named_parameters()}
    got["U"] = U.grad.clone()
    # the composed loop
    for p in mod.parameters():
        p.grad = None
    U2 = dev(unary).requires_grad_(True)
    local = mod.local_convolution_layer
    handle = S.ColorDistanceHandle(dev(img), local.kernel_size)
    q, ls = U2, torch.nn.functional.log_softmax(U2, dim=1)
    for _ in range(3):
        msg = torch.einsum("ij,bjhw->bihw", mod.compat, mod.local_activation_function(local(q, handle)))
        logq = torch.nn.functional.log_softmax(ls - msg, dim=1)
        q = torch.exp(logq)
    torch.nn.functional.nll_loss(logq, target).backward()
    want = {n: p.grad for n, p in mod.named_parameters()}
    want["U"] = U2.grad
    for n in want:
        assert got[n] is not None and torch.isfinite(got[n]).all(), n
        assert_close(host(got[n]), host(want[n]), 2e-4, n)
