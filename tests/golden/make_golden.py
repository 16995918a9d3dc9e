#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in a container that has the reference checkout (default /root/reference):

    python tests/golden/make_golden.py [--reference /root/reference]

The reference's Python never travels with this repository: this script imports
it at generation time, feeds it seeded synthetic inputs and stores only inputs
and outputs (.npz).  Nothing here restates reference code; the arithmetic is
done by the imported modules (LocalConvolution.Conv2dLocal,
AdvancedModel.DPNModel and its helpers).

AdvancedModel.py imports torchvision (transforms, vgg16) at module scope, which
is not installed here.  Neither is touched by DPNModel.forward, so an in-memory
placeholder module is registered for the duration of the import; predict()/fit()
pre-processing, which would need the real torchvision, is not exercised.

Each case is run twice: in float32 (what the reference produces as shipped) and
with float64 inputs (the reference code is dtype-agnostic apart from its float32
parameters), the latter to pin index maps and formulas far below fp32 rounding.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference(path):
    sys.path.insert(0, path)
    os.environ.setdefault("MPLBACKEND", "Agg")
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        md = types.ModuleType("torchvision.models")

        class _Placeholder:
            def __init__(self, *a, **k):
                pass

            def __call__(self, x):
                raise RuntimeError("torchvision is not available in this container")

        for name in ("Compose", "Resize", "ToTensor", "Normalize"):
            setattr(tr, name, _Placeholder)
        md.vgg16 = _Placeholder
        tv.transforms, tv.models = tr, md
        sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.models": md})
    import LocalConvolution
    import AdvancedModel
    return LocalConvolution, AdvancedModel


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def gen_color_distance(AM):
    g = torch.Generator().manual_seed(101)
    for (H, W, k) in [(5, 7, 1), (6, 5, 3), (7, 9, 4), (9, 8, 7), (4, 6, 11)]:
        img = torch.randn(2, 3, H, W, generator=g)
        fn = AM.DPNModel._color_distance(kernel_size=(k, k), channels=3)
        save(f"cd_h{H}_w{W}_k{k}", img=npy(img), k=np.int64(k),
             cd=npy(fn(img)), cd64=npy(fn(img.double())))


def run_local(LC, x, cd, pw, cw, k, dy, dtype):
    B, C, H, W = x.shape
    m = LC.Conv2dLocal(H, W, C, k)
    with torch.no_grad():
        m.position_weight.fill_(pw)
        m.color_weight.fill_(cw)
    xx = x.to(dtype).clone().requires_grad_(True)
    y = m(xx, cd.to(dtype))
    y.backward(dy.to(dtype))
    return (npy(y), npy(xx.grad), npy(m.position_weight.grad), npy(m.color_weight.grad),
            npy(m.position_distances))


def gen_local(LC):
    g = torch.Generator().manual_seed(202)
    cases = [  # B, C, H, W, k, pw, cw
        (2, 2, 5, 7, 3, 0.3, -0.7),
        (1, 3, 6, 5, 4, 0.3, -0.7),
        (2, 19, 8, 12, 7, -0.004, -0.003),
        (1, 21, 12, 9, 11, 0.3, -0.7),
        (1, 4, 6, 6, 1, 0.3, -0.7),
        (1, 5, 4, 4, 6, -0.004, -0.003),      # C*K > L: a row of the scrambled matrix is a fraction of one window set
        (1, 21, 16, 16, 7, -0.004, -0.003),
    ]
    for (B, C, H, W, k, pw, cw) in cases:
        x = torch.randn(B, C, H, W, generator=g)
        cd = torch.rand(B, H * W, k * k, generator=g) * 3
        dy = torch.randn(B, C, H, W, generator=g)
        y, dx, dpw, dcw, pd = run_local(LC, x, cd, pw, cw, k, dy, torch.float32)
        y64, dx64, dpw64, dcw64, _ = run_local(LC, x, cd, pw, cw, k, dy, torch.float64)
        save(f"local_c{C}_h{H}_w{W}_k{k}", x=npy(x), cd=npy(cd), dy=npy(dy), k=np.int64(k),
             pw=np.float32(pw), cw=np.float32(cw), pd=pd,
             y=y, dx=dx, dpw=dpw, dcw=dcw, y64=y64, dx64=dx64, dpw64=dpw64, dcw64=dcw64)


def gen_tail(AM):
    """a5 -> a8 -> a6 -> a7 with explicit parameters (AdvancedModel.py:223-244)."""
    g = torch.Generator().manual_seed(303)
    B, L, M, H, W = 2, 4, 5, 10, 11
    la = AM.LinearActivation(L, H, W)
    conv = nn.Conv2d(L, L * M, kernel_size=(9, 9), padding=(4, 4))
    ga = AM.LinearActivation(L * M, H, W)
    pool = AM.BlockMinPooling(kernel_size=M, dim=1)
    with torch.no_grad():
        for p in (la.weight, la.bias, conv.weight, conv.bias, ga.weight, ga.bias):
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
    t = torch.randn(B, L, H, W, generator=g)
    unary = torch.randn(B, L, H, W, generator=g) * 2
    with torch.no_grad():
        a = la(t)
        c = ga(conv(a))
        smooth = pool(c)
        out = nn.LogSoftmax(dim=1)(torch.nn.functional.log_softmax(unary, dim=1) - smooth)
    save("tail_l4_m5", t=npy(t), unary=npy(unary), mixtures=np.int64(M),
         la_w=npy(la.weight), la_b=npy(la.bias), conv_w=npy(conv.weight), conv_b=npy(conv.bias),
         ga_w=npy(ga.weight), ga_b=npy(ga.bias),
         a=npy(a), c=npy(c), smooth=npy(smooth), out=npy(out))


class _FixedUnary(nn.Module):
    """Stands in for the VGG-like unary stack: returns a stored synthetic unary."""

    def __init__(self, unary):
        super().__init__()
        self.unary = unary

    def forward(self, x):
        return self.unary


def gen_dpn(AM):
    """DPNModel.forward(phase=2,3,4) with the unary CNN bypassed (AdvancedModel.py:198-244).
    The model hard-codes a 50x50 window (AdvancedModel.py:135-137)."""
    g = torch.Generator().manual_seed(404)
    H, W, L, M = 8, 9, 4, 5
    torch.manual_seed(404)
    model = AM.DPNModel(H, W, num_labels=L, num_mixture_filters=M)
    unary = torch.randn(2, L, H, W, generator=g) * 2
    img = torch.randn(2, 3, H, W, generator=g)
    model.unary_terms_layers = _FixedUnary(unary)
    with torch.no_grad():
        # local_activation_function is never initialised by the reference (torch.empty).
        model.local_activation_function.weight.copy_(1 + 0.2 * torch.randn(1, L, H, W, generator=g))
        model.local_activation_function.bias.copy_(0.1 * torch.randn(1, L, H, W, generator=g))
        model.local_convolution_layer.position_weight.fill_(-0.004)
        model.local_convolution_layer.color_weight.fill_(-0.003)
        outs = {f"phase{p}": npy(model(img, phase=p)) for p in (2, 3, 4)}
    sd = model.state_dict()
    save("dpn_h8_w9_l4", img=npy(img), unary=npy(unary), mixtures=np.int64(M), k=np.int64(50),
         pw=np.float32(-0.004), cw=np.float32(-0.003),
         la_w=npy(sd["local_activation_function.weight"]), la_b=npy(sd["local_activation_function.bias"]),
         conv_w=npy(sd["global_convolution_layer.weight"]), conv_b=npy(sd["global_convolution_layer.bias"]),
         ga_w=npy(sd["global_activation_function.weight"]), ga_b=npy(sd["global_activation_function.bias"]),
         pd=npy(sd["local_convolution_layer.position_distances"]), **outs)


def gen_meanfield(LC, AM):
    """K-iteration composition: every arithmetic step is an imported reference module,
    only the for-loop is new.  compat is Conv2d(L, L, 1, bias=False) holding an LxL matrix."""
    g = torch.Generator().manual_seed(505)
    for (B, L, H, W, k, pw, cw, affine) in [(2, 3, 9, 11, 3, -0.004, -0.003, True),
                                            (1, 19, 10, 14, 7, -0.004, -0.003, False),
                                            (1, 5, 8, 8, 4, 0.05, -0.02, True)]:
        unary = torch.randn(B, L, H, W, generator=g) * 2
        img = torch.randn(B, 3, H, W, generator=g)
        compat = (1 - torch.eye(L)) + 0.01 * torch.randn(L, L, generator=g)
        la = AM.LinearActivation(L, H, W)
        with torch.no_grad():
            la.weight.copy_(1 + 0.2 * torch.randn(1, L, H, W, generator=g))
            la.bias.copy_(0.1 * torch.randn(1, L, H, W, generator=g))
        extra = {"la_w": npy(la.weight), "la_b": npy(la.bias)} if affine else {}
        for dtype, tag in ((torch.float32, ""), (torch.float64, "64")):
            local = LC.Conv2dLocal(H, W, L, k)
            conv = nn.Conv2d(L, L, kernel_size=1, bias=False)
            with torch.no_grad():
                local.position_weight.fill_(pw)
                local.color_weight.fill_(cw)
                conv.weight.copy_(compat.reshape(L, L, 1, 1))
            conv = conv.to(dtype)
            act = la.to(dtype) if affine else (lambda t: t)
            cdf = AM.DPNModel._color_distance(kernel_size=(k, k), channels=3)
            U = unary.to(dtype)
            with torch.no_grad():
                cd = cdf(img.to(dtype))
                q = U
                for it in range(1, 11):
                    msg = conv(act(local(q, cd)))
                    logq = nn.LogSoftmax(dim=1)(torch.nn.functional.log_softmax(U, dim=1) - msg)
                    q = torch.exp(logq)
                    if it in (1, 2, 5, 10):
                        extra[f"logq{tag}_it{it}"] = npy(logq)
        save(f"meanfield_l{L}_h{H}_w{W}_k{k}", unary=npy(unary), img=npy(img), compat=npy(compat),
             k=np.int64(k), pw=np.float32(pw), cw=np.float32(cw), **extra)


def gen_meanfield_grad(LC, AM):
    """Autograd gradients through the same K-iteration composition (what loss.backward(),
    AdvancedModel.py:299, derives), for K in {1, 2, 5}, on the inputs of the three meanfield_* cases:
    loss = sum(logq_K * G) with a stored cotangent G."""
    import glob
    for ci, path in enumerate(sorted(glob.glob(os.path.join(HERE, "meanfield_l*.npz")))):
        d = dict(np.load(path, allow_pickle=False))
        k, pw, cw = int(d["k"]), float(d["pw"]), float(d["cw"])
        B, L, H, W = d["unary"].shape
        affine = "la_w" in d
        g = torch.Generator().manual_seed(606 + ci)
        G = torch.randn(B, L, H, W, generator=g)
        out = {key: d[key] for key in ("unary", "img", "compat", "k", "pw", "cw")}
        if affine:
            out.update(la_w=d["la_w"], la_b=d["la_b"])
        out["G"] = npy(G)
        for dtype, tag in ((torch.float32, ""), (torch.float64, "64")):
            for K in (1, 2, 5):
                local = LC.Conv2dLocal(H, W, L, k)
                conv = nn.Conv2d(L, L, kernel_size=1, bias=False)
                la = AM.LinearActivation(L, H, W)
                with torch.no_grad():
                    local.position_weight.fill_(pw)
                    local.color_weight.fill_(cw)
                    conv.weight.copy_(torch.from_numpy(d["compat"]).reshape(L, L, 1, 1))
                    if affine:
                        la.weight.copy_(torch.from_numpy(d["la_w"]))
                        la.bias.copy_(torch.from_numpy(d["la_b"]))
                conv, la = conv.to(dtype), la.to(dtype)
                act = la if affine else (lambda t: t)
                cdf = AM.DPNModel._color_distance(kernel_size=(k, k), channels=3)
                U = torch.from_numpy(d["unary"]).to(dtype).requires_grad_(True)
                cd = cdf(torch.from_numpy(d["img"]).to(dtype))
                q = U
                for _ in range(K):
                    msg = conv(act(local(q, cd)))
                    logq = nn.LogSoftmax(dim=1)(torch.nn.functional.log_softmax(U, dim=1) - msg)
                    q = torch.exp(logq)
                (logq * G.to(dtype)).sum().backward()
                sfx = f"{tag}_K{K}"
                out[f"logq{sfx}"] = npy(logq)
                out[f"dU{sfx}"] = npy(U.grad)
                out[f"dpw{sfx}"] = npy(local.position_weight.grad)
                out[f"dcw{sfx}"] = npy(local.color_weight.grad)
                out[f"dcompat{sfx}"] = npy(conv.weight.grad.reshape(L, L))
                if affine:
                    out[f"dla_w{sfx}"] = npy(la.weight.grad)
                    out[f"dla_b{sfx}"] = npy(la.bias.grad)
        save("mfgrad_" + os.path.basename(path)[len("meanfield_"):-len(".npz")], **out)


def _metrics_stub(AM, num_labels):
    """An object with just what DPNModel's metric helpers read from `self` (AdvancedModel.py:427-517); building a
    whole DPNModel (139 M parameters) for them would add nothing."""
    stub = types.SimpleNamespace(num_labels=num_labels, image_size=None)
    stub.calc_intersection_over_union = lambda *a, **k: AM.DPNModel.calc_intersection_over_union(stub, *a, **k)
    return stub


def gen_metrics(AM):
    """DPNModel.calc_intersection_over_union / calc_mean_intersection_over_union (AdvancedModel.py:427-485) and
    DPNModel._calc_weights (AdvancedModel.py:500-517) on seeded label maps (tensors / arrays: the PIL branches only
    resize)."""
    rng = np.random.default_rng(707)
    C = 21
    out = {"num_labels": np.int64(C)}
    for ci, (shape, present) in enumerate([((3, 96, 130), 19), ((1, 40, 40), 21), ((2, 17, 23), 5)]):
        target = rng.integers(0, present, size=shape).astype(np.uint8)       # classes >= present never appear
        target[rng.random(shape) < 0.05] = 255                               # void
        pred = np.where(rng.random(shape) < 0.7, target, rng.integers(0, present, size=shape)).astype(np.uint8)
        pred[pred == 255] = 3
        stub = _metrics_stub(AM, C)
        out[f"pred{ci}"], out[f"target{ci}"] = pred, target
        for bg in (False, True):
            iou = AM.DPNModel.calc_intersection_over_union(stub, torch.from_numpy(pred.astype(np.int64)),
                                                           torch.from_numpy(target.astype(np.int64)), bg)
            mean = AM.DPNModel.calc_mean_intersection_over_union(stub, torch.from_numpy(pred.astype(np.int64)),
                                                                 torch.from_numpy(target.astype(np.int64)), bg)
            out[f"iou{ci}_bg{int(bg)}"] = npy(iou)
            out[f"miou{ci}_bg{int(bg)}"] = np.float64(mean)
        # the reference takes a list of annotated images; np.array(image) is all it does with each
        out[f"weights{ci}"] = npy(AM.DPNModel._calc_weights(stub, [target[b] for b in range(shape[0])]))
    save("metrics_l21", **out)


def gen_tail_grad(AM):
    """Autograd gradients through the label-context tail a5 -> a8 -> a6 -> a7 (AdvancedModel.py:223-244; what
    loss.backward(), AdvancedModel.py:299, derives in phases 3-4): loss = sum(out * G)."""
    g = torch.Generator().manual_seed(808)
    for (B, L, M, H, W, ks) in [(2, 4, 5, 10, 11, 9), (1, 7, 3, 14, 9, 9), (2, 6, 2, 9, 40, 5)]:
        la = AM.LinearActivation(L, H, W)
        conv = nn.Conv2d(L, L * M, kernel_size=(ks, ks), padding=(ks // 2, ks // 2))
        ga = AM.LinearActivation(L * M, H, W)
        pool = AM.BlockMinPooling(kernel_size=M, dim=1)
        with torch.no_grad():
            for p in (la.weight, la.bias, conv.weight, conv.bias, ga.weight, ga.bias):
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        t0 = torch.randn(B, L, H, W, generator=g)
        u0 = torch.randn(B, L, H, W, generator=g) * 2
        G = torch.randn(B, L, H, W, generator=g)
        out = dict(t=npy(t0), unary=npy(u0), G=npy(G), mixtures=np.int64(M), ks=np.int64(ks),
                   la_w=npy(la.weight), la_b=npy(la.bias), conv_w=npy(conv.weight), conv_b=npy(conv.bias),
                   ga_w=npy(ga.weight), ga_b=npy(ga.bias))
        for dtype, tag in ((torch.float32, ""), (torch.float64, "64")):
            la, conv, ga = la.to(dtype), conv.to(dtype), ga.to(dtype)
            for p in list(la.parameters()) + list(conv.parameters()) + list(ga.parameters()):
                p.grad = None
            t = t0.detach().clone().to(dtype).requires_grad_(True)
            u = u0.detach().clone().to(dtype).requires_grad_(True)
            smooth = pool(ga(conv(la(t))))
            res = nn.LogSoftmax(dim=1)(torch.nn.functional.log_softmax(u, dim=1) - smooth)
            (res * G.to(dtype)).sum().backward()
            out.update({f"out{tag}": npy(res), f"dt{tag}": npy(t.grad), f"dunary{tag}": npy(u.grad),
                        f"dla_w{tag}": npy(la.weight.grad), f"dla_b{tag}": npy(la.bias.grad),
                        f"dconv_w{tag}": npy(conv.weight.grad), f"dconv_b{tag}": npy(conv.bias.grad),
                        f"dga_w{tag}": npy(ga.weight.grad), f"dga_b{tag}": npy(ga.bias.grad)})
        save(f"tailgrad_l{L}_m{M}_k{ks}", **out)


class _GradUnary(nn.Module):
    """Stands in for the unary stack: a stored synthetic unary that takes a gradient."""

    def __init__(self, unary):
        super().__init__()
        self.unary = nn.Parameter(unary)

    def forward(self, x):
        return self.unary


def gen_dpn_grad(AM):
    """Autograd gradients of DPNModel.forward(phase=3) and (phase=4) with the unary CNN bypassed
    (AdvancedModel.py:198-244, 299): the model's own 50x50 window, loss = sum(output * G)."""
    g = torch.Generator().manual_seed(909)
    H, W, L, M = 8, 9, 4, 5
    torch.manual_seed(909)
    model = AM.DPNModel(H, W, num_labels=L, num_mixture_filters=M)
    unary = torch.randn(2, L, H, W, generator=g) * 2
    img = torch.randn(2, 3, H, W, generator=g)
    G = torch.randn(2, L, H, W, generator=g)
    model.unary_terms_layers = _GradUnary(unary.clone())
    with torch.no_grad():
        model.local_activation_function.weight.copy_(1 + 0.2 * torch.randn(1, L, H, W, generator=g))
        model.local_activation_function.bias.copy_(0.1 * torch.randn(1, L, H, W, generator=g))
        model.local_convolution_layer.position_weight.fill_(-0.004)
        model.local_convolution_layer.color_weight.fill_(-0.003)
        # (the reference's init leaves the context layers at ~1e-2: scaled up so every gradient is well above rounding)
        model.global_convolution_layer.weight.mul_(5)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    out = dict(img=npy(img), unary=npy(unary), G=npy(G), mixtures=np.int64(M), k=np.int64(50), ks=np.int64(9),
               pw=np.float32(-0.004), cw=np.float32(-0.003),
               la_w=npy(sd["local_activation_function.weight"]), la_b=npy(sd["local_activation_function.bias"]),
               conv_w=npy(sd["global_convolution_layer.weight"]), conv_b=npy(sd["global_convolution_layer.bias"]),
               ga_w=npy(sd["global_activation_function.weight"]), ga_b=npy(sd["global_activation_function.bias"]))
    names = {"unary_terms_layers.unary": "dunary", "local_convolution_layer.position_weight": "dpw",
             "local_convolution_layer.color_weight": "dcw", "local_activation_function.weight": "dla_w",
             "local_activation_function.bias": "dla_b", "global_convolution_layer.weight": "dconv_w",
             "global_convolution_layer.bias": "dconv_b", "global_activation_function.weight": "dga_w",
             "global_activation_function.bias": "dga_b"}
    for phase in (3, 4):
        model.zero_grad()
        res = model(img, phase=phase)
        (res * G).sum().backward()
        out[f"out_p{phase}"] = npy(res)
        for pname, key in names.items():
            p = dict(model.named_parameters())[pname]
            out[f"{key}_p{phase}"] = npy(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
    save("dpngrad_h8_w9_l4", **out)


def gen_fit_step(LC, AM):
    """One optimiser step's worth of gradients the way DPNModel.fit accumulates them (AdvancedModel.py:295-303):
    single-image forwards, loss = NLLLoss(weight=_calc_weights, ignore_index=255)(prediction, label) / batch_size,
    backward per image, gradients summed over the batch_size images.  The forward is the reference's phase-4 data
    flow with the label-context block an LxL matrix (Conv2d(L, L, 1)), K iterations of imported modules."""
    g = torch.Generator().manual_seed(1010)
    B, L, H, W, k, K = 4, 6, 12, 80, 3, 2
    unary = torch.randn(B, L, H, W, generator=g) * 2
    img = torch.randn(B, 3, H, W, generator=g)
    labels = torch.randint(0, L, (B, H, W), generator=g)
    labels[torch.rand(B, H, W, generator=g) < 0.05] = 255
    labels[3] = 255                                         # an image without a valid pixel cannot go through the
    labels[3, :4] = 1                                       # reference's loss (NaN): keep a few valid ones
    compat = (1 - torch.eye(L)) + 0.01 * torch.randn(L, L, generator=g)
    la_w = 1 + 0.2 * torch.randn(1, L, H, W, generator=g)
    la_b = 0.1 * torch.randn(1, L, H, W, generator=g)
    stub = _metrics_stub(AM, L)
    weight = AM.DPNModel._calc_weights(stub, [labels[b].numpy() for b in range(B)])
    out = dict(unary=npy(unary), img=npy(img), labels=labels.numpy().astype(np.int64), compat=npy(compat), la_w=npy(la_w),
               la_b=npy(la_b), k=np.int64(k), iters=np.int64(K), pw=np.float32(-0.004), cw=np.float32(-0.003),
               weight=npy(weight))
    for dtype, tag in ((torch.float32, ""), (torch.float64, "64")):
        local = LC.Conv2dLocal(H, W, L, k)
        conv = nn.Conv2d(L, L, kernel_size=1, bias=False)
        la = AM.LinearActivation(L, H, W)
        with torch.no_grad():
            local.position_weight.fill_(-0.004)
            local.color_weight.fill_(-0.003)
            conv.weight.copy_(compat.reshape(L, L, 1, 1))
            la.weight.copy_(la_w)
            la.bias.copy_(la_b)
        conv, la = conv.to(dtype), la.to(dtype)
        loss_function = nn.NLLLoss(weight=weight.to(dtype), ignore_index=255)
        cdf = AM.DPNModel._color_distance(kernel_size=(k, k), channels=3)
        U = unary.detach().clone().to(dtype).requires_grad_(True)
        total = 0.0
        for b in range(B):                                   # batch_size = B single-image micro-batches
            u = U[b:b + 1]
            cd = cdf(img[b:b + 1].to(dtype))
            q = u
            for _ in range(K):
                msg = conv(la(local(q, cd)))
                logq = nn.LogSoftmax(dim=1)(torch.nn.functional.log_softmax(u, dim=1) - msg)
                q = torch.exp(logq)
            loss = loss_function(logq, labels[b:b + 1]) / B
            loss.backward()
            total += loss.item()
        out.update({f"loss{tag}": np.float64(total), f"dU{tag}": npy(U.grad), f"dpw{tag}": npy(local.position_weight.grad),
                    f"dcw{tag}": npy(local.color_weight.grad), f"dcompat{tag}": npy(conv.weight.grad.reshape(L, L)),
                    f"dla_w{tag}": npy(la.weight.grad), f"dla_b{tag}": npy(la.bias.grad)})
    save("fitstep_l6_h12_w80_k3", **out)


def gen_preprocess():
    """DPNModel._preprocess (AdvancedModel.py:487-498) = transforms.Resize((H, W)) -> ToTensor -> Normalize.
    torchvision is not installed here; on a PIL image its Resize is `image.resize((W, H), Image.BILINEAR)`, ToTensor is
    uint8 HWC -> float CHW / 255 and Normalize is (x - mean) / std in float32: those calls are made on real Pillow and
    torch below ("restated from torchvision's documented behaviour": the fixture pins Pillow's resampling, which is
    where all the arithmetic of the resize lives).  Labels: `image.resize(size)` on a palette image = nearest
    (AdvancedModel.py:265)."""
    from PIL import Image
    rng = np.random.default_rng(1111)
    out = {}
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    for ci, (H0, W0, H, W) in enumerate([(37, 53, 64, 64), (120, 90, 64, 64), (64, 64, 64, 64), (75, 100, 32, 96), (50, 64, 100, 64)]):
        # smooth + noise, so that both interpolation and antialiasing matter
        yy, xx = np.mgrid[0:H0, 0:W0]
        base = 127 + 100 * np.sin(xx / 7.0 + ci)[..., None] * np.cos(yy / 5.0)[..., None] * np.array([1.0, 0.7, -0.8])
        a = np.clip(base + rng.normal(0, 25, (H0, W0, 3)), 0, 255).astype(np.uint8)
        resized = np.array(Image.fromarray(a, "RGB").resize((W, H), Image.BILINEAR))
        t = torch.from_numpy(resized).permute(2, 0, 1).contiguous().float().div(255)
        t = t.sub(mean).div(std)
        lab = rng.integers(0, 21, (H0, W0)).astype(np.uint8)
        lab[rng.random((H0, W0)) < 0.05] = 255
        lab_resized = np.array(Image.fromarray(lab, "P").resize((W, H)))
        out.update({f"image{ci}": a, f"resized{ci}": resized, f"tensor{ci}": npy(t), f"labels{ci}": lab,
                    f"labels_resized{ci}": lab_resized, f"size{ci}": np.array([H, W], np.int64)})
    save("preprocess_rgb", **out)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, help="run one generator (e.g. meanfield_grad)")
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    torch.set_num_threads(4)
    LC, AM = import_reference(args.reference)
    only = {"meanfield_grad": lambda: gen_meanfield_grad(LC, AM), "metrics": lambda: gen_metrics(AM),
            "tail_grad": lambda: gen_tail_grad(AM), "dpn_grad": lambda: gen_dpn_grad(AM),
            "fit_step": lambda: gen_fit_step(LC, AM), "preprocess": gen_preprocess}
    if args.only is not None:
        only[args.only]()
        return
    gen_color_distance(AM)
    gen_local(LC)
    gen_tail(AM)
    gen_dpn(AM)
    gen_meanfield(LC, AM)
    gen_meanfield_grad(LC, AM)
    gen_metrics(AM)
    gen_tail_grad(AM)
    gen_dpn_grad(AM)
    gen_fit_step(LC, AM)
    gen_preprocess()


if __name__ == "__main__":
    main()
