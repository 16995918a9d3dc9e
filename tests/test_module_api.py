"""Host-side mirror of the reference's operator interface (no GPU needed)."""
import numpy as np
import pytest
import torch

from conftest import golden_files, load
import semantic_segmentation_amd as S


def test_constructor_attributes_and_state_dict():
    m = S.Conv2dLocal(in_height=12, in_width=9, channels=21, kernel_size=11)
    assert (m.channels, m.kernel_size, m.in_height, m.in_width) == (21, (11, 11), 12, 9)
    sd = m.state_dict()
    assert list(sd) == ["color_weight", "position_weight", "position_distances"]
    assert sd["color_weight"].shape == (1,) and sd["color_weight"].dtype == torch.float32
    assert sd["position_distances"].dtype == torch.int64
    d = load([p for p in golden_files("local") if "k11" in p][0])
    assert np.array_equal(sd["position_distances"].numpy(), d["pd"])     # the reference's own buffer
    assert abs(m.position_weight.item()) <= 0.005 and abs(m.color_weight.item()) <= 0.005
    # a reference state dict loads as is
    m.load_state_dict({"color_weight": torch.tensor([0.25]), "position_weight": torch.tensor([-0.5]),
                       "position_distances": torch.from_numpy(d["pd"])})
    assert m.color_weight.item() == 0.25


def test_kernel_size_parsing():
    assert S.Conv2dLocal(4, 4, 2, (3, 3)).kernel_size == (3, 3)
    with pytest.raises(ValueError):
        S.Conv2dLocal(4, 4, 2, (3, 3, 3))
    m = S.Conv2dLocal(4, 4, 2, (3, 5))            # constructs, like the reference; forward raises
    assert m.position_distances.numel() == 15
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 4, 4), torch.zeros(1, 16, 15))


def test_forward_argument_errors():
    m = S.Conv2dLocal(4, 5, 3, 3)
    with pytest.raises(RuntimeError):              # colour distance is effectively mandatory
        m(torch.zeros(1, 3, 4, 5))
    with pytest.raises(RuntimeError):              # wrong spatial size
        m(torch.zeros(1, 3, 5, 5), torch.zeros(1, 25, 9))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 4, 5), torch.zeros(1, 20, 9))
    handle = S.color_distance(kernel_size=(3, 3), channels=3)(torch.zeros(1, 3, 4, 5))
    assert isinstance(handle, S.ColorDistanceHandle) and handle.shape == (1, 20, 9)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 4, 5), handle)
    other = S.color_distance(kernel_size=(5, 5))(torch.zeros(1, 3, 4, 5))
    with pytest.raises(RuntimeError, match="different window"):
        m(torch.zeros(1, 3, 4, 5), other)


def test_meanfield_module_parameters():
    m = S.MeanFieldLocal(8, 8, 5, 3, iters=4, compat=True, affine=True)
    names = [n for n, _ in m.named_parameters()]
    assert "local_convolution_layer.position_weight" in names and "local_convolution_layer.color_weight" in names
    assert m.compat.shape == (5, 5) and m.local_activation_function.weight.shape == (1, 5, 8, 8)
    with pytest.raises(ValueError):
        S.MeanFieldLocal(8, 8, 5, 3, iters=0)


def test_label_context_module_mirrors_the_reference_state_dict_keys():
    import semantic_segmentation_amd as S
    m = S.LabelContext(8, 9, num_labels=4, num_mixture_filters=5)
    keys = set(m.state_dict())
    assert keys == {"local_activation_function.weight", "local_activation_function.bias",
                    "global_convolution_layer.weight", "global_convolution_layer.bias",
                    "global_activation_function.weight", "global_activation_function.bias"}
    assert m.global_convolution_layer.weight.shape == (20, 4, 9, 9)
    assert m.global_activation_function.weight.shape == (1, 20, 8, 9)


def test_per_image_nll_equals_the_reference_accumulation():
    """Mean of per-image NLLLoss(weight, ignore_index=255) = sum over images of loss / batch_size
    (AdvancedModel.py:295-299); class weights as DPNModel._calc_weights (AdvancedModel.py:500-517)."""
    import torch
    from semantic_segmentation_amd import training
    g = torch.Generator().manual_seed(1)
    B, L, H, W = 3, 5, 6, 7
    logp = torch.log_softmax(torch.randn(B, L, H, W, generator=g), dim=1)
    labels = torch.randint(0, 4, (B, H, W), generator=g)          # class 4 never appears
    labels[torch.rand(B, H, W, generator=g) < 0.2] = 255
    w = training.class_weights(labels, L)
    counts = torch.tensor([(labels == c).sum() for c in range(L)], dtype=torch.float32)
    assert torch.equal(w, torch.where(counts > 0, 1 / counts.clamp(min=1), torch.zeros(L)))
    for weight in (None, w):
        loss_fn = torch.nn.NLLLoss(weight=weight, ignore_index=255)
        want = sum(loss_fn(logp[b:b + 1], labels[b:b + 1]) / B for b in range(B))
        got = training.per_image_nll(logp, labels, weight)
        assert abs(float(got) - float(want)) < 1e-6


def test_checkpoint_format_round_trip(tmp_path):
    import torch
    import semantic_segmentation_amd as S
    from semantic_segmentation_amd import training
    m = S.LabelContext(4, 5, num_labels=3, num_mixture_filters=2, kernel_size=3)
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9)
    path = str(tmp_path / "ckpt.pt")
    training.save_checkpoint(path, m, opt, epoch=7, phase=4, loss=1.5)
    raw = torch.load(path, weights_only=True)
    assert set(raw) == {"model_state_dict", "optimizer_state_dict", "epoch", "phase", "loss"}
    m2 = S.LabelContext(4, 5, num_labels=3, num_mixture_filters=2, kernel_size=3)
    info = training.load_checkpoint(path, m2, torch.optim.SGD(m2.parameters(), lr=0.1, momentum=0.9))
    assert info["optimizer_loaded"] and info["epoch"] == 7 and not info["missing"] and not info["unexpected"]
    for a, b in zip(m.state_dict().values(), m2.state_dict().values()):
        assert torch.equal(a, b)
    # a checkpoint of a bigger model (extra keys, as a DPNModel's) loads non-strictly
    sd = dict(m.state_dict())
    sd["unary_terms_layers.0.weight"] = torch.zeros(2)
    torch.save({"model_state_dict": sd, "optimizer_state_dict": {"state": {}, "param_groups": []}}, path)
    info = training.load_checkpoint(path, m2, torch.optim.SGD(m2.parameters(), lr=0.1))
    assert info["unexpected"] == ["unary_terms_layers.0.weight"] and not info["optimizer_loaded"]


def test_resample_tables_reproduce_pillow_fixture():
    """Host logic of the device preprocessing: the windows and fixed-point coefficients of preprocess.resample_tables,
    applied in integer arithmetic with numpy, give Pillow's resized 8-bit image bit for bit (fixture made by Pillow,
    tests/golden/make_golden.py: gen_preprocess); nearest_indices gives Pillow's nearest-neighbour label resize."""
    import numpy as np
    from conftest import golden_files, load
    from semantic_segmentation_amd import preprocess as P
    d = load(golden_files("preprocess")[0])

    def axis_pass(img, axis, out_size):
        coef, bounds = P.resample_tables(img.shape[axis], out_size)
        img = np.moveaxis(img, axis, 0).astype(np.int64)
        out = np.empty((out_size,) + img.shape[1:], np.uint8)
        for i in range(out_size):
            lo, n = bounds[i]
            acc = (1 << (P.PRECISION_BITS - 1)) + np.tensordot(coef[i, :n].astype(np.int64), img[lo:lo + n], axes=1)
            out[i] = np.clip(acc >> P.PRECISION_BITS, 0, 255)
        return np.moveaxis(out, 0, axis)

    for ci in range(5):
        a, (H, W) = d[f"image{ci}"], d[f"size{ci}"]
        r = a if a.shape[1] == W else axis_pass(a, 1, int(W))
        r = r if r.shape[0] == H else axis_pass(r, 0, int(H))
        assert np.array_equal(r, d[f"resized{ci}"]), ci
        lab = d[f"labels{ci}"]
        got = lab[P.nearest_indices(lab.shape[0], int(H))][:, P.nearest_indices(lab.shape[1], int(W))]
        assert np.array_equal(got, d[f"labels_resized{ci}"]), ci


def test_dpn_head_keeps_the_reference_state_dict_keys_and_phase_schedule():
    import torch
    import semantic_segmentation_amd as S
    unary_net = torch.nn.Conv2d(3, 4, 1)
    head = S.DPNHead(8, 9, num_labels=4, num_mixture_filters=5, unary_terms_layers=unary_net)
    keys = set(head.state_dict())
    for k in ("local_convolution_layer.color_weight", "local_convolution_layer.position_weight",
              "local_convolution_layer.position_distances", "local_activation_function.weight",
              "global_convolution_layer.weight", "global_convolution_layer.bias", "global_activation_function.bias",
              "unary_terms_layers.weight"):
        assert k in keys
    assert head.global_convolution_layer.weight.shape == (20, 4, 9, 9)
    trainable = lambda m: all(p.requires_grad for p in m.parameters())
    # AdvancedModel.py:364-377
    head.set_phase_trainable(0)
    assert trainable(unary_net) and trainable(head.local_convolution_layer)
    head.set_phase_trainable(1)
    assert not trainable(unary_net) and trainable(head.local_convolution_layer)
    head.set_phase_trainable(2)
    assert not trainable(head.local_convolution_layer) and not trainable(head.local_activation_function)
    assert trainable(head.global_convolution_layer)
    head.set_phase_trainable(3)
    assert trainable(unary_net) and trainable(head.local_convolution_layer) and trainable(head.local_activation_function)
    with pytest.raises(ValueError):
        head.forward_from_unary(torch.zeros(1, 4, 8, 9), torch.zeros(1, 3, 8, 9), phase=5)
