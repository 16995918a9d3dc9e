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
