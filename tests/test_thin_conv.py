"""networks/thin_conv.py on the host: the blocked-GEMM weight gradients (and the data gradients written as the adjoint
convolution) against autograd through the stock operators, float64, including the two layer shapes they exist for at reduced
size (ConvTranspose3d(C -> 1, k4 s2 p1), Conv3d(2 -> C, k8 s2 p3)), odd sizes, chunked taps and ragged voxel blocks."""
import copy

import pytest
import torch

import genre_shapehd_amd  # noqa: F401
from genre_shapehd_amd.networks import thin_conv as TC


@pytest.mark.parametrize("cls,cin,cout,k,s,p,size", [
    ("t", 6, 1, 4, 2, 1, (8, 8, 8)), ("t", 5, 2, 4, 2, 1, (5, 6, 7)), ("t", 3, 1, 4, 1, 0, (1, 1, 1)), ("t", 4, 3, 3, 1, 1, (6, 5, 4)),
    ("c", 2, 7, 8, 2, 3, (16, 16, 16)), ("c", 1, 5, 4, 2, 1, (10, 12, 14)), ("c", 3, 4, 3, 1, 1, (7, 6, 5)), ("c", 2, 3, 4, 2, 1, (9, 11, 13))])
@pytest.mark.parametrize("chunked", [False, True])
def test_thin_convolutions_equal_the_stock_operators(cls, cin, cout, k, s, p, size, chunked, monkeypatch):
    if chunked:
        monkeypatch.setattr(TC, "_MAX_UNFOLD", 1)          # one slab of taps per pass
        monkeypatch.setattr(TC, "_BLOCK", 8)
    torch.manual_seed(cin * 100 + cout * 10 + k)
    mod = (TC.ThinConvTranspose3d if cls == "t" else TC.ThinConv3d)(cin, cout, k, s, p).double()
    ref = copy.deepcopy(mod)
    mod.force_custom = True
    ref.force_stock = True
    x = torch.randn((3, cin) + size, dtype=torch.float64)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = mod(xa), ref(xb)
    assert torch.equal(ya, yb)
    g = torch.randn_like(ya)
    ya.backward(g)
    yb.backward(g)
    for a, b, what in ((xa.grad, xb.grad, "dx"), (mod.weight.grad, ref.weight.grad, "dw"), (mod.bias.grad, ref.bias.grad, "db")):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 1e-12 * max(1.0, b.abs().max().item()), what


def test_thin_modules_keep_the_state_dict_and_the_stock_path_off_the_gpu():
    import torch.nn as nn
    a, b = TC.ThinConvTranspose3d(4, 1, 4, 2, 1), nn.ConvTranspose3d(4, 1, 4, 2, 1)
    assert list(a.state_dict()) == list(b.state_dict())
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 4, 3, 3, 3)
    assert torch.equal(a(x), b(x))
    assert not TC._custom_path(x, a, 128 ** 3)                 # CPU tensors: stock operator, differentiable twice
    y = a(x.requires_grad_(True))
    gx, = torch.autograd.grad(y.sum(), x, create_graph=True)
    gx.pow(2).sum().backward()
    assert a.weight.grad is not None


@pytest.mark.parametrize("cls,cin,cout,k,s,p,size", [("c", 1, 4, 4, 2, 1, (8, 8, 8)), ("c", 3, 2, 4, 2, 1, (7, 9, 8)),
                                                      ("t", 3, 2, 4, 2, 1, (4, 5, 3))])
def test_thin_convolutions_under_a_gradient_penalty(cls, cin, cout, k, s, p, size, monkeypatch):
    """models/wgangp.py:131-147: the critic's INPUT gradient is taken with create_graph=True and a function of it is
    differentiated again with respect to the weights -- the second differentiation runs through the mirror-image Function
    (and its GEMM weight gradient) and must equal the stock operators'"""
    torch.manual_seed(5 + cin)
    mod = (TC.ThinConvTranspose3d if cls == "t" else TC.ThinConv3d)(cin, cout, k, s, p).double()
    ref = copy.deepcopy(mod)
    mod.force_custom = True
    ref.force_stock = True
    x = torch.randn((2, cin) + size, dtype=torch.float64)
    res = []
    calls = []
    for name in ("transposed_weight_grad", "regular_weight_grad"):       # count the GEMM weight gradients
        fn = getattr(TC, name)
        monkeypatch.setattr(TC, name, lambda *a, _fn=fn, **k: (calls.append(1), _fn(*a, **k))[1])
    for m in (mod, ref):
        xi = x.clone().requires_grad_(True)
        score = m(xi).tanh().sum((1, 2, 3, 4))
        # the first-order pass wants the input gradient only (models/shapehd.py: WGANGP._penalty does the same): no weight-
        # gradient GEMM runs in it -- ctx.needs_input_grad alone cannot tell (fixed at forward time)
        with TC.input_grad_only():
            gx, = torch.autograd.grad(score, xi, torch.ones_like(score), create_graph=True, retain_graph=True)
        assert not calls
        pen = ((gx.reshape(2, -1).norm(2, dim=1) - 1) ** 2).mean() + score.mean()
        pen.backward()
        assert bool(calls) == (m is mod)                                 # ... the second differentiation does run them
        del calls[:]
        res.append((gx.detach(), m.weight.grad.clone(), m.bias.grad.clone(), xi.grad.clone()))
    for a, b, what in zip(res[0], res[1], ("gx", "dw", "db", "dx")):
        assert (a - b).abs().max().item() <= 1e-11 * max(1.0, b.abs().max().item()), what
