"""GPU parity: calc_prob vs the CPU oracle.  Tolerance 1e-5 absolute
(north_star) on values in [0,1]; gradients relative to max(1,|g|)."""
import numpy as np
import pytest
import torch

import inputs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


CASES = {
    "uniform_32x32": lambda: inputs.uniform_prob((1, 1, 32, 32, 256)),
    "binary_32x32": lambda: inputs.binary_prob((1, 1, 32, 32, 256)),
    "uniform_full": lambda: inputs.uniform_prob((1, 1, 128, 128, 256)),
    "binary_full": lambda: inputs.binary_prob((1, 1, 128, 128, 256)),
    "batch_nc": lambda: inputs.uniform_prob((2, 3, 8, 5, 256), seed=11),
    "long_ray_600": lambda: inputs.binary_prob((1, 1, 4, 7, 600), seed=12),
    "short_ray_36": lambda: inputs.uniform_prob((1, 1, 6, 6, 36), seed=13),
    "odd_ray_37": lambda: inputs.uniform_prob((1, 2, 3, 5, 37), seed=14),      # generic kernel (Z % 4 != 0)
    "single_sample": lambda: inputs.uniform_prob((1, 1, 3, 3, 1), seed=15),
}


@pytest.mark.parametrize("name", list(CASES))
def test_forward_backward(name, genre, oracle, dev):
    p = CASES[name]()
    s_o = oracle.calc_prob_forward(p)
    g = np.random.default_rng(9).standard_normal(p.shape).astype(np.float32)
    w = s_o * g                                               # calc_prob.py:27 (fp32 product)
    gp_o = oracle.calc_prob_backward(p, w)
    pt = t(p, dev).requires_grad_(True)
    s = genre.CalcStopProb.apply(pt)
    assert np.abs(s.detach().cpu().numpy() - s_o).max() <= TOL
    s.backward(t(g, dev))
    diff = np.abs(pt.grad.cpu().numpy() - gp_o) / np.maximum(1.0, np.abs(gp_o))
    assert diff.max() <= TOL, diff.max()


def test_unfused_backward_entry(genre, oracle, dev):
    """the reference's 3-argument calc_prob_backward (calc_prob.h:2) is kept as is"""
    from genre_shapehd_amd.toolbox.calc_prob.calc_prob._ext import calc_prob_lib
    p = inputs.binary_prob((1, 1, 16, 16, 256), seed=21)
    s_o = oracle.calc_prob_forward(p)
    w = s_o * np.random.default_rng(9).standard_normal(p.shape).astype(np.float32)
    gp_o = oracle.calc_prob_backward(p, w)
    out = torch.empty_like(t(p, dev))
    calc_prob_lib.calc_prob_backward(t(p, dev), t(w, dev), out)
    diff = np.abs(out.cpu().numpy() - gp_o) / np.maximum(1.0, np.abs(gp_o))
    assert diff.max() <= TOL


def test_strided_views(genre, oracle, dev):
    """non-contiguous in/out (permuted, sliced) go through the generic kernel"""
    from genre_shapehd_amd.toolbox.calc_prob.calc_prob._ext import calc_prob_lib
    p = inputs.uniform_prob((2, 1, 6, 5, 64), seed=31)
    s_o = oracle.calc_prob_forward(p)
    base = torch.zeros((2, 1, 6, 64, 5), device=dev)
    pv = base.permute(0, 1, 2, 4, 3)                            # z stride 5
    pv.copy_(t(p, dev))
    out = torch.zeros((2, 1, 6, 5, 128), device=dev)[..., ::2]  # z stride 2
    calc_prob_lib.calc_prob_forward(pv, out)
    assert np.abs(out.cpu().numpy() - s_o).max() <= TOL
    # sliced along y: vec4 path with a non-collapsible ray pitch
    p2 = inputs.uniform_prob((1, 1, 4, 8, 256), seed=32)
    s2 = oracle.calc_prob_forward(p2[:, :, :, ::2].copy())
    out2 = torch.empty((1, 1, 4, 4, 256), device=dev)
    calc_prob_lib.calc_prob_forward(t(p2, dev)[:, :, :, ::2], out2)
    assert np.abs(out2.cpu().numpy() - s2).max() <= TOL


def test_properties_full_size(genre, dev):
    """size-independent properties at BASELINE size x batch 8: telescoping identity
    sum_z s[z] + prod_z (1-p[z]) == 1, monotone survival, and linearity of the adjoint."""
    p = torch.from_numpy(inputs.uniform_prob((8, 1, 128, 128, 256), seed=41)).to(dev)
    s = genre.CalcStopProb.apply(p)
    total = s.double().sum(-1) + (1.0 - p.double()).prod(-1)
    assert (total - 1.0).abs().max().item() <= 1e-5
    assert (s >= 0).all() and (s <= p).all()
    from genre_shapehd_amd.toolbox.calc_prob.calc_prob._ext import calc_prob_lib
    g1 = torch.randn_like(p)
    g2 = torch.randn_like(p)
    o1, o2, o12 = torch.empty_like(p), torch.empty_like(p), torch.empty_like(p)
    calc_prob_lib.calc_prob_backward_fused(p, s, g1, o1)
    calc_prob_lib.calc_prob_backward_fused(p, s, g2, o2)
    calc_prob_lib.calc_prob_backward_fused(p, s, g1 + g2, o12)
    rel = ((o1 + o2 - o12).abs() / (1 + o12.abs())).max().item()
    assert rel <= 1e-4, rel
