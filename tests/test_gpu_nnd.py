"""GPU parity: nndistance vs the CPU oracle (= my_lib.c as shipped).
idx bit-exact, dist bit-exact (same un-fused fp32 expression), gradients 1e-5."""
import numpy as np
import pytest
import torch

import inputs

pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


SHAPES = [(1, 2048, 2048), (1, 50, 50), (3, 777, 1301), (2, 1, 5), (2, 5, 1), (4, 64, 17), (1, 1000, 3000),
          (8, 2048, 2048)]


@pytest.mark.parametrize("b,n,m", SHAPES)
def test_forward_exact_backward_close(b, n, m, genre, oracle, dev):
    x1, x2 = inputs.clouds(b, n, m, seed1=10 * n + b, seed2=10 * m + b + 1)
    d1o, d2o, i1o, i2o = oracle.nnd_forward(x1, x2)
    a = t(x1, dev).requires_grad_(True)
    bb = t(x2, dev).requires_grad_(True)
    d1, d2, i1, i2 = genre.nndistance_w_idx(a, bb)
    assert i1.dtype == torch.int32 and i2.dtype == torch.int32
    assert np.array_equal(i1.cpu().numpy(), i1o) and np.array_equal(i2.cpu().numpy(), i2o)
    assert np.array_equal(d1.detach().cpu().numpy(), d1o) and np.array_equal(d2.detach().cpu().numpy(), d2o)
    rng = np.random.default_rng(77)
    g1 = rng.standard_normal(d1o.shape).astype(np.float32)
    g2 = rng.standard_normal(d2o.shape).astype(np.float32)
    gx1o, gx2o = oracle.nnd_backward(x1, x2, g1, g2, i1o, i2o)
    (d1 * t(g1, dev)).sum().add((d2 * t(g2, dev)).sum()).backward()
    assert np.abs(a.grad.cpu().numpy() - gx1o).max() <= 1e-5
    assert np.abs(bb.grad.cpu().numpy() - gx2o).max() <= 1e-5


@pytest.mark.parametrize("b,n,m,slices", [(512, 128, 128, 8), (1024, 128, 100, 4), (4096, 128, 128, 2), (64, 2048, 2048, 4)])
def test_every_slice_count(b, n, m, slices, genre, oracle, dev):
    """the launcher gives a workgroup 16, 8, 4 or 2 waves depending on how many workgroups the batch makes
    (nnd.hip, genre_nnd_forward); all of them must agree bit for bit with the serial scan, ties included"""
    x1, x2 = inputs.clouds(b, n, m, seed1=b + 1, seed2=b + 2)
    x1[::3] = np.round(x1[::3] * 4) / 4                    # a third of the batches on a coarse lattice: exact ties
    x2[::3] = np.round(x2[::3] * 4) / 4
    d1o, d2o, i1o, i2o = oracle.nnd_forward(x1, x2)
    d1, d2, i1, i2 = genre.nndistance_w_idx(t(x1, dev), t(x2, dev))
    assert np.array_equal(i1.cpu().numpy(), i1o) and np.array_equal(i2.cpu().numpy(), i2o)
    assert np.array_equal(d1.cpu().numpy(), d1o) and np.array_equal(d2.cpu().numpy(), d2o)


def test_ties_first_minimum_wins(genre, oracle, dev):
    """integer lattice clouds are full of exact ties: the lowest index must win (my_lib.c:20)"""
    rng = np.random.default_rng(5)
    x1 = rng.integers(0, 4, (2, 300, 3)).astype(np.float32)
    x2 = rng.integers(0, 4, (2, 500, 3)).astype(np.float32)
    d1o, d2o, i1o, i2o = oracle.nnd_forward(x1, x2)
    d1, d2, i1, i2 = genre.nndistance_w_idx(t(x1, dev), t(x2, dev))
    assert np.array_equal(i1.cpu().numpy(), i1o) and np.array_equal(i2.cpu().numpy(), i2o)
    assert np.array_equal(d1.cpu().numpy(), d1o) and np.array_equal(d2.cpu().numpy(), d2o)


def test_transposed_input_and_score(genre, oracle, dev):
    x1, x2 = inputs.clouds(2, 100, 120, 3, 4)
    d1o, d2o, _, _ = oracle.nnd_forward(x1, x2)
    d1, d2 = genre.NNDModule()(t(x1, dev).transpose(1, 2).contiguous(), t(x2, dev))    # [B,3,n] form
    assert np.array_equal(d1.cpu().numpy(), d1o) and np.array_equal(d2.cpu().numpy(), d2o)
    score = genre.nndistance_score(t(x1, dev), t(x2, dev)).cpu().numpy()
    ref = np.sqrt(d1o + 1e-10).mean(1) + np.sqrt(d2o + 1e-10).mean(1)
    assert np.abs(score - ref).max() <= 1e-5


def test_self_distance_zero(genre, dev):
    """property at a large size: a cloud against itself has dist 0 and idx = identity (distinct points)"""
    x = torch.rand((4, 8192, 3), device=dev)
    d1, d2, i1, i2 = genre.nndistance_w_idx(x, x.clone())
    assert (d1 == 0).all() and (d2 == 0).all()
    ar = torch.arange(8192, device=dev, dtype=torch.int32).expand(4, -1)
    assert (i1 == ar).all() and (i2 == ar).all()


@pytest.mark.parametrize("b,n,m", [(1, 2048, 2048), (2, 300, 1000), (1, 17, 2048), (5, 33, 16)])
def test_small_problem_kernel_with_exact_ties(b, n, m, genre, oracle, dev):
    """few 128-query workgroups (a lone cloud pair, configs[0]) take the 16-queries-per-workgroup kernel whose minima
    are merged on (distance bits, index) keys: the lowest index among equal distances must win, as in my_lib.c:19"""
    x1, x2 = inputs.clouds(b, n, m, seed1=3 * n + b, seed2=5 * m + b)
    x1 = (np.round(x1 * 6) / 6).astype(np.float32)             # coarse lattice: many exactly equal distances
    x2 = (np.round(x2 * 6) / 6).astype(np.float32)
    d1o, d2o, i1o, i2o = oracle.nnd_forward(x1, x2)
    d1, d2, i1, i2 = genre.nndistance_w_idx(t(x1, dev), t(x2, dev))
    assert np.array_equal(i1.cpu().numpy(), i1o) and np.array_equal(i2.cpu().numpy(), i2o)
    assert np.array_equal(d1.cpu().numpy(), d1o) and np.array_equal(d2.cpu().numpy(), d2o)
