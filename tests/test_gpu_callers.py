"""GPU parity of the GenRe caller glue (first step) against the reference's Python lines
restated on CPU torch + the oracle (oracle/torch_oracle.py: GenReGlueCPU)."""
import numpy as np
import pytest
import torch

import inputs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_refiner_input_forward_backward(genre, oracle, dev):
    from genre_shapehd_amd.callers import GenReGeometry
    from oracle.torch_oracle import GenReGlueCPU
    rng = np.random.default_rng(31)
    n = 2
    core = np.concatenate([1 - inputs.sph_depth_map(seed=40 + i) for i in range(n)]).astype(np.float32)   # 1 - d
    sph = np.pad(core, ((0, 0), (0, 0), (16, 16), (16, 16)), mode="edge")
    proj = (rng.uniform(-0.2, 60.0, (n, 1, 128, 128, 128))).astype(np.float32)         # both sides of the clamp
    g = rng.standard_normal((n, 2, 128, 128, 128)).astype(np.float32)
    cpu = GenReGlueCPU(oracle)
    sc = torch.from_numpy(sph).requires_grad_(True)
    pc = torch.from_numpy(proj).requires_grad_(True)
    ref, cnt_ref = cpu.refiner_input(sc, pc)
    ref.backward(torch.from_numpy(g))
    geo = GenReGeometry().to(dev)
    sg = torch.from_numpy(sph).to(dev).requires_grad_(True)
    pg = torch.from_numpy(proj).to(dev).requires_grad_(True)
    out, cnt = geo.refiner_input(sg, pg)
    assert out.shape == (n, 2, 128, 128, 128)
    assert np.array_equal(cnt.cpu().numpy(), cnt_ref.numpy())
    o, r = out.detach().cpu().numpy(), ref.detach().numpy()
    # channel 1 is one torch elementwise op; ROCm torch divides by a scalar via the reciprocal (1 ulp vs CPU)
    assert np.abs(o[:, 1] - r[:, 1]).max() <= 2e-7
    assert np.abs(o[:, 0] - r[:, 0]).max() <= 128 * TOL
    single = cnt_ref.numpy()[:, 0] <= 1
    assert np.array_equal(o[:, 0][single], r[:, 0][single])        # unique summation order: bit-exact
    out.backward(torch.from_numpy(g).to(dev))
    assert np.abs(pg.grad.cpu().numpy() - pc.grad.numpy()).max() <= 1e-6    # a handful of mask flips at the bounds aside

    d = np.abs(sg.grad.cpu().numpy() - sc.grad.numpy()) / np.maximum(1.0, np.abs(sc.grad.numpy()))
    assert d.max() <= 128 * TOL, d.max()
    assert (sg.grad[:, :, :16].abs().sum() == 0) and (sg.grad[:, :, :, -16:].abs().sum() == 0)   # margin: no gradient


def test_depth_to_spherical(genre, oracle, dev):
    from genre_shapehd_amd.callers import GenReGeometry
    from oracle.torch_oracle import GenReGlueCPU
    d = inputs.batch_depth(2)
    ref_pd, ref_sph = GenReGlueCPU(oracle).depth_to_spherical(torch.from_numpy(d))
    geo = GenReGeometry().to(dev)
    pd, sph = geo.depth_to_spherical(torch.from_numpy(d).to(dev))
    assert sph.shape == (2, 1, 160, 160)
    assert (pd.cpu() - ref_pd).abs().max().item() <= 50 * 128 * TOL
    assert (sph.cpu() - ref_sph).abs().max().item() <= TOL
