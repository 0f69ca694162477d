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


@pytest.mark.parametrize("n,h,w", [(2, 256, 256), (3, 40, 72), (1, 33, 17)])
def test_get_abs_depth_matches_reference_lines(n, h, w, genre, dev):
    """GenReGeometry.get_abs_depth == depth_pred_with_sph_inpaint.py:131-142 run on CPU torch: plain fp32
    elementwise arithmetic in the same order (true division) -> bit-identical values; gradient w.r.t. the
    predicted depth to 1 ulp-level (autograd's division by a scalar may be a reciprocal multiply)"""
    from genre_shapehd_amd.callers import GenReGeometry
    from oracle.torch_oracle import GenReGlueCPU
    rng = np.random.default_rng(11)
    pred = torch.from_numpy(rng.uniform(0, 100, (n, 1, h, w)).astype(np.float32))
    sil = torch.from_numpy((rng.uniform(0, 100, (n, 1, h, w))).astype(np.float32))
    sil[:, :, : h // 3] = 0.0                                              # background rows
    mm = torch.from_numpy(np.stack([rng.uniform(1.5, 1.9, n), rng.uniform(2.4, 2.9, n)], 1).astype(np.float32))
    a = pred.clone().requires_grad_(True)
    ref = GenReGlueCPU.get_abs_depth(a, mm, sil)
    geo = GenReGeometry().to(dev)
    b = pred.to(dev).requires_grad_(True)
    out = geo.get_abs_depth(b, mm.to(dev), sil.to(dev))
    assert out.shape == ref.shape == (n, 1, w, h)
    assert torch.equal(out.cpu(), ref.detach())
    g = torch.from_numpy(rng.standard_normal((n, 1, w, h)).astype(np.float32))
    ref.backward(g)
    out.backward(g.to(dev))
    assert (b.grad.cpu() - a.grad).abs().max().item() <= 2e-7 * max(1.0, a.grad.abs().max().item())
    # strided inputs (a channel slice) go through the same kernel
    wide = torch.stack([pred, pred + 1], 1).reshape(n, 2, h, w).to(dev)
    out2 = geo.get_abs_depth(wide[:, 0:1], mm.to(dev), sil.to(dev))
    assert torch.equal(out2, out.detach())
