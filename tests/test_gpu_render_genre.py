"""GPU parity of the FUSED render_spherical on the input class configs[1] / bench.py feed it:
near-binary GenRe occupancy volumes (tests/inputs.py: genre_offclamp_volumes), forward AND gradient, against

  * the reference's op sequence on CPU torch + the C oracle's calc_prob (oracle/torch_oracle.py:
    RenderSphericalCPU, i.e. toolbox/spherical_proj.py:62-72 with calc_prob_kernel.cu:113-189), and
  * the EXACT value of that fp32-defined operator (RenderSphericalExact: ATen's float32 sample values and float32
    trilinear weights -- they are part of the operator's definition -- with every product, scan and accumulation in
    float64).  The reference's fp32 chain is itself 1e-5 ... 1.3e-4 away from it (fp32 accumulation of up to 2^17
    contributions per voxel around the centre and the polar axis, 255 fp32-rounded recursion steps in
    calc_prob_kernel.cu:169-187): measured and printed per image.

Two input classes, same bars (measured on MI355X: kernels 2e-6 ... 9e-6 from exact, the fp32 chain 8e-5 ... 2.7e-4):
  "sharp"  empty 3e-5*(1+u), solid 1 - 3e-5*(1+u'): GenRe's actual levels, lifted just off the clamp bounds
           (1/(1-p) ~ 3e4 amplification behind the surface).
  "soft"   solid level 1 - 0.02*(1+u').
Checked per image: map 1e-5 against both references; gradient within 1e-5 * max(s', |g|) per voxel of the exact value,
s' = the image's own upstream gradient scale (1 ... 2^-24 across the batch) times max(1, pre_scale) -- an image whose
gradient is 2^-24 of its neighbour's must still be resolved (per-image fixed-point scale / fp64 tiles); and no further
from the fp32 chain than the chain is from exact.
"""
import numpy as np
import pytest
import torch

import inputs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _batch_minor(t):
    n, c, x, y, z = t.shape
    out = torch.empty_strided((n, c, x, y, z), (1, n * x * y * z, y * z * n, z * n, n), dtype=t.dtype, device=t.device)
    out.copy_(t)
    return out


@pytest.fixture(scope="module")
def volumes(oracle):
    return {"sharp": inputs.genre_offclamp_volumes(oracle, 32),
            "soft": inputs.genre_offclamp_volumes(oracle, 32, seed=60, solid_margin=0.02)}


def _reference(oracle, vol, g, pre_scale, pad, f64):
    """forward + gradient of the reference chain for ONE image (batch items are independent)"""
    from oracle.torch_oracle import RenderSphericalCPU, RenderSphericalExact, sph_pad
    x = torch.from_numpy(vol).requires_grad_(True)
    v = x if pre_scale is None else torch.clamp(x * pre_scale, 1e-5, 1 - 1e-5)     # depth_pred_with_sph_inpaint.py:124
    out = (RenderSphericalExact() if f64 else RenderSphericalCPU(oracle))(v)
    if pad:
        out = sph_pad(out, pad)                                                  # :126
    out.backward(torch.from_numpy(g).to(out.dtype))
    return out.detach().float(), x.grad


def _g_scales(n):
    """upstream gradient scale per image: 1 ... 2^-24 across the batch"""
    return np.exp2(-24.0 * np.arange(n) / max(n - 1, 1)).astype(np.float32)


@pytest.mark.parametrize("cls", ["sharp", "soft"])
@pytest.mark.parametrize("n,pre_scale,pad,layout", [
    (1, None, 0, "std"), (8, 50.0, 16, "std"), (32, None, 16, "std"),
    (32, 50.0, 16, "bm"), (19, None, 0, "bm"), (40, 50.0, 0, "bm"),
])
def test_fused_render_gradient_on_genre_class_volumes(n, pre_scale, pad, layout, cls, volumes, genre, oracle, dev):
    from genre_shapehd_amd.toolbox import _fused_render
    assert _fused_render.available()
    vols = np.concatenate([volumes[cls], volumes[cls][:8]])[:n].copy()
    if pre_scale is not None:
        vols = (vols / np.float32(pre_scale)).astype(np.float32)       # the kernel (and the reference) re-scale it
    rng = np.random.default_rng(7 + n)
    side = 128 + 2 * pad
    scales = _g_scales(n)
    g = (rng.standard_normal((n, 1, side, side)).astype(np.float32) * scales[:, None, None, None]).astype(np.float32)
    x = torch.from_numpy(vols).to(dev)
    if layout == "bm":
        x = _batch_minor(x)
        assert x.stride(0) == 1
    x.requires_grad_(True)
    mod = genre.render_spherical(fused=True).to(dev)
    out = mod(x, pre_scale=pre_scale, pad=pad)
    out.backward(torch.from_numpy(g).to(dev))
    torch.cuda.synchronize()
    got_out, got_grad = out.detach().cpu(), x.grad.cpu()
    assert torch.isfinite(got_grad).all()
    check = sorted(set([0, n - 1]))
    for i in check:
        s = float(scales[i])
        f64_out, f64_grad = _reference(oracle, vols[i:i + 1], g[i:i + 1], pre_scale, pad, f64=True)
        ref_out, ref_grad = _reference(oracle, vols[i:i + 1], g[i:i + 1], pre_scale, pad, f64=False)
        assert (got_out[i] - ref_out[0]).abs().max().item() <= TOL, (i, "map vs fp32 reference chain")
        assert (got_out[i] - f64_out[0]).abs().max().item() <= TOL, (i, "map vs float64")
        s = s * max(1.0, pre_scale or 1.0)                                      # d/dx of clamp(x * pre_scale)
        den = f64_grad[0].abs().clamp(min=s)
        e_k = (got_grad[i] - f64_grad[0]).abs() / den                           # kernel vs float64
        e_r = (ref_grad[0] - f64_grad[0]).abs() / den                           # fp32 reference chain vs float64
        rel, ref_rel = e_k.max().item(), e_r.max().item()
        err = ((got_grad[i] - ref_grad[0]).abs() / ref_grad[0].abs().clamp(min=s)).max().item()
        ax = torch.arange(128, dtype=torch.float32) - 63.5
        rad = (ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2).sqrt()[None]
        shells = ["r<%d: k %.1e ref %.1e" % (hi_, e_k[(rad >= lo_) & (rad < hi_)].max().item(),
                                              e_r[(rad >= lo_) & (rad < hi_)].max().item())
                  for lo_, hi_ in ((0, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 200))]
        print("   error by distance from the centre (kernel, fp32 chain):", "; ".join(shells))
        print("%s image %d scale %.1e: kernel-vs-exact %.2e, kernel-vs-fp32-chain %.2e, fp32-chain-vs-exact %.2e"
              % (cls, i, s, rel, err, ref_rel))
        # the kernels: within 1e-5 of the exact value of the operator on BOTH classes, relative to the image's own scale
        assert rel <= TOL, (i, "gradient vs the exact value, relative to the image's gradient scale", rel, s)
        # and no further from the reference's fp32 chain than that chain is from the exact value
        assert err <= ref_rel + 2 * TOL, (i, "gradient vs fp32 reference chain", err, ref_rel)


@pytest.mark.parametrize("layout", ["std", "bm"])
def test_non_finite_upstream_gradient_is_not_swallowed(layout, genre, dev):
    """an Inf / NaN in grad_out must come out as NaN in grad_vox of THAT image (the reference chain propagates it;
    a fixed-point tile or an fmaxf-based scale would turn it into finite garbage) and leave the other images alone"""
    n = 16
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.uniform(0.001, 0.05, (n, 1, 128, 128, 128)).astype(np.float32)).to(dev)
    if layout == "bm":
        x = _batch_minor(x)
    x.requires_grad_(True)
    out = genre.render_spherical(fused=True).to(dev)(x)
    g = torch.from_numpy(rng.standard_normal(tuple(out.shape)).astype(np.float32)).to(dev)
    g[3, 0, 10, 20] = float("inf")
    g[5, 0, 100, 7] = float("nan")
    out.backward(g)
    grad = x.grad
    for i in range(n):
        finite = torch.isfinite(grad[i]).all().item()
        assert finite == (i not in (3, 5)), (i, finite)
    assert torch.isnan(grad[3]).any() and torch.isnan(grad[5]).any()


@pytest.mark.parametrize("n,res,sph,zr,pre_scale,pad", [(16, 32, 24, 64, None, 0), (32, 32, 24, 64, None, 0),
                                                       (17, 33, 20, 100, 3.0, 4), (40, 24, 16, 32, None, 8),
                                                       (16, 13, 8, 12, 2.0, 0)])
def test_batch_minor_small_geometry_against_oracle(n, res, sph, zr, pre_scale, pad, genre, oracle, dev):
    """the batch-minor kernels (forward and backward) directly against the CPU reference chain at sizes the oracle
    finishes in a second, every image checked: volumes that are not a multiple of the 4x8x8 brick (partial bricks,
    tiles that stick out of the volume), batches that are not a multiple of 32 (a partly filled image group, two
    groups), short rays, the folded clamp and the padded map"""
    from oracle.torch_oracle import RenderSphericalCPU, RenderSphericalExact, sph_pad
    rng = np.random.default_rng(90 + n + res)
    ax = (np.arange(res) + 0.5) / res - 0.5
    vols = np.empty((n, 1, res, res, res), np.float32)
    for i in range(n):
        c = (rng.random(3) - 0.5) * 0.3
        r2 = (ax[:, None, None] - c[0]) ** 2 + (ax[None, :, None] - c[1]) ** 2 + (ax[None, None, :] - c[2]) ** 2
        vols[i, 0] = np.clip(0.002 + 0.95 * (r2 < (0.12 + 0.1 * rng.random()) ** 2) + rng.uniform(0, 0.01, r2.shape), 2e-5, 1 - 2e-5)
    if pre_scale:
        vols = (vols / np.float32(pre_scale) * np.float32(1.2)).astype(np.float32)   # some voxels clamp at the top
    vc = torch.from_numpy(vols).requires_grad_(True)
    vin = vc if not pre_scale else torch.clamp(vc * pre_scale, 1e-5, 1 - 1e-5)
    ref = RenderSphericalCPU(oracle, sph, zr)(vin)
    if pad:
        ref = sph_pad(ref, pad)
    g = torch.from_numpy(rng.standard_normal(ref.shape).astype(np.float32))
    # gradient yardstick: the exact value of the operator (the fp32 chain itself is ~1e-4 off, see the module docstring)
    ex = RenderSphericalExact(sph, zr)(vin)
    (sph_pad(ex, pad) if pad else ex).backward(g.double())
    xb = _batch_minor(torch.from_numpy(vols).to(dev)).requires_grad_(True)
    out = genre.render_spherical(sph, zr, fused=True).to(dev)(xb, pre_scale=pre_scale, pad=pad)
    out.backward(g.to(dev))
    assert xb.grad.stride(0) == 1
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= TOL
    err = ((xb.grad.cpu() - vc.grad).abs() / vc.grad.abs().clamp(min=max(1.0, pre_scale or 1.0))).max().item()
    assert err <= TOL, err


def test_clamp_boundary_gradient_is_characterised(genre, oracle, dev):
    """The documented deviation: on a volume that was clamped BEFORE it is rendered -- the
    reference's own call, render_spherical(clamp(proj * 50, 1e-5, 1 - 1e-5)), depth_pred_with_sph_inpaint.py:124 --
    every solid voxel sits exactly on the upper bound, a sample inside the solid interpolates eight equal values with
    fp32 weights that sum to 1 +- 1 ulp, and whether the second clamp (spherical_proj.py:66) passes its gradient is
    decided by that last bit: ATen's summation order and ours disagree on some samples.  This test gives the
    deviation a number on the real chain (analytic sphere depth -> cam_bp -> shift -> x50 -> clamp): the forward maps
    agree to 1e-5, and the gradient w.r.t. the clamped volume differs from the reference's fp32 chain by more than
    1e-5 * max(1, |g|) on a bounded FRACTION of the voxels (asserted; measured on MI355X: 254 of 2 097 152 voxels =
    0.012 %, worst 9e-5, while the reference gradient is non-zero on 1.8 M voxels) -- everywhere else the two agree.
    (Both bounds take part: the empty voxels hold exactly 1e-5, so the flips are not confined to the solid.)"""
    from oracle.torch_oracle import RenderSphericalCPU
    d = inputs.sphere_depth(noise_seed=2)
    fl, cd = inputs.cam_params(1)
    tdf, _ = oracle.back_projection_forward(d, cd, fl)
    vol = np.clip((1 - 128 * tdf) * 50, 1e-5, 1 - 1e-5).astype(np.float32)        # camera_backprojection_module.py:25-28, :124
    assert ((vol == np.float32(1e-5)) | (vol == np.float32(1 - 1e-5))).all()       # the real volume is two-valued
    g = torch.from_numpy(np.random.default_rng(12).standard_normal((1, 1, 128, 128)).astype(np.float32))
    vc = torch.from_numpy(vol).requires_grad_(True)
    ref = RenderSphericalCPU(oracle)(vc)
    ref.backward(g)
    mod = genre.render_spherical(fused=True).to(dev)
    vt = torch.from_numpy(vol).to(dev).requires_grad_(True)
    out = mod(vt)
    out.backward(g.to(dev))
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= TOL
    gr, gk = vc.grad, vt.grad.cpu()
    err = (gk - gr).abs() / gr.abs().clamp(min=1.0)
    differ = err > TOL
    frac = differ.float().mean().item()
    print("clamp-boundary gradient: %d of %d voxels (%.4f %%) differ from the fp32 reference chain by > 1e-5; "
          "worst %.2e; reference gradient non-zero on %d voxels"
          % (int(differ.sum()), differ.numel(), 100 * frac, err.max().item(), int((gr != 0).sum())))
    assert (gr != 0).sum().item() > 1000000             # a real gradient field, not a trivially equal one
    assert frac <= 1e-3, frac                          # the deviation is a small, bounded set of voxels ...
    assert err.max().item() <= 1e-3, err.max().item()  # ... by a bounded amount
