"""Size-independent properties at bench.py's full size (batch 32 of configs[1]) -- where the oracle is too slow to
be the checker: batch invariance (image i of a batch == the same image run alone), permutation equivariance, and
for the deterministic ops bit-equality between the two; the batch-1 results themselves are pinned against the
oracle by the other test files."""
import numpy as np
import pytest
import torch

import inputs

pytestmark = pytest.mark.gpu

B = 32


def _chain(genre, dev):
    cam = genre.Camera_back_projection_layer().to(dev)
    render = genre.render_spherical().to(dev)

    def run(depth):
        d = depth.clone().requires_grad_(True)
        proj = cam(d)
        out = render(proj, pre_scale=50.0, pad=16)
        g = torch.linspace(-1, 1, out[0].numel(), device=dev).reshape(out.shape[1:]).expand_as(out)
        out.backward(g.contiguous())
        return proj.detach(), out.detach(), d.grad
    return run


def test_hot_path_batch_invariance_at_bench_size(genre, dev):
    depth = torch.from_numpy(inputs.batch_depth(B)).to(dev)
    run = _chain(genre, dev)
    proj, out, gd = run(depth)
    assert out.shape == (B, 1, 160, 160) and torch.isfinite(out).all() and torch.isfinite(gd).all()
    for i in (0, 7, 31):
        p1, o1, g1 = run(depth[i:i + 1])
        # single-hit voxels are bit-exact, multi-hit ones differ by the (undefined) order of the float atomics
        assert (p1[0] - proj[i]).abs().max().item() <= 1e-5
        assert (o1[0] - out[i]).abs().max().item() <= 1e-5
        assert (g1[0] - gd[i]).abs().max().item() <= 1e-5 * max(1.0, gd[i].abs().max().item())
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(dev)
    _, out_p, _ = run(depth[perm])
    assert (out_p - out[perm]).abs().max().item() <= 1e-5


def test_deterministic_ops_are_bitwise_batch_invariant(genre, dev):
    """calc_prob, the fused renderer and Chamfer contain no atomics on their forward path: a batch must
    reproduce the single-image results bit for bit, whatever grouping the kernels choose internally
    (two images per workgroup in the sampler, 8 / 4 / 2 target slices in nnd)"""
    rng = np.random.default_rng(21)
    vox = torch.from_numpy(rng.uniform(0, 0.05, (B, 1, 128, 128, 128)).astype(np.float32)).to(dev)
    render = genre.render_spherical().to(dev)
    with torch.no_grad():
        full = render(vox, pre_scale=20.0, pad=16)
        for i in (0, 1, 30, 31):
            assert torch.equal(render(vox[i:i + 1], pre_scale=20.0, pad=16)[0], full[i])
        odd = render(vox[:3], pre_scale=20.0, pad=16)                # a group with one image only
        assert torch.equal(odd, full[:3])
    p = torch.from_numpy(rng.uniform(1e-5, 1 - 1e-5, (B, 1, 128, 128, 256)).astype(np.float32)).to(dev)
    s = genre.CalcStopProb.apply(p)
    assert torch.equal(genre.CalcStopProb.apply(p[5:6])[0], s[5])
    x1, x2 = inputs.clouds(B, 2048, 2048, seed1=3, seed2=4)
    a, b = torch.from_numpy(x1).to(dev), torch.from_numpy(x2).to(dev)
    d1, d2, i1, i2 = genre.nndistance_w_idx(a, b)
    e1, e2, j1, j2 = genre.nndistance_w_idx(a[9:10], b[9:10])
    assert torch.equal(e1[0], d1[9]) and torch.equal(e2[0], d2[9]) and torch.equal(j1[0], i1[9]) and torch.equal(j2[0], i2[9])
