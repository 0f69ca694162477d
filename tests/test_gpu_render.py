"""GPU parity: render_spherical / the configs[1] chain against the
CPU-torch restatement (oracle/torch_oracle.py: torch CPU ops with align_corners=True + the C
oracle's calc_prob).  Tolerance 1e-5 absolute on maps in (0,1]."""
import numpy as np
import pytest
import torch

import inputs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def vox_cases(oracle):
    d = inputs.sphere_depth(noise_seed=2)
    fl, cd = inputs.cam_params(1)
    tdf, _ = oracle.back_projection_forward(d, cd, fl)
    proj = 1 - 128 * tdf
    rng = np.random.default_rng(12)
    ax = (np.arange(128) + 0.5) / 128 - 0.5
    r2 = ax[:, None, None] ** 2 + (ax[None, :, None] - 0.1) ** 2 + (ax[None, None, :] + 0.05) ** 2
    return {
        "genre_binary": np.clip(proj * 50, 1e-5, 1 - 1e-5).astype(np.float32),     # depth_pred_with_sph_inpaint.py:124
        "soft": np.clip(proj * 0.7, 1e-5, 1 - 1e-5).astype(np.float32),
        "random": rng.uniform(0.001, 0.05, proj.shape).astype(np.float32),
        "blob": (0.002 + 0.6 * np.exp(-r2 / 0.02)).astype(np.float32)[None, None],
    }


# The clamp inside render_spherical has a discontinuous derivative exactly where GenRe's
# near-binary volumes put most samples (v == 1e-5 up to one ulp), so the reference's own gradient
# flips there with rounding; gradients are compared on fields whose samples stay off the bounds.
GRAD_CASES = ("random", "blob")


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", ["genre_binary", "soft", "random", "blob"])
def test_render_spherical_map_within_1e_5_and_gradient_within_2e_5_of_the_fp32_chain(name, fused, genre, oracle, dev):
    """(the bar is in the name, VERDICT r3: the map holds north_star's 1e-5; the gradient is compared with the reference's OWN
    fp32 op sequence on the CPU, which is itself 8e-5 ... 2.7e-4 off the exact value of its operator -- DESIGN.md 5 -- so the
    bar against it is 2e-5; against the float64 exact-operator yardstick the gradient holds 1e-5: test_gpu_render_genre.py)"""
    from genre_shapehd_amd.toolbox import _fused_render
    if fused and not _fused_render.available():
        pytest.skip("fused render kernel not in this build")
    from oracle.torch_oracle import RenderSphericalCPU
    v = vox_cases(oracle)[name]
    vc = torch.from_numpy(v).requires_grad_(True)
    ref = RenderSphericalCPU(oracle)(vc)
    g = torch.from_numpy(np.random.default_rng(3).standard_normal(ref.shape).astype(np.float32))
    ref.backward(g)
    vt = torch.from_numpy(v).to(dev).requires_grad_(True)
    out = genre.render_spherical(fused=fused).to(dev)(vt)
    assert out.shape == (1, 1, 128, 128)
    assert (out.detach().cpu() - ref.detach()).abs().max().item() <= TOL
    out.backward(g.to(dev))
    assert torch.isfinite(vt.grad).all()
    if name in GRAD_CASES:
        diff = (vt.grad.cpu() - vc.grad).abs() / (1 + vc.grad.abs())
        assert diff.max().item() <= 2e-5, diff.max().item()


def test_chain_config2(genre, oracle, dev):
    """configs[1]: depth -> cam_bp -> x50 clamp -> render_spherical -> sph_pad(16) -> 160x160"""
    from oracle.torch_oracle import HotPathCPU
    d = inputs.batch_depth(2)
    ref = HotPathCPU(oracle).forward(torch.from_numpy(d))
    layer = genre.Camera_back_projection_layer().to(dev)
    render = genre.render_spherical().to(dev)
    proj = layer(torch.from_numpy(d).to(dev))
    out = genre.sph_pad(render(torch.clamp(proj * 50, 1e-5, 1 - 1e-5)), 16)
    assert out.shape == (2, 1, 160, 160)
    assert (out.cpu() - ref).abs().max().item() <= TOL


def test_pre_scale_fusion(genre, oracle, dev):
    """render(x, pre_scale=50) == render(clamp(x*50, 1e-5, 1-1e-5)): forward bit-for-bit (the tile holds
    exactly the materialised values), gradient w.r.t. x within tolerance; also against the CPU chain"""
    from genre_shapehd_amd.toolbox import _fused_render
    if not _fused_render.available():
        pytest.skip("fused render kernel not in this build")
    d = inputs.batch_depth(2)
    fl, cd = inputs.cam_params(2)
    tdf, _ = oracle.back_projection_forward(d, cd, fl)
    rng = np.random.default_rng(5)
    # proj-like volume with values on both sides of the clamp bounds after x50
    x = ((1 - 128 * tdf) * rng.uniform(0.0, 0.03, tdf.shape)).astype(np.float32)
    g = torch.from_numpy(rng.standard_normal((2, 1, 128, 128)).astype(np.float32)).to(dev)
    mod = genre.render_spherical(fused=True).to(dev)
    xa = torch.from_numpy(x).to(dev).requires_grad_(True)
    xb = torch.from_numpy(x).to(dev).requires_grad_(True)
    oa = mod(xa, pre_scale=50.0)
    ob = mod(torch.clamp(xb * 50.0, 1e-5, 1 - 1e-5))
    assert torch.equal(oa, ob)
    oa.backward(g)
    ob.backward(g)
    diff = (xa.grad - xb.grad).abs() / (1 + xb.grad.abs())
    assert diff.max().item() <= 1e-5, diff.max().item()
    from oracle.torch_oracle import RenderSphericalCPU
    ref = RenderSphericalCPU(oracle)(torch.clamp(torch.from_numpy(x) * 50.0, 1e-5, 1 - 1e-5))
    assert (oa.detach().cpu() - ref).abs().max().item() <= TOL


@pytest.mark.parametrize("res,sph_res,z_res", [(24, 16, 32), (40, 24, 64), (16, 8, 12), (33, 20, 100)])
def test_render_odd_geometries_map_within_1e_5_gradient_within_2e_5_of_the_fp32_chain(res, sph_res, z_res, genre, oracle, dev):
    """partial bricks (res not a multiple of 16), small ray fans, short rays: the brick tables are built for
    any geometry; forward and backward against the CPU restatement, fused vs unfused on the GPU"""
    from genre_shapehd_amd.toolbox import _fused_render
    if not _fused_render.available():
        pytest.skip("fused render kernel not in this build")
    from oracle.torch_oracle import RenderSphericalCPU
    rng = np.random.default_rng(res * 1000 + z_res)
    ax = (np.arange(res) + 0.5) / res - 0.5
    r2 = ax[:, None, None] ** 2 + (ax[None, :, None] - 0.07) ** 2 + (ax[None, None, :] + 0.04) ** 2
    v = (0.003 + 0.5 * np.exp(-r2 / 0.03) + rng.uniform(0, 0.02, (res,) * 3)).astype(np.float32)[None, None]
    v = np.concatenate([v, v[:, :, ::-1].copy()], 0)
    vc = torch.from_numpy(v).requires_grad_(True)
    ref = RenderSphericalCPU(oracle, sph_res, z_res)(vc)
    g = torch.from_numpy(rng.standard_normal(ref.shape).astype(np.float32))
    ref.backward(g)
    outs = {}
    for fused in (True, False):
        vt = torch.from_numpy(v).to(dev).requires_grad_(True)
        out = genre.render_spherical(sph_res, z_res, fused=fused).to(dev)(vt)
        assert out.shape == (2, 1, sph_res, sph_res)
        out.backward(g.to(dev))
        outs[fused] = (out.detach().cpu(), vt.grad.cpu())
        assert (outs[fused][0] - ref.detach()).abs().max().item() <= TOL, (fused, res)
        d = (outs[fused][1] - vc.grad).abs() / (1 + vc.grad.abs())
        assert d.max().item() <= 2e-5, (fused, res, d.max().item())


@pytest.mark.parametrize("pad", [1, 16, 64])
def test_fused_pad_matches_sph_pad(pad, genre, dev):
    """render(vox, pad=m) == sph_pad(render(vox), m) (spherical_proj.py:21-28,:126): values bit-equal (the same
    numbers, fanned out), gradient equal up to the order of the <= 2(m+1) additions per map pixel"""
    rng = np.random.default_rng(5)
    vox = torch.from_numpy(rng.uniform(0.0, 0.04, (2, 1, 128, 128, 128)).astype(np.float32)).to(dev)
    mod = genre.render_spherical().to(dev)
    a = vox.clone().requires_grad_(True)
    b = vox.clone().requires_grad_(True)
    out_f = mod(a, pad=pad)
    out_r = genre.sph_pad(mod(b), pad)
    assert out_f.shape == out_r.shape == (2, 1, 128 + 2 * pad, 128 + 2 * pad)
    assert torch.equal(out_f, out_r)
    g = torch.from_numpy(rng.standard_normal(tuple(out_f.shape)).astype(np.float32)).to(dev)
    out_f.backward(g)
    out_r.backward(g)
    scale = b.grad.abs().max().item()
    assert (a.grad - b.grad).abs().max().item() <= 1e-5 * max(1.0, scale)
    # and against the reference's op sequence (unfused module + torch sph_pad)
    ref = genre.render_spherical(fused=False).to(dev)
    assert (genre.sph_pad(ref(vox), pad) - out_f).abs().max().item() <= 1e-5


def _batch_minor(t):
    n, c, x, y, z = t.shape
    out = torch.empty_strided((n, c, x, y, z), (1, n * x * y * z, y * z * n, z * n, n), dtype=t.dtype, device=t.device)
    out.copy_(t)
    return out


@pytest.mark.parametrize("n,pre_scale,pad", [(32, None, 0), (32, 20.0, 16), (19, 20.0, 16), (40, None, 16)])
def test_batch_minor_layout_matches_standard(n, pre_scale, pad, genre, dev):
    """the same logical volume with the image index fastest in memory takes the batch-minor kernels (half-wave =
    32 images of one sample, serial per-lane scans, pull-scatter backward): values agree with the standard path to fp32
    rounding (the scan order differs), gradients to 1e-5 of their maximum"""
    rng = np.random.default_rng(31)
    vox = torch.from_numpy(rng.uniform(0.0, 0.05, (n, 1, 128, 128, 128)).astype(np.float32)).to(dev)
    vox[:, :, 40:60, 50:70, 30:90] = 0.9                              # a solid block: transmittance underflows behind it
    mod = genre.render_spherical().to(dev)
    a = vox.clone().requires_grad_(True)
    b = _batch_minor(vox).requires_grad_(True)
    assert b.stride(0) == 1 and torch.equal(a, b)
    out_a = mod(a, pre_scale=pre_scale, pad=pad)
    out_b = mod(b, pre_scale=pre_scale, pad=pad)
    assert out_a.shape == out_b.shape and out_b.is_contiguous()
    assert (out_a - out_b).abs().max().item() <= 1e-6
    g = torch.from_numpy(rng.standard_normal(tuple(out_a.shape)).astype(np.float32)).to(dev)
    out_a.backward(g)
    out_b.backward(g)
    assert b.grad.stride(0) == 1                                       # the gradient comes back in the same layout
    scale = a.grad.abs().max().item()
    assert (a.grad - b.grad).abs().max().item() <= 1e-5 * max(1.0, scale)


def test_batch_minor_backward_skips_what_the_clamp_blocks(genre, dev):
    """the backward of a group of 32 images none of whose voxels passes clamp(x * pre_scale) -- GenRe's own chain:
    every occupied voxel saturates the x50 clamp, every empty one lies below its lower bound
    (depth_pred_with_sph_inpaint.py:124) -- and of every brick whose masks are all zero writes zeros and does nothing
    else (csrc/sph_render_bm.hip: group word behind the masks, per-brick test in bm_scatter_kernel).  Group 0 = 32
    volumes of the real chain, group 1 = 8 volumes with a gradient everywhere but inside a saturated block: group 0's
    gradient is exactly zero (also under a NaN upstream gradient: the clamp adjoint is a select, not a product), group
    1's equals the standard-layout path's."""
    rng = np.random.default_rng(37)
    d = torch.from_numpy(inputs.batch_depth(32)).to(dev)
    with torch.no_grad():
        proj = genre.Camera_back_projection_layer().to(dev)(d)                       # 1 - 128 tdf: 0 or >= 0.13
    soft = torch.from_numpy(rng.uniform(0.001, 0.019, (8, 1, 128, 128, 128)).astype(np.float32)).to(dev)
    soft[:, :, 40:60, 50:70, 30:90] = 0.9
    vox = torch.cat((proj, soft), 0)
    mod = genre.render_spherical().to(dev)
    a = vox.clone().requires_grad_(True)
    b = _batch_minor(vox).requires_grad_(True)
    out_a, out_b = mod(a, pre_scale=50.0, pad=16), mod(b, pre_scale=50.0, pad=16)
    assert (out_a - out_b).abs().max().item() <= 1e-5                  # (near-binary volumes: the two scan orders, 3e-6)
    g = torch.from_numpy(rng.standard_normal(tuple(out_a.shape)).astype(np.float32)).to(dev)
    g[3] = float("nan")
    out_a.backward(g)
    out_b.backward(g)
    assert torch.count_nonzero(b.grad[:32]).item() == 0
    # (the standard-layout path skips the same work since round 5 -- csrc/sph_render.hip: live words -- and its clamp adjoint is
    # the same select: zero for all 32 images, the one with the NaN upstream gradient included.  PyTorch 0.4.1's clamp
    # backward was a product with the mask and would have returned NaN there: a documented deviation on non-finite input)
    assert torch.count_nonzero(a.grad[:32]).item() == 0
    scale = a.grad[32:].abs().max().item()
    assert scale > 0 and torch.count_nonzero(b.grad[32:, :, 44:56, 56:64, 40:80]).item() == 0
    assert (a.grad[32:] - b.grad[32:]).abs().max().item() <= 1e-5 * max(1.0, scale)


def test_standard_layout_backward_skips_what_the_clamp_blocks(genre, dev):
    """csrc/sph_render.hip, round 5: with pre_scale the forward leaves one word per image and per 16^3 brick ("some voxel
    passes clamp(x * pre_scale)"), and the backward writes zeros for what the clamp blocks instead of computing it.  Five
    images (an odd count: the last sampler group holds one image), dead ones -- GenRe's own chain -- between live ones whose
    saturated blocks cover whole bricks: the gradient with the live words is BIT-IDENTICAL to the gradient computed in full
    (live = None), and the words are what the volume says."""
    from genre_shapehd_amd.toolbox import _fused_render as F
    rng = np.random.default_rng(41)
    d = torch.from_numpy(inputs.batch_depth(2)).to(dev)
    with torch.no_grad():
        proj = genre.Camera_back_projection_layer().to(dev)(d)                       # 1 - 128 tdf: 0 or >= 0.13
    soft = torch.from_numpy(rng.uniform(0.001, 0.019, (3, 1, 128, 128, 128)).astype(np.float32)).to(dev)
    soft[:, :, 32:64, 48:80, 16:96] = 0.9                              # whole 16^3 bricks saturated ...
    soft[1, :, 70:75, 3:9, 100:128] = 0.9                              # ... and a block that covers none
    soft[2, :, 0:96] = 0.0                                             # most of an image below the lower bound
    vox = torch.stack((proj[0], soft[0], soft[1], proj[1], soft[2]), 0).contiguous()
    n = vox.shape[0]
    mod = genre.render_spherical().to(dev)
    lib = F._loader().render_lib
    T = F.tables_for(vox.shape, dev, mod._dirs64, mod.z_res)
    dirs = mod._dirs64.view(torch.float32)
    g = torch.from_numpy(rng.standard_normal((n, 1, 160, 160)).astype(np.float32)).to(dev)
    nb = 8 * 8 * 8
    grads, maps = [], []
    for use_live in (False, True):
        out = torch.empty((n, 1, 160, 160), device=dev)
        v = torch.empty((n * 128 * 128 * 256,), device=dev)
        live = torch.full((n * (1 + nb),), 7, dtype=torch.int32, device=dev) if use_live else None
        lib.render_spherical_forward(vox, dirs, mod.depth_weight, out, v, T["fwd_table"], T["fwd_chunks"], T["kin"], 50.0, live)
        gv = torch.full_like(vox, float("nan"))
        scratch = torch.empty((v.numel() + n,), device=dev)
        lib.render_spherical_backward(vox, dirs, mod.depth_weight, g, gv, scratch, T["bwd_table"], T["bwd_chunks"], v, T["kin"],
                                      50.0, live)
        grads.append(gv)
        maps.append(out)
    assert torch.equal(maps[0], maps[1])
    assert torch.equal(grads[0], grads[1])                              # NaN prefill: every voxel was written, identically
    assert torch.count_nonzero(grads[1][[0, 3]]).item() == 0 and grads[1][[1, 2, 4]].abs().max().item() > 0
    # the words themselves, against the volume
    lv = live.view(n, 1 + nb).cpu()
    t = vox * 50
    passes = ((t >= 1e-5) & (t <= 1 - 1e-5)).view(n, 8, 16, 8, 16, 8, 16).permute(0, 1, 3, 5, 2, 4, 6).reshape(n, nb, -1).any(-1).cpu()
    assert torch.equal(lv[:, 1:] != 0, passes)
    assert torch.equal(lv[:, 0] != 0, passes.any(-1)) and lv[:, 0].tolist() == [0, 1, 1, 0, 1]
    assert (~passes[1]).sum().item() >= 2 * 2 * 5                       # image 1 has dead bricks (its saturated block)
    # ... and through autograd (RenderSphericalFused allocates the words when a gradient is wanted)
    # (round 6: autograd runs the segment forward and the segment form of the dL/dp phase, csrc/sph_render_seg.hip -- fp32 scans
    # per segment chained in fp64 instead of fp64 scans per ray: the same gradient to 1e-5 of its scale, the same zeros)
    x = vox.clone().requires_grad_(True)
    mod(x, pre_scale=50.0, pad=16).backward(g)
    assert (x.grad - grads[0]).abs().max().item() <= 1e-5 * max(1.0, grads[0].abs().max().item())
    assert torch.count_nonzero(x.grad[[0, 3]]).item() == 0                # what the clamp blocks: exact zeros, not small numbers
    dead = ~passes.view(n, 8, 8, 8)[:, :, None, :, None, :, None].expand(n, 8, 16, 8, 16, 8, 16).reshape(n, 1, 128, 128, 128).to(dev)
    assert torch.count_nonzero(x.grad[dead]).item() == 0                  # ... brick by brick


@pytest.mark.parametrize("n", [32, 19, 40])
def test_occupancy_hint_changes_nothing_but_the_traffic(n, genre, dev):
    """Camera_back_projection_layer (image-minor, camera by value) hangs the leader pass's occupancy words on the volume it
    returns; the batch-minor forward then copies precomputed constants for tiles whose bricks hold only the fill value instead
    of reading them (csrc/sph_render_bm.hip).  Same map bit for bit, same gradients (to the 1 ulp of the backward's own atomics) as without the words --
    at pre_scale 50 (GenRe's chain: zero gradient) and 0.9 (occupied voxels pass the clamp: a live gradient) --, words that
    really are sparse, and a hint that is ignored as soon as the volume is written to."""
    from genre_shapehd_amd.toolbox import _fused_render as F
    d = torch.from_numpy(inputs.batch_depth(n, seed=11)).to(dev)
    layer = genre.Camera_back_projection_layer(batch_minor=True).to(dev)
    mod = genre.render_spherical().to(dev)
    rng = np.random.default_rng(5)
    g = torch.from_numpy(rng.standard_normal((n, 1, 160, 160)).astype(np.float32)).to(dev)
    for scale in (50.0, 0.9):
        da, db = d.clone().requires_grad_(True), d.clone().requires_grad_(True)
        pa = layer(da)
        words, fill, ver = pa._genre_brick_hint
        assert pa.stride(0) == 1 and fill == 0.0 and ver == pa._version
        live = (words != 0).float().mean().item()
        assert 0.02 < live < 0.6, live                                   # surfaces: most tiles are empty
        # word (g, b) is set iff brick b or one of its high-side neighbours holds anything but the fill value in some image of g
        occ = (pa.detach() != fill).reshape(n, 32, 4, 16, 8, 16, 8).any(6).any(4).any(2)          # [n, 32, 16, 16] bricks
        for gi in range(words.shape[0]):
            b = torch.nn.functional.pad(occ[gi * 32:(gi + 1) * 32].any(0).float(), (0, 1, 0, 1, 0, 1))
            tile = torch.stack([b[x:x + 32, y:y + 16, z:z + 16] for x in (0, 1) for y in (0, 1) for z in (0, 1)]).amax(0) > 0
            assert torch.equal(tile, words[gi] != 0)
        pb = layer(db)
        del pb._genre_brick_hint                                          # the same volume without the words
        assert F.occupancy_hint(pb, None, scale, None) == (None, None)
        oa, ob = mod(pa, pre_scale=scale, pad=16), mod(pb, pre_scale=scale, pad=16)
        assert torch.equal(oa, ob)
        oa.backward(g)
        ob.backward(g)
        # (the forward's whole saved state -- scratch lines, masks, saved samples -- is bit-identical with and without the words;
        # the renderer's backward itself is reproducible to 1 ulp only: the bricks of the dense centre are split over several
        # workgroups that add onto pre-zeroed voxels with float atomics, csrc/sph_render_bm.hip -- 12 of 67 M voxels differ by
        # 1.2e-7 between two runs on identical inputs)
        scale_g = max(1.0, db.grad.abs().max().item())
        assert (da.grad - db.grad).abs().max().item() <= 1e-6 * scale_g
        assert (da.grad.abs().max().item() > 0) == (scale == 0.9)
    # a volume that was written to after the producer returned it: the words are stale and must not be used
    with torch.no_grad():
        pc = layer(d)
        pc[:, :, 3:9, 100:120, 60:64] += 0.004                            # a dead region becomes non-constant (in place: version moves)
        assert F.occupancy_hint(pc, None, 50.0, None) == (None, None)
        ref = F.empty_batch_minor(pc.shape, torch.float32, dev)
        ref.copy_(pc)
        assert torch.equal(mod(pc, pre_scale=50.0, pad=16), mod(ref, pre_scale=50.0, pad=16))


def test_camera_layer_batch_minor_option(genre, dev):
    """Camera_back_projection_layer(batch_minor=True): same values, image-minor memory; the chain through the
    renderer equals the standard-layout chain"""
    d = torch.from_numpy(inputs.batch_depth(16)).to(dev)
    std, bm = genre.Camera_back_projection_layer().to(dev), genre.Camera_back_projection_layer(batch_minor=True).to(dev)
    ps, pb = std(d), bm(d)
    assert pb.stride(0) == 1 and not pb.is_contiguous() and pb.shape == ps.shape
    assert (ps - pb).abs().max().item() <= 1e-5                        # multi-hit voxels: float-atomic order
    render = genre.render_spherical().to(dev)
    with torch.no_grad():
        assert (render(ps, pre_scale=50.0, pad=16) - render(pb, pre_scale=50.0, pad=16)).abs().max().item() <= 1e-5
    assert bm(d[:4]).is_contiguous()                                    # small batches keep the standard layout
