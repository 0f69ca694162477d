"""GPU parity of the segment forward (csrc/sph_render_seg.hip; SURVEY 8 f-1 on the layout the reference's callers run): against
the reference's op sequence on the host (oracle/torch_oracle.py: CPU torch with align_corners=True + the C oracle's calc_prob), against the
per-sample forward it replaces (csrc/sph_render.hip: fp64 scan over the raw sample values), with and without the camera
forward's occupancy words, and the backward that now recomputes what the forward no longer saves."""
import numpy as np
import pytest
import torch

import inputs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _old_forward(F, mod, vox, pre_scale=0.0, pad=0, live=None):
    """the per-sample forward of csrc/sph_render.hip (v[ray, k] scratch + fp64 scan per ray)"""
    lib = F._loader().render_lib
    n, c = vox.shape[:2]
    res, zr = mod.sph_res, mod.z_res
    T = F.tables_for(vox.shape, vox.device, mod._dirs64, zr)
    out = torch.empty((n, c, res + 2 * pad, res + 2 * pad), device=vox.device)
    v = torch.empty((n * c * res * res * zr,), device=vox.device)
    lib.render_spherical_forward(vox, mod._dirs64.view(torch.float32), mod.depth_weight, out, v, T["fwd_table"], T["fwd_chunks"],
                                 T["kin"], pre_scale, live)
    return out


@pytest.mark.parametrize("shape,sph,zr", [((1, 1, 128, 128, 128), 128, 256), ((3, 1, 128, 128, 128), 128, 256),
                                          ((5, 2, 33, 33, 33), 24, 64), ((2, 1, 40, 24, 56), 16, 32), ((7, 1, 16, 16, 16), 8, 12)])
@pytest.mark.parametrize("pre_scale,pad", [(0.0, 0), (50.0, 16), (0.9, 3)])
def test_segment_forward_equals_the_per_sample_forward(shape, sph, zr, pre_scale, pad, genre, dev):
    """same operator, two formulations: fp32 (P, S) per segment of <= 16 samples chained in fp64 against an fp64 scan over every
    sample -- 5e-6 on maps in (0, 1] (measured 3.2e-6; north_star's bar against the reference chain is 1e-5, next test); odd
    geometries, several channels, odd image counts (the last workgroup holds one image)"""
    from genre_shapehd_amd.toolbox import _fused_render as F
    if 2 * pad > sph:
        pad = sph // 4
    rng = np.random.default_rng(sum(shape) + sph)
    vox = rng.uniform(0.0, 0.03 if pre_scale == 50.0 else 0.6, shape).astype(np.float32)
    vox[:, :, : shape[2] // 3] = 0.0                                     # an empty slab and a saturated block
    vox[:, :, shape[2] // 2:, shape[3] // 2:, : shape[4] // 4] = 1.0
    vt = torch.from_numpy(vox).to(dev)
    mod = genre.render_spherical(sph_res=sph, z_res=zr, fused=True).to(dev)
    with torch.no_grad():
        new = mod(vt, pre_scale=pre_scale or None, pad=pad)
    old = _old_forward(F, mod, vt, pre_scale, pad)
    assert new.shape == old.shape and torch.isfinite(new).all()
    assert (new - old).abs().max().item() <= 5e-6, (new - old).abs().max().item()


def test_segment_forward_against_the_reference_chain_on_the_host(genre, oracle, dev):
    """configs[1]'s renderer on its stated input, against toolbox/spherical_proj.py:62-72 on CPU torch + the C oracle"""
    from oracle.torch_oracle import RenderSphericalCPU
    d = inputs.sphere_depth(noise_seed=2)
    fl, cd = inputs.cam_params(1)
    tdf, _ = oracle.back_projection_forward(d, cd, fl)
    for vol in (np.clip((1 - 128 * tdf) * 50, 1e-5, 1 - 1e-5).astype(np.float32),
                np.random.default_rng(5).uniform(0.001, 0.05, tdf.shape).astype(np.float32)):
        ref = RenderSphericalCPU(oracle)(torch.from_numpy(vol))
        with torch.no_grad():
            out = genre.render_spherical(fused=True).to(dev)(torch.from_numpy(vol).to(dev))
        assert (out.cpu() - ref).abs().max().item() <= TOL


@pytest.mark.parametrize("n", [1, 2, 5])
def test_camera_cell_words_and_the_hinted_forward(n, genre, dev):
    """Camera_back_projection_layer (dense volume, camera by value: the brick kernel) hangs one word per image and 8x8x32-voxel
    cell on the volume it returns -- set iff a point landed in the cell -- and the segment forward copies the geometry's constants
    for tiles none of whose cells is set instead of reading them: the same map BIT FOR BIT, words that are what the volume says,
    a hint that dies with any write to the volume (ATen in place: version counter; raw C ABI: _loader drops it)."""
    from genre_shapehd_amd.toolbox import _fused_render as F
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    d = torch.from_numpy(inputs.batch_depth(n, seed=3)).to(dev)
    layer = genre.Camera_back_projection_layer().to(dev)
    mod = genre.render_spherical().to(dev)
    with torch.no_grad():
        pa = layer(d)
        words, fill, ver, cell = pa._genre_cell_hint
        assert pa.is_contiguous() and fill == 0.0 and ver == pa._version and cell == 80832
        occ = (pa != fill).reshape(n, 16, 8, 16, 8, 4, 32).any(6).any(4).any(2)
        assert torch.equal(occ, words.view(n, 16, 16, 4) != 0)
        assert 0.01 < occ.float().mean().item() < 0.5
        pb = pa.clone()                                                   # the same values without the words
        for scale, pad in ((50.0, 16), (None, 0), (0.9, 4)):
            assert torch.equal(mod(pa, pre_scale=scale, pad=pad), mod(pb, pre_scale=scale, pad=pad))
        # written to after the producer returned it -- through ATen (version counter) ...
        pc = layer(d)
        pc[:, :, 3:9, 100:120, 60:64] += 0.004
        t = F.seg_tables_for(pc.shape, dev, mod._dirs64, mod.depth_weight)
        assert F.occupancy_hint_std(pc, t, mod._dirs64, mod.depth_weight, 50.0, None) == (None, None, 0)
        assert torch.equal(mod(pc, pre_scale=50.0, pad=16), mod(pc.clone(), pre_scale=50.0, pad=16))
        # ... and through the raw C ABI, as the reference's caller-allocates convention invites (cam_back_projection.py:22-25):
        # the hinted volume is re-used as the OUTPUT of another camera forward
        pd = layer(d)
        assert getattr(pd, "_genre_cell_hint", None) is not None
        d2 = torch.from_numpy(inputs.batch_depth(n, seed=99)).to(dev)
        cam_bp_lib.back_projection_forward_const(d2, 2.2, 418.3, pd, torch.empty_like(pd), shifted=True)
        assert getattr(pd, "_genre_cell_hint", None) is None
        assert torch.equal(mod(pd, pre_scale=50.0, pad=16), mod(pd.clone(), pre_scale=50.0, pad=16))
        assert not torch.equal(pd, pa)


def test_hint_is_not_used_where_a_gradient_could_come_back_through_a_dead_tile(genre, dev):
    """a dead tile's clamp pass words / saved state are only right when the fill value is blocked by the
    pre_scale clamp.  Without pre_scale (or with a fill value that passes it) and a gradient wanted, the hint is ignored --
    both layouts -- and the gradient equals the un-hinted one."""
    from genre_shapehd_amd.toolbox import _fused_render as F
    assert F._hint_usable(0.0, 50.0, True) and F._hint_usable(0.0, 0.0, False)
    assert not F._hint_usable(0.0, 0.0, True) and not F._hint_usable(1.0 / 128, 50.0, True)
    mod = genre.render_spherical().to(dev)
    g = torch.from_numpy(np.random.default_rng(2).standard_normal((16, 1, 128, 128)).astype(np.float32)).to(dev)
    d = torch.from_numpy(inputs.batch_depth(16, seed=4)).to(dev)
    for bm in (False, True):
        layer = genre.Camera_back_projection_layer(batch_minor=bm).to(dev)
        da, db = d.clone().requires_grad_(True), d.clone().requires_grad_(True)
        pa, pb = layer(da), layer(db)
        name = "_genre_brick_hint" if bm else "_genre_cell_hint"
        assert getattr(pa, name, None) is not None
        delattr(pb, name)
        oa, ob = mod(pa), mod(pb)                                          # no pre_scale: raw volume, gradient everywhere
        assert torch.equal(oa, ob)
        oa.backward(g)
        ob.backward(g)
        assert torch.isfinite(da.grad).all() and da.grad.abs().max().item() > 0
        assert (da.grad - db.grad).abs().max().item() <= 1e-6 * max(1.0, db.grad.abs().max().item())


def test_layer_under_inference_mode_and_single_image_batch_minor(genre, dev):
    """tensors of torch.inference_mode() have no version counter -- the layer returns them without a hint
    instead of raising; ShiftedCameraBackProjection(batch_minor=True) on ONE image has float4-aligned rows -- the library takes
    the brick kernel there and Python no longer insists on the leader pass's sparse cnt"""
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp.functions.cam_back_projection import ShiftedCameraBackProjection
    mod = genre.render_spherical().to(dev)
    for n, bm in ((16, True), (2, False)):
        d = torch.from_numpy(inputs.batch_depth(n, seed=6)).to(dev)
        layer = genre.Camera_back_projection_layer(batch_minor=bm).to(dev)
        with torch.no_grad():
            ref = mod(layer(d), pre_scale=50.0, pad=16)
        with torch.inference_mode():
            p = layer(d)
            assert getattr(p, "_genre_brick_hint", None) is None and getattr(p, "_genre_cell_hint", None) is None
            assert torch.equal(mod(p, pre_scale=50.0, pad=16), ref)
    d1 = torch.from_numpy(inputs.batch_depth(1, seed=6)).to(dev)
    fl, cd = torch.full((1, 1), 418.3, device=dev), torch.full((1, 1), 2.2, device=dev)
    with torch.no_grad():
        a = ShiftedCameraBackProjection.apply(d1, fl, cd, 128, True, (418.3, 2.2))
        b = ShiftedCameraBackProjection.apply(d1, fl, cd, 128, False, (418.3, 2.2))
    assert torch.equal(a, b)


def test_segment_form_backward_equals_the_per_sample_backward(genre, dev):
    """the standard-layout backward in segment form (per-ray chains over the forward's (P, S) pairs, dL/dp per segment from the
    sample values the forward saved where a gradient can come back; csrc/sph_render_seg.hip) against round 5's (every sample
    value saved, wave-per-ray fp64 scans; csrc/sph_render.hip): the same gradient to 1e-5 of its scale, zeros where the clamp
    blocks an image or a brick, with and without pre_scale, an odd image count with a dead image between live ones"""
    from genre_shapehd_amd.toolbox import _fused_render as F
    rng = np.random.default_rng(8)
    vox = torch.from_numpy(rng.uniform(0.001, 0.019, (3, 1, 128, 128, 128)).astype(np.float32)).to(dev)
    vox[1] = 0.0                                                         # a dead image between two live ones
    vox[2, :, 32:64, 48:80, 16:96] = 0.9                                 # whole bricks saturated
    mod = genre.render_spherical().to(dev)
    lib = F._loader().render_lib
    T = F.tables_for(vox.shape, dev, mod._dirs64, mod.z_res)
    dirs = mod._dirs64.view(torch.float32)
    g = torch.from_numpy(rng.standard_normal((3, 1, 160, 160)).astype(np.float32)).to(dev)
    for scale in (50.0, 0.0):
        live = torch.empty((3 * 513,), dtype=torch.int32, device=dev) if scale else None
        out = torch.empty((3, 1, 160, 160), device=dev)
        v = torch.empty((3 * 128 * 128 * 256,), device=dev)
        lib.render_spherical_forward(vox, dirs, mod.depth_weight, out, v, T["fwd_table"], T["fwd_chunks"], T["kin"], scale, live)
        gv = torch.full_like(vox, float("nan"))
        scratch = torch.empty((v.numel() + 3,), device=dev)
        lib.render_spherical_backward(vox, dirs, mod.depth_weight, g, gv, scratch, T["bwd_table"], T["bwd_chunks"], v, T["kin"],
                                      scale, live)
        x = vox.clone().requires_grad_(True)
        y = mod(x, pre_scale=scale or None, pad=16)
        assert (y - out).abs().max().item() <= 5e-6
        y.backward(g)
        assert torch.isfinite(x.grad).all()
        s = gv.abs().max().item()
        assert s > 0 and (x.grad - gv).abs().max().item() <= 1e-5 * max(1.0, s), ((x.grad - gv).abs().max().item(), s)
        assert torch.count_nonzero(x.grad[1]).item() == 0 and x.grad[0].abs().max().item() > 0
        if scale:
            assert torch.count_nonzero(x.grad[2, :, 32:64, 48:80, 16:96]).item() == 0      # saturated bricks: blocked


def test_zero_gradient_words_reach_the_camera_backward(genre, dev):
    """the renderer's backward knows from the forward's clamp words which images' gradient is identically zero and hangs that on
    the gradient tensor it returns (toolbox/_fused_render.py: attach_zero_hint); the camera layer's backward then writes zeros for
    those images without reading anything (genre_back_projection_backward_hinted).  Same gradients as without the words -- on
    GenRe's own chain (pre_scale 50: every image blocked), on a chain with a live gradient (0.9: words say "alive", nothing is
    skipped), both layouts --, the hinted entry IS what runs, and words alone decide: a raw call with mixed words"""
    from genre_shapehd_amd.toolbox import _fused_render as F
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    calls = []
    real = cam_bp_lib.back_projection_backward_hinted
    real_attach = F.attach_zero_hint
    cam_bp_lib.back_projection_backward_hinted = staticmethod(lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    rng = np.random.default_rng(12)
    try:
        for n, batch_minor in ((3, False), (32, True)):
            d0 = torch.from_numpy(inputs.batch_depth(n, seed=5)).to(dev)
            g = torch.from_numpy(rng.standard_normal((n, 1, 160, 160)).astype(np.float32)).to(dev)
            layer = genre.Camera_back_projection_layer(batch_minor=batch_minor).to(dev)
            mod = genre.render_spherical().to(dev)
            for scale in (50.0, 0.9):
                grads = []
                for hinted in (True, False):
                    F.attach_zero_hint = real_attach if hinted else (lambda grad, *a: grad)
                    d = d0.clone().requires_grad_(True)
                    before = len(calls)
                    mod(layer(d), pre_scale=scale, pad=16).backward(g)
                    assert (len(calls) > before) == hinted
                    grads.append(d.grad.clone())
                # (a live gradient: the renderer's backward sums a brick's low faces with atomics -- two runs differ in last bits)
                top = grads[1].abs().max().item()
                assert (grads[0] - grads[1]).abs().max().item() <= 1e-5 * top
                assert (top == 0) == (scale == 50.0) and (grads[0].abs().max().item() == 0) == (scale == 50.0)
    finally:
        cam_bp_lib.back_projection_backward_hinted = staticmethod(real)
        F.attach_zero_hint = real_attach
    # the entry itself: words decide, image by image (group 1) and in groups of two
    n = 4
    d = torch.from_numpy(inputs.batch_depth(n, seed=6)).to(dev)
    fl = torch.full((n, 1), 418.3, device=dev); cd = torch.full((n, 1), 2.2, device=dev)
    vol, cnt = torch.empty((n, 1, 128, 128, 128), device=dev), torch.empty((n, 1, 128, 128, 128), device=dev)
    cam_bp_lib.back_projection_forward_shifted(d, cd, fl, vol, cnt)
    gin = torch.from_numpy(rng.standard_normal((n, 1, 128, 128, 128)).astype(np.float32)).to(dev)
    ref = [torch.empty_like(d), torch.empty((n, 1), device=dev), torch.empty((n, 1), device=dev)]
    cam_bp_lib.back_projection_backward_shifted(d, fl, cd, cnt, gin, *ref)
    words = torch.tensor([7, 1, 7, 0, 7, 1, 7, 1], dtype=torch.int32, device=dev)       # stride 2, offset 1: images 0, 2, 3 alive, 1 dead
    got = [torch.full_like(d, float("nan")), torch.empty((n, 1), device=dev), torch.empty((n, 1), device=dev)]
    cam_bp_lib.back_projection_backward_hinted(d, fl, cd, cnt, gin, *got, words, 2, 1, 1, shifted=True)
    for i in range(n):
        if i == 1:
            assert torch.count_nonzero(got[0][i]).item() == 0 and got[1][i].item() == 0 and got[2][i].item() == 0
        else:
            assert torch.equal(got[0][i], ref[0][i])
            for k in (1, 2):                                                  # (block partials meet in fp32 atomics: order is not fixed)
                assert abs(got[k][i].item() - ref[k][i].item()) <= 1e-5 * abs(ref[k][i].item())
    words2 = torch.tensor([0, 5], dtype=torch.int32, device=dev)                          # groups of two images: 0, 1 dead
    cam_bp_lib.back_projection_backward_hinted(d, fl, cd, cnt, gin, *got, words2, 1, 0, 2, shifted=True)
    assert torch.count_nonzero(got[0][:2]).item() == 0 and torch.equal(got[0][2:], ref[0][2:])
    with pytest.raises(RuntimeError, match="zero_words"):
        cam_bp_lib.back_projection_backward_hinted(d, fl, cd, cnt, gin, *got, words2, 1, 0, 1, shifted=True)   # four images, two words


@pytest.mark.parametrize("pre_scale", [None, 3.0])
def test_segment_renderer_on_strided_volumes_and_two_channels(pre_scale, genre, dev):
    """the standard-layout segment forward / backward read the volume through its strides (rows that are no float4, permuted
    axes, NC = 2): same map and gradient as on a contiguous copy"""
    rng = np.random.default_rng(21)
    big = torch.from_numpy(rng.uniform(0.0, 0.4, (2, 2, 40, 44, 49)).astype(np.float32)).to(dev)
    mod = genre.render_spherical(sph_res=24, z_res=64).to(dev)
    g = torch.from_numpy(rng.standard_normal((2, 2, 24 + 8, 24 + 8)).astype(np.float32)).to(dev)
    for view in (big[:, :, :, :40, 3:43], big[:, :, :, :40, :40].transpose(2, 3)):
        assert not view.is_contiguous()
        res = []
        for v in (view, view.contiguous()):
            x = v.detach().clone(memory_format=torch.preserve_format) if v.is_contiguous() else v.detach()
            x = x.requires_grad_(True)
            y = mod(x, pre_scale=pre_scale, pad=4)
            y.backward(g)
            res.append((y.detach(), x.grad.detach()))
        assert torch.equal(res[0][0], res[1][0])
        top = res[1][1].abs().max().item()
        assert top > 0 and (res[0][1] - res[1][1]).abs().max().item() <= 2e-6 * top


@pytest.mark.parametrize("n,batch_minor", [(2, False), (32, True)])
def test_provably_blocked_chain_launches_no_renderer_backward(n, batch_minor, genre, dev, monkeypatch):
    """GenRe's own chain (layer -> clamp(x50) folded into render_spherical): the layer hangs its value range on the volume, the
    renderer sees on the host that the clamp blocks every voxel, saves nothing and its backward launches nothing -- the gradient
    is a stride-0 view of one zero that every reader sees as zeros and the camera layer's backward does not read at all.  Same
    map, same (zero) gradient as with GENRE_LAZY_ZERO_GRAD=0, where the kernels find the zeros themselves; a pre_scale that lets
    occupied voxels through takes the usual path."""
    from genre_shapehd_amd.toolbox import _fused_render as F
    lib = F._loader().render_lib
    calls = []
    for name in ("render_seg_backward", "render_bm_backward"):
        real = getattr(lib, name)
        monkeypatch.setattr(lib, name, staticmethod(lambda *a, _r=real, _n=name, **k: (calls.append(_n), _r(*a, **k))[1]))
    d0 = torch.from_numpy(inputs.batch_depth(n, seed=9)).to(dev)
    g = torch.randn((n, 1, 160, 160), device=dev)
    layer = genre.Camera_back_projection_layer(batch_minor=batch_minor).to(dev)
    mod = genre.render_spherical().to(dev)
    res = {}
    for lazy in ("1", "0"):
        monkeypatch.setenv("GENRE_LAZY_ZERO_GRAD", lazy)
        d = d0.clone().requires_grad_(True)
        proj = layer(d)
        proj.retain_grad()
        before = len(calls)
        out = mod(proj, pre_scale=50.0, pad=16)
        out.backward(g)
        assert (len(calls) == before) == (lazy == "1")
        assert proj.grad.shape == proj.shape and torch.count_nonzero(proj.grad).item() == 0
        assert torch.count_nonzero(d.grad).item() == 0
        res[lazy] = out.detach().clone()
    # (the standard layout's forward rounds a segment's transmittance product once when it saves for a backward, per factor when
    # it does not: the two maps agree to a few 1e-7)
    assert (res["1"] - res["0"]).abs().max().item() <= 1e-6
    monkeypatch.setenv("GENRE_LAZY_ZERO_GRAD", "1")
    d = d0.clone().requires_grad_(True)
    before = len(calls)
    mod(layer(d), pre_scale=0.9, pad=16).backward(g)
    assert len(calls) == before + 1 and d.grad.abs().max().item() > 0
