"""SURVEY 8 f-2 / f-3 on the GPU: the GenRe model (three networks around the native ops) against the same modules on
CPU torch with the oracle's ops in between (oracle/torch_oracle.py: GenReGlueCPU), stage by stage -- the geometric
stages contain floor() decisions, so every stage is fed the CPU chain's input and compared on its own -- plus an
end-to-end run, the HIP-graph inference entry, and one joint fine-tuning step with the Chamfer term."""
import numpy as np
import pytest
import torch

import networks_fill as NF

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair(genre, dev):
    from genre_shapehd_amd.models import GenReNet
    torch.manual_seed(0)
    cpu = NF.fill_state(GenReNet(), seed=3).eval()
    gpu = NF.fill_state(GenReNet(), seed=3).eval().to(dev)
    return cpu, gpu


def _inputs(n, seed=5):
    rng = np.random.default_rng(seed)
    rgb = torch.from_numpy(rng.uniform(0, 1, (n, 3, 256, 256)).astype(np.float32))
    ax = np.linspace(-1, 1, 256)
    sil = ((ax[:, None] ** 2 + ax[None, :] ** 2) < 0.5).astype(np.float32)[None, None].repeat(n, 0) * 100
    return rgb, torch.from_numpy(sil)


def _rel(a, b):
    """largest difference relative to the tensor's own scale (seeded random weights give activations of ~1e3)"""
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


def test_genre_forward_stage_by_stage(pair, oracle, dev):
    from genre_shapehd_amd.models import Inputs
    from oracle.torch_oracle import GenReGlueCPU
    cpu, gpu = pair
    rgb, sil = _inputs(1)
    glue = GenReGlueCPU(oracle)
    with torch.no_grad():
        # stage 1: MarrNet-1 (MIOpen vs CPU convolutions)
        o_c = cpu.depth_and_inpaint.net1(Inputs(rgb, sil))
        o_g = gpu.depth_and_inpaint.net1(Inputs(rgb.to(dev), sil.to(dev)))
        for k in ("depth", "normal", "silhou", "depth_minmax"):
            assert _rel(o_g[k].cpu(), o_c[k]) <= 1e-4, k
        # a plausible depth for the geometry (random weights give garbage min/max): 40..60 of 100, range [1.9, 2.5]
        o_c["depth"] = 50 + 10 * torch.tanh(o_c["depth"])
        o_c["depth_minmax"] = torch.tensor([[1.9, 2.5]])
        # stage 2: get_abs_depth -> cam_bp -> render -> pad, from the CPU chain's net1 output
        d_c = glue.get_abs_depth(o_c["depth"], o_c["depth_minmax"], sil)
        proj50_c, sph_c = glue.depth_to_spherical(d_c)
        og = {k: v.to(dev) for k, v in o_c.items()}
        d_g = gpu.depth_and_inpaint.get_abs_depth(og, Inputs(None, sil.to(dev)))
        assert torch.equal(d_g.cpu(), d_c)
        proj_g = gpu.depth_and_inpaint.proj_depth(d_g)
        assert (proj_g.cpu() * 50 - proj50_c).abs().max().item() <= 50 * 128e-5
        sph_g = gpu.depth_and_inpaint.render_spherical(proj_g, pre_scale=50.0, pad=16)
        assert (sph_g.cpu() - sph_c).abs().max().item() <= 1e-5
        # stage 3: the inpainting network on the CPU chain's map
        full_c = cpu.depth_and_inpaint.net2(sph_c)["spherical"]
        full_g = gpu.depth_and_inpaint.net2(sph_c.to(dev))["spherical"]
        assert _rel(full_g.cpu(), full_c) <= 1e-4
        # stage 4: spherical back-projection + refiner input; a map in (0.2, 0.8) so that the points land in the cube
        full_c = 0.5 + 0.3 * torch.tanh(full_c)
        ri_c, cnt_c = glue.refiner_input(full_c, proj50_c)
        from genre_shapehd_amd.callers import RefinerInput
        grid = gpu.grid.expand(1, -1, -1, -1, -1)
        ri_g, cnt_g = RefinerInput.apply(full_c.to(dev), grid, proj50_c.to(dev), 16)
        assert torch.equal(cnt_g.cpu(), cnt_c)
        assert (ri_g.cpu() - ri_c).abs().max().item() <= 1e-4
        # stage 5: the 3-D refiner on the CPU chain's input
        vox_c = cpu.refine_net(ri_c)
        vox_g = gpu.refine_net(ri_c.to(dev))
        assert _rel(vox_g.cpu(), vox_c) <= 1e-4


def test_genre_end_to_end_and_graph_replay(pair, dev):
    """the whole forward on the GPU: finite, right shapes; the HIP-graph entry replays what the eager forward computes.
    Seeded random weights make MarrNet-1 emit depths of ~1e4 with a garbage range, where MIOpen's run-to-run rounding
    (1e-6 relative, measured) flips voxels; the heads are therefore scaled so that the predicted geometry is a plausible
    surface inside the cube (a checkpoint would do the same)."""
    import copy
    from genre_shapehd_amd.models import GenReInference, Inputs
    gpu = copy.deepcopy(pair[1])
    with torch.no_grad():
        n1 = gpu.depth_and_inpaint.net1
        n1.decoder_depth[4][3].weight.mul_(1e-5)                         # depth ~ 0 -> abs depth = max of the range
        n1.decoder_minmax[9].weight.zero_()
        n1.decoder_minmax[9].bias.copy_(torch.tensor([1.9, 2.4]))
        gpu.depth_and_inpaint.net2.deconv2.weight.mul_(1e-6)
    rgb, sil = _inputs(1, seed=9)
    with torch.no_grad():
        out = gpu(Inputs(rgb.to(dev), sil.to(dev)))
    assert out["pred_voxel"].shape == (1, 1, 128, 128, 128) and torch.isfinite(out["pred_voxel"]).all()
    assert out["pred_sph_partial"].shape == (1, 1, 160, 160) and out["proj_depth"].shape == (1, 1, 128, 128, 128)
    assert (out["proj_depth"] != 0).sum().item() > 1000                   # the predicted surface landed in the cube
    inf = GenReInference(gpu, device=dev, graph=True)
    a = inf.predict(rgb, sil)["pred_voxel"].clone()
    b = inf.predict(rgb, sil)["pred_voxel"].clone()                    # second call = pure replay
    # float atomics in cam_bp make multi-hit voxels order-dependent (as in the reference): equal to rounding
    scale = max(1.0, out["pred_voxel"].abs().max().item())
    assert (a - b).abs().max().item() <= 1e-4 * scale
    assert (a - out["pred_voxel"]).abs().max().item() <= 1e-4 * scale
    rgb2, _ = _inputs(1, seed=10)
    sil2 = torch.roll(sil, 40, 3)
    c = inf.predict(rgb2, sil2)["pred_voxel"]
    assert (a - c).abs().max().item() > 1e-3 * scale                      # the static inputs really are refreshed


def test_genre_geometry_follows_the_batch_size_into_the_batch_minor_layout(pair, dev):
    """DepthInpaintNet picks the volume's memory layout from the batch size (models/genre.py): at batch 16 the projected
    volume is image-minor and the renderer's tile kernels run; the geometry outputs equal the NCXYZ path's on the same
    MarrNet-1 output (multi-hit voxels: float-atomic order, as in test_camera_layer_batch_minor_option)"""
    from genre_shapehd_amd.models import Inputs
    gpu = pair[1]
    di = gpu.depth_and_inpaint
    rgb, sil = _inputs(16, seed=21)
    with torch.no_grad():
        o = di.net1(Inputs(rgb.to(dev), sil.to(dev)))
        o["depth"] = 50 + 10 * torch.tanh(o["depth"])
        o["depth_minmax"] = torch.tensor([[1.9, 2.5]], device=dev).repeat(16, 1)
        d = di.get_abs_depth(o, Inputs(None, sil.to(dev)))
        proj = di.proj_depth(d)
        assert proj.stride(0) == 1 and not proj.is_contiguous()         # image-minor at batch 16 ...
        assert di.proj_depth(d[:8]).is_contiguous()                      # ... NCXYZ at the reference's batch sizes
        sph = di.render_spherical(proj, pre_scale=50.0, pad=16)
        std = type(di.proj_depth)()(d)
        assert std.is_contiguous() and (std - proj).abs().max().item() <= 128e-5
        assert (di.render_spherical(std, pre_scale=50.0, pad=16) - sph).abs().max().item() <= 1e-5
        from genre_shapehd_amd.callers import RefinerInput
        full = torch.rand(16, 1, 160, 160, device=dev) * 0.4 + 0.3
        grid = gpu.grid.expand(16, -1, -1, -1, -1)
        ri_bm, _ = RefinerInput.apply(full, grid, proj * 50, 16)        # the refiner input is NCXYZ whatever proj's layout
        ri_std, _ = RefinerInput.apply(full, grid, std * 50, 16)
        assert ri_bm.is_contiguous() and (ri_bm - ri_std).abs().max().item() <= 128e-5


def test_projection_gradients_of_the_joint_step_against_the_cpu_chain(oracle, dev):
    """What reaches MarrNet-1's depth map THROUGH THE GEOMETRY in the joint step (genre_full_model.py:122-131,
    depth_pred_with_sph_inpaint.py:120-129), branch by branch, against the reference's lines on CPU torch + the oracle
    (GenReGlueCPU):

      (i)  through render_spherical <- clamp(proj * 50) <- cam_bp: EXACTLY zero -- every occupied voxel saturates the x50 clamp,
           every empty one sits below its lower bound (:124), on both sides.  The reference's behaviour, pinned here: net2 and
           everything behind it never train MarrNet-1 through this branch.
      (ii) through the refiner's second channel clamp(proj_depth / 50) <- cam_bp <- get_abs_depth (genre_full_model.py:126,
           the un-clamped proj * 50 of :128): the live branch; equal to the CPU chain's gradient to 128 * 1e-5 of its scale
           (shift_tdf multiplies cam_bp's 1e-5 bar by the resolution)."""
    from genre_shapehd_amd.callers import AbsDepth, RefinerInput, GenReGeometry
    from oracle.torch_oracle import GenReGlueCPU
    rng = np.random.default_rng(17)
    n = 2
    _, sil = _inputs(n)
    ax = np.linspace(-1, 1, 256)
    bump = np.exp(-(ax[:, None] ** 2 + ax[None, :] ** 2) * 2)[None, None]
    pred = torch.from_numpy((45 + 12 * bump + rng.uniform(0, 2, (n, 1, 256, 256))).astype(np.float32))    # of scale_25d = 100
    mm = torch.tensor([[1.9, 2.5], [1.85, 2.45]])
    ga = torch.from_numpy(rng.standard_normal((n, 1, 160, 160)).astype(np.float32))
    gb = torch.from_numpy(rng.standard_normal((n, 1, 128, 128, 128)).astype(np.float32))
    full = torch.from_numpy(rng.uniform(0.3, 0.7, (n, 1, 160, 160)).astype(np.float32))                 # net2's map: no gradient here

    # the reference's lines on the host
    glue = GenReGlueCPU(oracle)
    pc = pred.clone().requires_grad_(True)
    d_c = glue.get_abs_depth(pc, mm, sil)
    tdf = glue.hot.cam(d_c, torch.full((n, 1), glue.hot.fl), torch.full((n, 1), glue.hot.cam_dist), 128)
    proj_c = 1 - 128 * tdf
    sph_c = glue.hot.render(torch.clamp(proj_c * 50, 1e-5, 1 - 1e-5))
    (g_render_c,) = torch.autograd.grad((sph_c * ga[:, :, 16:144, 16:144]).sum(), pc, retain_graph=True)
    ch1_c = torch.clamp(proj_c * 50 / 50, 1e-5, 1 - 1e-5)
    (g_ch1_c,) = torch.autograd.grad((ch1_c * gb).sum(), pc)
    assert g_render_c.abs().max().item() == 0.0                         # the reference's own chain: exactly zero
    assert (g_ch1_c != 0).sum().item() > 1000

    # the product
    geo = GenReGeometry().to(dev)
    pg = pred.to(dev).requires_grad_(True)
    d_g = AbsDepth.apply(pg, mm.to(dev), sil.to(dev), 100.0)
    assert torch.equal(d_g.detach().cpu(), d_c.detach())
    proj50_g, sph_g = geo.depth_to_spherical(d_g)
    (g_render_g,) = torch.autograd.grad((sph_g * ga.to(dev)).sum(), pg, retain_graph=True)
    assert torch.count_nonzero(g_render_g).item() == 0                  # (i)
    ri, _ = RefinerInput.apply(full.to(dev), geo.grid.expand(n, -1, -1, -1, -1), proj50_g, 16)
    (g_ch1_g,) = torch.autograd.grad((ri[:, 1:2] * gb.to(dev)).sum(), pg)
    scale = max(1.0, g_ch1_c.abs().max().item())
    err = (g_ch1_g.cpu() - g_ch1_c).abs().max().item()
    print("gradient through refiner channel 1 <- cam_bp: |g| max %.3e, GPU vs CPU chain %.3e" % (scale, err))
    assert err <= 128e-5 * scale                                        # (ii)


def test_joint_finetune_step_reaches_marrnet1_through_the_refiner_channel(dev):
    """config #5 at world_size 1: one Adam step of the joint loss with NO Chamfer term and the 2.5-D losses switched off by
    construction of the check: the voxel loss's gradient arrives in MarrNet-1's depth decoder through Unet_3D's second input
    channel clamp(proj_depth / 50) <- cam_bp <- get_abs_depth ONLY (the branch through net2 <- render_spherical carries an
    identically zero gradient: the test above) -- so the depth head's gradient of the VOXEL loss alone must be
    non-zero and finite"""
    from genre_shapehd_amd import train as T
    from genre_shapehd_amd.models import GenReNet, GenReOptions
    from genre_shapehd_amd.models.genre import genre_loss
    torch.manual_seed(1)
    opt = GenReOptions(joint_train=True)
    net = NF.fill_state(GenReNet(opt), seed=4).to(dev).train()
    with torch.no_grad():                                               # a depth range that puts the surface in the cube
        head = net.depth_and_inpaint.net1.decoder_minmax[9]
        head.weight.zero_()
        head.bias.copy_(torch.tensor([1.9, 2.4]))
        # seeded random weights emit depths of ~1e4 (of scale_25d = 100): scaled so that the predicted surface is a plausible
        # one inside the cube, as in test_genre_end_to_end_and_graph_replay
        net.depth_and_inpaint.net1.decoder_depth[4][3].weight.mul_(1e-5)
    inputs, gt = T.genre_batch(2, dev, seed=11)
    depth_w = net.depth_and_inpaint.net1.decoder_depth[4][3].weight
    pred = net(inputs)
    (g_vox,) = torch.autograd.grad(genre_loss(pred, gt, opt, joint=False), depth_w, allow_unused=True)   # voxel loss ONLY
    assert g_vox is not None and torch.isfinite(g_vox).all() and g_vox.abs().max().item() > 0
    assert (pred["proj_depth"] != 0).sum().item() > 1000                 # ... because the surface did land in the cube
    # and the whole step (all joint losses, no Chamfer) runs and moves the weights
    optim = torch.optim.Adam(net.parameters(), lr=1e-6)
    w0 = depth_w.detach().clone()
    loss = T.genre_train_step(net, optim, inputs, gt, opt, chamfer_weight=0.0)
    assert torch.isfinite(loss)
    assert not torch.equal(w0, depth_w.detach())
    for p in net.refine_net.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()
