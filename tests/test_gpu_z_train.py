"""SURVEY 8 f-3 on the GPU: the train steps of BASELINE.json configs[3] (ShapeHD fine-tuning, 3-D WGAN-GP) and
configs[4] (GenRe joint fine-tuning through the differentiable projections + Chamfer) at the REFERENCE's network
widths.  (The file sorts last on purpose: the driver runs `pytest -x`, and nothing here may hide a hot-path result.)

How parity is stated here, and why.  At random initialisation the gradient of these steps is ILL-CONDITIONED: a relative
perturbation of 1e-7 of the weights -- in float64, no rounding involved -- moves single tensors of the ShapeHD step by
3e-2 of their largest entry (condition number 3e5; BatchNorm over a batch of two, a 200-d bottleneck, a mean over 4 M
voxels whose residuals are uncorrelated with the features; measured by `condition_numbers` below).  No float32
evaluation -- the CPU's, MIOpen's, any solver's -- reproduces another to 1e-4 of such a tensor (the CPU's own fp32 step is
1.5e-2 off its float64 step), and a bar fitted to one run of one solver set fails on the next box (round 3).  So the
claim is split into three statements that each hold with an a-priori bar:

  1. every convolution / batch-norm CONFIGURATION the steps run -- forward, data gradient, weight gradient, on random
     data: well conditioned by construction -- agrees with float64 to the textbook bound of an fp32 dot product of its
     reduction length K:  8 sqrt(K) 2^-24 of the largest reference value.  This is where an inaccurate MIOpen solver (or
     a wrong entry in the shipped find-db) shows up, by name.
  2. the step itself -- network wiring, losses, the second-order gradient penalty, optimiser -- is device independent:
     the SAME step in float64 on the GPU equals the float64 step on the CPU to 1e-8 on every tensor.
  3. the fp32 step on the GPU (MIOpen) is a backward-stable evaluation of that function: the loss to 1e-5, the whole
     gradient vector within 1e-4 + 64 kappa 2^-24 (relative L2; kappa = its measured condition number), every tensor
     within 1e-4 + 2048 kappa_t 2^-24 (assert_backward_stable: why two constants).

Reference: models/shapehd.py:82-118, models/marrnet2.py:46-54, models/wgangp.py:77-164,
models/depth_pred_with_sph_inpaint.py:113-129, models/genre_full_model.py:116-143."""
import copy
import math
import zlib

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

U32 = 2.0 ** -24                                   # unit roundoff of float32
CONVS = (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d, nn.ConvTranspose3d)
NORMS = (nn.BatchNorm2d, nn.BatchNorm3d)


def grads(net):
    return {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters() if p.grad is not None}


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp(min=1e-300)).item()


def flat(g, keys):
    return torch.cat([g[k].double().flatten() for k in keys])


def condition_numbers(run64, groups=("",), eps=1e-7, seed=9):
    """kappa_t = (relative L2 change of gradient tensor t) / eps under a relative perturbation eps of every parameter, in
    float64: how strongly ANY evaluation's rounding errors are amplified in that tensor.  `run64(perturb)` -> (scalars,
    gradients) of one step, `perturb` = None or (eps, seed).  -> (scalars of the unperturbed step, its gradients,
    {tensor: kappa}, {key prefix in `groups`: kappa of the gradient vector of the tensors with that prefix})"""
    scalars, base = run64(None)
    _, pert = run64((eps, seed))
    kap = {k: max(rel_l2(pert[k], base[k]) / eps, 1.0) for k in base}
    whole = {}
    for pre in groups:
        keys = sorted(k for k in base if k.startswith(pre))
        whole[pre] = max(rel_l2(flat(pert, keys), flat(base, keys)) / eps, 1.0)
    return scalars, base, kap, whole


def perturb_(net, eps, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1 + eps * torch.randn(p.shape, generator=g, dtype=torch.float64).to(p.dtype))
    return net


def assert_backward_stable(got, ref64, kap, kap_whole, what, c=64.0, c_tensor=2048.0, min_tight=0.0):
    """statement 3 of the module docstring.  Two constants: the whole gradient vector is held to 64 unit roundoffs times its
    condition number; a single tensor to 2048 -- statement 1 allows every convolution 8 sqrt(K) 2^-24 (250 ... 2000 unit
    roundoffs at the reduction lengths of these networks; MIOpen's Winograd and implicit-GEMM solvers do use a few tens of
    them: measured 2.5e-6 on single layers), dozens of layers deep, and a tensor's condition number multiplies THAT, not one
    roundoff (measured on MI355X: ResNet-18's layer2.0.bn1.bias, kappa 5e2, 2.1e-2 off in eval mode).  For the
    ill-conditioned tensors (kappa > 1e4) the tensor bar is vacuous and says so; the vector bar is not."""
    keys = sorted(ref64)
    assert set(got) == set(ref64), (what, set(got) ^ set(ref64))
    top = max(v.abs().max().item() for v in ref64.values())
    worst = (0.0, None, 0.0, 0.0)
    for k in keys:
        if ref64[k].abs().max().item() <= 1e-9 * top:        # analytically zero (a convolution bias in front of a BatchNorm)
            assert got[k].abs().max().item() <= 1e-6 * top, (what, k)
            continue
        e, bar = rel_l2(got[k], ref64[k]), 1e-4 + c_tensor * kap[k] * U32
        if e / bar > worst[0]:
            worst = (e / bar, k, e, bar)
    whole = rel_l2(flat(got, keys), flat(ref64, keys))
    bar_w = 1e-4 + c * kap_whole * U32
    # How much of the per-tensor statement is NOT vacuous: a tensor's bar 1e-4 + 2048 kappa_t 2^-24 says something only while it
    # stays below ~1 (kappa_t <= 1e4: bar <= 1.3); the count and the element share of those tensors are printed and asserted, so
    # that a regression cannot hide behind a condition number -- and the well-conditioned tensors are also held, as
    # ONE vector, to the vector constant (64 unit roundoffs times THEIR condition number, measured like kappa_whole).
    live = [k for k in keys if ref64[k].abs().max().item() > 1e-9 * top]
    tight = [k for k in live if kap[k] <= 1e4]
    share = sum(ref64[k].numel() for k in tight) / max(1, sum(ref64[k].numel() for k in live))
    print("%s: %d tensors; whole gradient off by %.2e (bar %.2e, kappa %.1e); worst tensor %s: %.2e of its bar %.2e "
          "(kappa %.1e); non-vacuous tensor bars (kappa_t <= 1e4): %d of %d tensors = %.1f %% of the gradient's elements"
          % (what, len(keys), whole, bar_w, kap_whole, worst[1], worst[2], worst[3], kap.get(worst[1], 0), len(tight), len(live),
             100 * share))
    assert whole <= bar_w, (what, whole, bar_w)
    assert worst[0] <= 1.0, (what,) + worst
    assert len(tight) >= min_tight * len(live), (what, "only %d of %d tensor bars are non-vacuous" % (len(tight), len(live)))
    if tight:
        k_t = max(kap[k] for k in tight)
        e_t = rel_l2(flat(got, tight), flat(ref64, tight))
        assert e_t <= 1e-4 + c_tensor * k_t * U32, (what, "well-conditioned tensors as one vector", e_t, k_t)


def assert_same_function(got, ref, what, tol=1e-8):
    """statement 2: float64 on the GPU against float64 on the CPU, every tensor, relative to its largest entry"""
    assert set(got) == set(ref), (what, set(got) ^ set(ref))
    top = max(v.abs().max().item() for v in ref.values())
    worst = (0.0, None)
    for k in sorted(ref):
        scale = max(ref[k].abs().max().item(), 1e-6 * top)     # (floor: tensors whose gradient is analytically zero -- a
        e = (got[k].double() - ref[k].double()).abs().max().item() / scale      #  bias in front of a BatchNorm -- hold noise)
        if e > worst[0]:
            worst = (e, k)
    print("%s in float64, GPU vs CPU: %d tensors, worst %.2e (%s)" % ((what, len(ref)) + worst))
    assert worst[0] <= tol, (what,) + worst


# ---- statement 1: every convolution / batch-norm configuration -------------------------------------------------
def _collect(net, run, seen):
    def hook(name):
        def fn(mod, inp, out):
            if isinstance(mod, CONVS):
                base = [c.__name__ for c in CONVS if isinstance(mod, c)][0]        # (thin_conv.py subclasses nn.Conv*3d)
                key = ("conv", base, mod.in_channels, mod.out_channels, tuple(mod.kernel_size),
                       tuple(mod.stride), tuple(mod.padding), tuple(getattr(mod, "output_padding", ())),
                       tuple(inp[0].shape))
            else:
                key = ("norm", type(mod).__name__, mod.num_features, mod.training, tuple(inp[0].shape))
            seen.setdefault(key, name)
        return fn
    handles = [m.register_forward_hook(hook(n)) for n, m in net.named_modules() if isinstance(m, CONVS + NORMS)]
    with torch.no_grad():
        run()
    for h in handles:
        h.remove()


def _check_op(key, dev):
    torch.manual_seed(zlib.crc32(repr(key).encode()) & 0xFFFF)
    shape = key[-1]
    n, spatial = shape[0], shape[2:]
    if key[0] == "conv":
        _, cls, cin, cout, k, s, p, op, _ = key
        kw = dict(kernel_size=k, stride=s, padding=p, bias=False)
        if "Transpose" in cls:
            kw["output_padding"] = op
        m64 = getattr(nn, cls)(cin, cout, **kw).double()
    else:
        _, cls, c, training, _ = key
        m64 = getattr(nn, cls)(c).double().train(training)
        with torch.no_grad():
            m64.weight.uniform_(0.5, 1.5), m64.bias.normal_()
    x64 = torch.randn(shape, dtype=torch.float64)
    y64 = m64(x64.requires_grad_(True))
    g64 = torch.randn_like(y64)
    y64.backward(g64)
    m = copy.deepcopy(m64).float().to(dev)
    m.zero_grad()
    x = x64.detach().float().to(dev).requires_grad_(True)
    y = m(x)
    y.backward(g64.float().to(dev))
    kk = math.prod(k) if key[0] == "conv" else 1
    out_sp = math.prod(y64.shape[2:])
    if key[0] == "conv":
        transposed = "Transpose" in cls
        red = {"y": cin * kk, "dx": cout * kk, "dw": n * (math.prod(spatial) if transposed else out_sp)}
        pairs = {"y": (y, y64), "dx": (x.grad, x64.grad), "dw": (m.weight.grad, m64.weight.grad)}
    else:
        K = n * math.prod(spatial)
        red = {"y": K, "dx": K, "dw": K, "db": K}
        pairs = {"y": (y, y64), "dx": (x.grad, x64.grad), "dw": (m.weight.grad, m64.weight.grad),
                 "db": (m.bias.grad, m64.bias.grad)}
    out = {}
    for name, (a, b) in pairs.items():
        err = ((a.detach().double().cpu() - b.detach()).abs().max() / b.detach().abs().max()).item()
        out[name] = (err, 8.0 * math.sqrt(red[name]) * U32)
    return out


def test_every_convolution_and_batchnorm_of_the_train_steps_is_within_the_fp32_bound_of_float64(genre, dev):
    """statement 1.  Configurations: ShapeHD step (MarrNet-2 + critic), WGAN-GP (generator + critic), GenRe joint step
    (MarrNet-1, the inpainting net, the 3-D refiner), batch 2."""
    from genre_shapehd_amd import train as T
    from genre_shapehd_amd.models.shapehd import ShapeHDNet, WGANGP
    from genre_shapehd_amd.models import GenReNet, GenReOptions
    to = lambda ns: type(ns)(**{k: v.to(dev) for k, v in vars(ns).items()})        # noqa: E731
    seen = {}
    shd = ShapeHDNet().to(dev).train()
    ins, _ = T.sketch_batch(2, "cpu", seed=21)
    _collect(shd, lambda: shd(to(ins)), seen)
    del shd
    gan = WGANGP()
    gan.net_g.to(dev), gan.net_d.to(dev)
    both = nn.ModuleList([gan.net_g, gan.net_d])
    _collect(both, lambda: gan.net_d(gan.net_g(torch.randn(2, 200, 1, 1, 1, device=dev))), seen)
    del gan, both
    g = _plausible_geometry(GenReNet(GenReOptions(joint_train=True))).to(dev).train()
    gin, _ = T.genre_batch(2, "cpu", seed=23)
    _collect(g, lambda: g(to(gin)), seen)
    del g
    torch.cuda.empty_cache()
    bad, worst = [], (0.0, None)
    for key, name in seen.items():
        if key[0] == "norm" and key[3] and key[-1][0] * math.prod(key[-1][2:]) < 16:
            continue        # batch statistics over < 16 values (Unet_3D's 1^3 bottleneck at batch 2: x_hat = +-1, dx == 0 analytically)
        for part, (err, bar) in _check_op(key, dev).items():
            if err / bar > worst[0]:
                worst = (err / bar, (name, part, err, bar))
            if not err <= bar:
                bad.append((name, key, part, err, bar))
    print("%d distinct convolution / batch-norm configurations; closest to its bound: %s" % (len(seen), worst[1]))
    assert len(seen) > 60 and not bad, bad


# ---- ShapeHD -----------------------------------------------------------------------------------------------------
def _shapehd_step(net, inputs, voxel, device, dtype):
    from genre_shapehd_amd import train as T
    net = net.to(device, dtype)
    ins = type(inputs)(**{k: v.to(device, dtype) for k, v in vars(inputs).items()})
    optim = torch.optim.SGD(net.marrnet2.parameters(), lr=0.0)             # lr 0: the step leaves the gradients in place
    loss, _ = T.shapehd_train_step(net, optim, ins, voxel.to(device, dtype), w_gan_loss=0.5)
    return loss.item(), grads(net)


@pytest.mark.parametrize("mode", ["train", "eval"])
def test_shapehd_train_step_on_the_gpu_is_the_cpu_step(genre, dev, mode):
    """configs[3]: MarrNet-2 (ResNet-18 encoder, 200-d code, nf=512 decoder to 128^3) fine-tuned against the frozen
    3-D critic (nf=64); the per-rank shard of batch 64 over 8 GPUs reduced to batch 2 for the CPU side.  `train`:
    BatchNorm on batch statistics (what the step does); `eval`: on its running statistics (a better conditioned instance
    of the same graph -- the whole-gradient bar is then its 1e-4 floor)."""
    from genre_shapehd_amd import train as T
    from genre_shapehd_amd.models.shapehd import ShapeHDNet
    torch.manual_seed(3)
    net0 = ShapeHDNet().train(mode == "train")
    inputs, voxel = T.sketch_batch(2, "cpu", seed=21)

    def run64(perturb):
        net = copy.deepcopy(net0).double()
        if perturb:
            perturb_(net.marrnet2, *perturb)
        return _shapehd_step(net, inputs, voxel, "cpu", torch.float64)
    loss64, g64, kap, whole = condition_numbers(run64)
    assert all(k.startswith("marrnet2.") for k in g64) and len(g64) > 50
    # statement 2: the same function on the device
    loss_d64, g_d64 = _shapehd_step(copy.deepcopy(net0), inputs, voxel, dev, torch.float64)
    assert abs(loss_d64 - loss64) <= 1e-10 * max(1.0, abs(loss64))
    assert_same_function(g_d64, g64, "shapehd_train_step (%s)" % mode)
    # statement 3: the fp32 step (MIOpen)
    loss32, g32 = _shapehd_step(copy.deepcopy(net0), inputs, voxel, dev, torch.float32)
    assert abs(loss32 - loss64) <= 1e-5 * max(1.0, abs(loss64)), (loss32, loss64)
    # (measured on MI355X, round 5: kappa_t <= 1e4 for 10 of 79 tensors with batch statistics, for all 84 with running statistics)
    assert_backward_stable(g32, g64, kap, whole[""], "shapehd_train_step (%s), fp32 on the GPU" % mode,
                           min_tight=0.1 if mode == "train" else 0.9)


# ---- WGAN-GP -------------------------------------------------------------------------------------------------------
def _wgangp_batch(g0, d0, real, device, dtype):
    from genre_shapehd_amd.models.shapehd import WGANGP
    gan = WGANGP(g0.to(device, dtype), d0.to(device, dtype), lr=1e-6, generator=torch.Generator().manual_seed(77))
    if dtype == torch.float64:                                              # the same fp32 random draws, promoted
        gan._random = lambda fn, shape, device, gen=gan.generator: fn(*shape, generator=gen).double().to(device)
    log = gan.train_on_batch(0, real.to(device, dtype))
    out = {"d." + k: v for k, v in grads(gan.net_d).items()}
    out.update({"g." + k: v for k, v in grads(gan.net_g).items()})
    return {k: float(v) for k, v in log.items()}, out


def test_wgangp_train_on_batch_on_the_gpu_is_the_cpu_step(genre, dev):
    """configs[3]'s critic: one WGAN-GP batch (critic step with the second-order gradient penalty, generator step) at
    the reference's widths (nz=200, nf=64, 128^3), latent codes and interpolation weights drawn from one host generator
    on every side"""
    from genre_shapehd_amd import train as T
    from genre_shapehd_amd.networks import VoxelGenerator, VoxelDiscriminator
    torch.manual_seed(5)
    g0, d0 = VoxelGenerator(), VoxelDiscriminator()
    _, real = T.sketch_batch(2, "cpu", seed=22)

    def run64(perturb):
        g, d = copy.deepcopy(g0).double(), copy.deepcopy(d0).double()
        if perturb:
            perturb_(g, perturb[0], perturb[1]), perturb_(d, perturb[0], perturb[1] + 1)
        return _wgangp_batch(g, d, real, "cpu", torch.float64)
    log64, g64, kap, whole = condition_numbers(run64, groups=("d.", "g."))
    log_d64, g_d64 = _wgangp_batch(copy.deepcopy(g0), copy.deepcopy(d0), real, dev, torch.float64)
    for k, v in log64.items():
        assert abs(log_d64[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, log_d64[k], v)
    assert_same_function(g_d64, g64, "WGANGP.train_on_batch")
    log32, g32 = _wgangp_batch(copy.deepcopy(g0), copy.deepcopy(d0), real, dev, torch.float32)
    for k, v in log64.items():
        assert abs(log32[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, log32[k], v)
    # the critic and the generator step are separate backward passes: judged as two gradient vectors
    for part, name in (("d.", "critic step (with the second-order gradient penalty)"), ("g.", "generator step")):
        sub = lambda g: {k: v for k, v in g.items() if k.startswith(part)}      # noqa: E731
        # (round 5: all 6 critic tensors have kappa_t <= 1e4; none of the generator's 16 does -- its statement is the vector bar)
        assert_backward_stable(sub(g32), sub(g64), kap, whole[part], "wgangp " + name + ", fp32 on the GPU",
                               min_tight=0.9 if part == "d." else 0.0)


def _plausible_geometry(net):
    """default-initialised heads predict a garbage depth range; pin the range so that the predicted surface lies inside
    the voxel cube (a checkpoint does the same) -- the depth itself stays the network's output"""
    with torch.no_grad():
        head = net.depth_and_inpaint.net1.decoder_minmax[9]
        head.weight.zero_()
        head.bias.copy_(torch.tensor([1.9, 2.4]))
    return net


def compare(got, want, tol, what):
    """per tensor: max |got - want| <= tol * max(max |want|, 1e-6 * the largest gradient of the whole step)"""
    assert set(got) == set(want), (what, set(got) ^ set(want))
    top = max(v.abs().max().item() for v in want.values())
    assert top > 0, what
    worst = (0.0, None)
    for k in sorted(want):
        scale = max(want[k].abs().max().item(), 1e-6 * top)
        rel = (got[k] - want[k]).abs().max().item() / scale
        if rel > worst[0]:
            worst = (rel, k)
    print("%s: %d tensors, worst relative gradient difference %.2e (%s)" % (what, len(want), worst[0], worst[1]))
    assert worst[0] <= tol, (what,) + worst
    return worst[0]


def test_genre_joint_step_gradient_reaches_marrnet1_like_the_cpu_chain(genre, oracle, dev):
    """configs[4]: the joint loss + Chamfer term.  CPU side: the same networks with the oracle's ops between them
    (oracle/torch_oracle.py: GenReCPU).  Two comparisons:

    (a) the gradient that reaches MarrNet-1's predicted depth map THROUGH THE PROJECTIONS only -- voxel + surface loss
        (Unet_3D <- clamp(proj) <- cam_bp <- get_abs_depth; the branch through render_spherical carries an exactly zero
        gradient on both sides because every occupied voxel saturates the x50 clamp, depth_pred_with_sph_inpaint.py:124)
        plus the Chamfer term -- per pixel.  The geometric ops contain floor() decisions, so the two chains may put a
        few points into neighbouring voxels: the fraction of pixels that differ is asserted, the rest must match;
    (b) the product's train step (all loss terms) against the CPU chain, every trainable tensor of the three modules.

    BatchNorm runs on its running statistics here (eval mode; the gradients still reach every weight).  With batch
    statistics over TWO samples the refiner's bottleneck layers (1^3 ... 4^3 voxels) normalise two numbers to +-1: the
    handful of voxels that the two chains' floor() decisions put elsewhere then changes the refiner's output by 5 % and
    its input gradient by 100 % (measured, first version of this test) -- chaos of the test configuration, not of the
    ops; the train-mode BatchNorm path itself is covered by the ShapeHD and WGAN-GP steps above."""
    import torch.nn.functional as F
    from genre_shapehd_amd import train as T
    from genre_shapehd_amd.callers import AbsDepth
    from genre_shapehd_amd.models import GenReNet, GenReOptions, genre_loss
    from genre_shapehd_amd.models.genre import SCALE_25D
    from genre_shapehd_amd.toolbox.nndistance.functions.nnd import nndistance
    from oracle.torch_oracle import GenReCPU
    torch.manual_seed(7)
    opt = GenReOptions(joint_train=True)
    cpu = _plausible_geometry(GenReNet(opt)).eval()
    gpu = copy.deepcopy(cpu).to(dev)
    inputs, gt = T.genre_batch(2, "cpu", seed=23)
    idx = torch.randint(0, 256 * 256, (2, 2048), generator=torch.Generator().manual_seed(9))
    w_ch = 0.1
    to = lambda ns, d: type(ns)(**{k: v.to(d) for k, v in vars(ns).items()})        # noqa: E731
    in_g, gt_g = to(inputs, dev), to(gt, dev)
    chain = GenReCPU(oracle, cpu.depth_and_inpaint.net1, cpu.depth_and_inpaint.net2, cpu.refine_net)
    stages = ("pred_voxel", "proj_depth", "pred_sph_full", "depth")

    # ---- CPU: one forward; (a) the projection-path loss, (b) the full joint loss
    pred_c = chain.forward(inputs)
    pts = T.depth_to_points(pred_c["abs_depth"], inputs.silhou, idx=idx).contiguous()
    d1, d2 = chain.nnd(pts, gt.cloud.contiguous())
    ch_c = w_ch * (d1.mean() + d2.mean())
    l_c = genre_loss(pred_c, gt, opt, joint=False) + ch_c
    gs_c = torch.autograd.grad(l_c, [pred_c[k] for k in stages], retain_graph=True)
    gsph_c, = torch.autograd.grad(F.mse_loss(pred_c["pred_sph_full"], gt.spherical_object), pred_c["depth"], retain_graph=True)
    loss_c = genre_loss(pred_c, gt, opt, joint=True) + ch_c
    loss_c.backward()
    g_cpu = grads(cpu)

    # ---- GPU (a)
    pred_g = gpu(in_g)
    depth = AbsDepth.apply(pred_g["depth"], pred_g["depth_minmax"], in_g.silhou, SCALE_25D)
    e1, e2 = nndistance(T.depth_to_points(depth, in_g.silhou, idx=idx.to(dev)).contiguous(), gt_g.cloud.contiguous())
    l_g = genre_loss(pred_g, gt_g, opt, joint=False) + w_ch * (e1.mean() + e2.mean())
    gs_g = torch.autograd.grad(l_g, [pred_g[k] for k in stages], retain_graph=True)
    gsph_g, = torch.autograd.grad(F.mse_loss(pred_g["pred_sph_full"], gt_g.spherical_object), pred_g["depth"])
    for k, a, b in zip(stages, gs_g, gs_c):                          # where along the chain the two sides part, if they do
        nb = max(b.double().norm().item(), 1e-300)
        print("   d loss / d %-14s |cpu| %.3e  |gpu - cpu| / |cpu| %.2e  (forward values: %.2e)" % (
            k, nb, (a.cpu().double() - b.double()).norm().item() / nb,
            ((pred_g[k].detach().cpu().double() - pred_c[k].detach().double()).norm() / pred_c[k].detach().double().norm()).item()))
    assert abs(l_g.item() - l_c.item()) <= 1e-4 * max(1.0, abs(l_c.item())), (l_g.item(), l_c.item())
    gd_g, gd_c = gs_g[-1], gs_c[-1]
    top = gd_c.abs().max().item()
    live = (gd_c != 0).sum().item()
    assert top > 0 and live > 10000, (top, live)                     # the gradient really arrives through cam_bp
    diff = (gd_g.cpu() - gd_c).abs()
    off = (diff > 1e-3 * top).sum().item()
    rel_l2 = (diff.double().norm() / gd_c.double().norm()).item()
    print("projection-path gradient at the depth map: max %.3e, %d live pixels, %d differ by > 1e-3 of max (%.3f %%), "
          "relative L2 %.2e" % (top, live, off, 100.0 * off / live, rel_l2))
    assert off <= 0.005 * live, (off, live)
    assert rel_l2 <= 0.05, rel_l2
    # render_spherical's branch: exactly zero on the CPU chain (saturated clamp), and on the GPU
    assert gsph_c.abs().max().item() == 0 and gsph_g.abs().max().item() == 0
    del pred_g, l_g, gs_g

    # ---- GPU (b): the product's train step (lr 0 keeps the gradients in place)
    optim = torch.optim.SGD(gpu.parameters(), lr=0.0)
    loss_g = T.genre_train_step(gpu, optim, in_g, gt_g, opt, chamfer_weight=w_ch, chamfer_idx=idx.to(dev))
    g_gpu = grads(gpu)
    assert abs(loss_g.item() - loss_c.item()) <= 1e-4 * max(1.0, abs(loss_c.item())), (loss_g.item(), loss_c.item())
    key = "depth_and_inpaint.net1.decoder_depth.4.3.weight"
    a, b = g_gpu[key].double(), g_cpu[key].double()
    rel_l2 = ((a - b).norm() / b.norm()).item()
    print("depth head gradient: |g| = %.3e, relative L2 difference %.2e" % (b.norm().item(), rel_l2))
    assert b.abs().max().item() > 0 and rel_l2 <= 1e-3
    # every tensor: the handful of points that the two chains' floor() decisions put into neighbouring voxels (d loss / d
    # pred_sph_full differs by 4e-3 above) reaches the refiner's weight gradients at that level -- measured worst: 5.1e-3
    # of max |g| (a transposed-convolution bias of the refiner) -- while MarrNet-1's depth head agrees to 2e-7
    compare(g_gpu, g_cpu, 2e-2, "genre joint step")
    # MarrNet-1 as ONE gradient vector (its 2-D heads are plain MIOpen Conv2d against CPU fp32: single small tensors differ by
    # 1-2e-3 of their maximum from run to run -- solver choice, atomics order --, the vector does not)
    keys = sorted(k for k in g_cpu if ".net1." in k)
    va, vb = flat(g_gpu, keys), flat(g_cpu, keys)
    vec = ((va - vb).norm() / vb.norm()).item()
    print("genre joint step, MarrNet-1 as one vector: %d tensors, relative L2 %.2e" % (len(keys), vec))
    assert vec <= 1e-3, vec
