"""Deterministic weights for network parity tests: every tensor of a state_dict is filled from a generator seeded by
its KEY, so two implementations with the same keys and shapes carry identical weights without a checkpoint file."""
import zlib

import torch


def fill_state(module, seed=0):
    sd = module.state_dict()
    seen = {}
    for k in sorted(sd):
        t = sd[k]
        if not t.dtype.is_floating_point or k.rsplit(".", 1)[-1] in ("grid", "depth_weight"):
            continue                                   # counters; geometry constants of the renderer / back-projection
        if t.data_ptr() in seen:                       # one module registered under two names (Net_inpaint.deconv2)
            continue
        g = torch.Generator().manual_seed(seed * 1000003 + zlib.crc32(k.encode()))
        if k.endswith("running_var"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif k.endswith("running_mean") or k.endswith("bias"):
            v = torch.randn(t.shape, generator=g) * 0.05
        elif t.dim() == 1:                             # normalisation scale
            v = torch.rand(t.shape, generator=g) * 0.5 + 0.75
        else:
            fan = t[0].numel() if t.dim() > 1 else t.numel()
            v = torch.randn(t.shape, generator=g) * (1.5 / max(fan, 1) ** 0.5)
        t.copy_(v)
        seen[t.data_ptr()] = k
    return module


def cases():
    """(name, constructor kwargs key, input shape) of the golden forward passes"""
    return [("uresnet_net", (1, 3, 64, 64)), ("uresnet_inpaint", (1, 1, 96, 96)), ("unet3d", (1, 2, 128, 128, 128)),
            ("image_encoder", (2, 4, 64, 64)), ("voxel_decoder", (1, 200)), ("voxel_generator64", (1, 200, 1, 1, 1)),
            ("voxel_discriminator64", (2, 1, 64, 64, 64))]


def build(ns, name):
    """ns: a namespace with the network classes (the reference's modules or ours)"""
    return {"uresnet_net": lambda: ns.Net([3, 1, 1], ["normal", "depth", "silhou"]),
            "uresnet_inpaint": lambda: ns.Net_inpaint([1], ["spherical"], input_planes=1),
            "unet3d": lambda: ns.Unet_3D(),
            "image_encoder": lambda: ns.ImageEncoder(4),
            "voxel_decoder": lambda: ns.VoxelDecoder(),
            "voxel_generator64": lambda: ns.VoxelGenerator(res=64),
            "voxel_discriminator64": lambda: ns.VoxelDiscriminator(res=64)}[name]()


def make_input(shape, seed=11):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def digest(out):
    """outputs (tensor or dict of tensors) -> {name: (subsample, sum, abs-sum)}; small enough to commit"""
    if not isinstance(out, dict):
        out = {"out": out}
    res = {}
    for k, v in out.items():
        f = v.detach().double().reshape(-1)
        step = max(1, f.numel() // 4096)
        res[k] = (f[::step].float().numpy(), float(f.sum()), float(f.abs().sum()))
    return res


def genre_plausible_geometry(state):
    """key-seeded weights make the GenRe networks emit garbage geometry (depths of ~1e4, a spherical map of ~1e3).
    These edits -- applied to a state_dict by KEY, so to the reference's classes and to ours alike -- keep every network
    active but put the predicted surfaces inside the voxel cube: MarrNet-1 predicts a near-constant relative depth over the
    range [1.9, 2.4], the inpainting network a spherical map of ~0.6 (a sphere of radius 0.4 after `1 - x`)."""
    import torch
    with torch.no_grad():
        state["depth_and_inpaint.net1.decoder_depth.4.3.weight"].mul_(1e-5)
        state["depth_and_inpaint.net1.decoder_minmax.9.weight"].zero_()
        state["depth_and_inpaint.net1.decoder_minmax.9.bias"].copy_(torch.tensor([1.9, 2.4]))
        state["depth_and_inpaint.net2.decoder_spherical.4.1.weight"].mul_(1e-3)
        state["depth_and_inpaint.net2.decoder_spherical.4.1.bias"].fill_(1.0)
        state["depth_and_inpaint.net2.deconv2.weight"].fill_(0.6 / 1024)
        state["depth_and_inpaint.net2.decoder_spherical.4.3.weight"].fill_(0.6 / 1024)    # (the same tensor under its second name)
    return state


def genre_inputs(n=1, seed=9):
    """rgb [n,3,256,256] in [0,1) and a disc silhouette (x100, the data pipeline's scale_25d)"""
    import numpy as np
    import torch
    rng = np.random.default_rng(seed)
    rgb = torch.from_numpy(rng.uniform(0, 1, (n, 3, 256, 256)).astype(np.float32))
    ax = np.linspace(-1, 1, 256)
    sil = ((ax[:, None] ** 2 + ax[None, :] ** 2) < 0.5).astype(np.float32)[None, None].repeat(n, 0) * 100
    return rgb, torch.from_numpy(sil)
