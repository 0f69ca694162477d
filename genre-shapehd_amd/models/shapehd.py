"""ShapeHD on PyTorch-ROCm: MarrNet-2 (2.5-D sketches -> 128^3 voxels), its fine-tuning against a frozen 3-D WGAN-GP
critic as a naturalness loss (models/shapehd.py:82-118, marrnet2.py:88-111), and the WGAN-GP training step itself
(models/wgangp.py:77-164) with the gradient penalty through a second-order autograd.grad.  No hot-path kernels here
 -- these are the stock Conv2d / Conv3d / ConvTranspose3d networks the DDP launcher (train.py) trains."""
import torch
from torch import nn
import torch.nn.functional as F

from ..networks import ImageEncoder, VoxelDecoder, VoxelGenerator, VoxelDiscriminator


class MarrNet2Net(nn.Module):
    def __init__(self, in_planes=4, encode_dims=200, silhou_thres=0, nf=512):
        super().__init__()
        self.encoder = ImageEncoder(in_planes, encode_dims=encode_dims)
        self.decoder = VoxelDecoder(n_dims=encode_dims, nf=nf)
        self.silhou_thres = silhou_thres

    def forward(self, input_struct):
        bg = input_struct.silhou <= self.silhou_thres                 # background pixels of both sketches -> 0
        depth = input_struct.depth.masked_fill(bg, 0)
        normal = input_struct.normal.masked_fill(bg.expand(-1, 3, -1, -1), 0)
        return self.decoder(self.encoder(torch.cat((depth, normal), 1)))


class ShapeHDNet(nn.Module):
    """fine-tuned MarrNet-2 + its frozen copy + the frozen critic"""

    def __init__(self, marrnet2_path=None, gan_path=None, d_nf=64, **net_kw):
        super().__init__()
        self.marrnet2 = MarrNet2Net(4, **net_kw)
        self.marrnet2_noft = MarrNet2Net(4, **net_kw)
        if marrnet2_path:
            sd = torch.load(marrnet2_path, map_location="cpu")["nets"][0]
            self.marrnet2.load_state_dict(sd)
            self.marrnet2_noft.load_state_dict(sd)
        self.d = VoxelDiscriminator(nf=d_nf)
        if gan_path:
            self.d.load_state_dict(torch.load(gan_path, map_location="cpu")["nets"][1])
        for frozen in (self.d, self.marrnet2_noft):
            for p in frozen.parameters():
                p.requires_grad = False
        self.sigmoid = nn.Sigmoid()

    def forward(self, input_struct):
        with torch.no_grad():
            noft = self.marrnet2_noft(input_struct)
        voxel = self.marrnet2(input_struct)
        return {"voxel_noft": noft, "voxel": voxel, "is_real": self.d(self.sigmoid(voxel))}


def shapehd_loss(pred, gt_voxel, w_gan_loss=0.0):
    """supervised BCE on the logits + w * (-critic score) (models/shapehd.py:67-79)"""
    sup = F.binary_cross_entropy_with_logits(pred["voxel"], gt_voxel)
    gan = -pred["is_real"].mean() * w_gan_loss
    return sup + gan, {"sup": sup.detach(), "gan": gan.detach()}


class WGANGP:
    """3-D WGAN-GP (models/wgangp.py): one critic step (real, fake, gradient penalty with create_graph=True so the
    penalty itself is differentiated) and, every gan_d_iter-th batch, one generator step.  `net_d` / `net_g` may be
    DistributedDataParallel wrappers: each loss.backward() then all-reduces that network's gradients over RCCL."""

    def __init__(self, net_g=None, net_d=None, nz=200, lr=1e-4, betas=(0.5, 0.9), lam=10.0, norm=1.0, d_iter=1,
                 generator=None):
        """generator: optional CPU torch.Generator -- latent codes and the interpolation weights of the penalty are then
        drawn on the host from it and copied to the device (a run is reproducible across devices); default: drawn on
        the device, as the reference does (wgangp.py:98,134)"""
        self.net_g = net_g if net_g is not None else VoxelGenerator(nz)
        self.net_d = net_d if net_d is not None else VoxelDiscriminator()
        self.nz, self.lam, self.norm, self.d_iter = nz, lam, norm, d_iter
        self.opt_g = torch.optim.Adam(self.net_g.parameters(), lr=lr, betas=betas)
        self.opt_d = torch.optim.Adam(self.net_d.parameters(), lr=lr, betas=betas)
        self._last_err_g = None
        self.generator = generator

    def _random(self, fn, shape, device):
        if self.generator is None:
            return fn(*shape, device=device)
        return fn(*shape, generator=self.generator).to(device)

    def sample(self, n, device):
        return self.net_g(self._random(torch.randn, (n, self.nz, 1, 1, 1), device))

    def grad_penalty(self, real, fake):
        alpha = self._random(torch.rand, (real.shape[0],) + (1,) * (real.dim() - 1), real.device)
        inter = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
        score = self.net_d(inter)
        from ..networks.thin_conv import input_grad_only
        with input_grad_only():     # this pass wants d score / d inter and nothing else: no weight-gradient GEMMs, no nodes for them
            grads, = torch.autograd.grad(score, inter, torch.ones_like(score), create_graph=True, retain_graph=True)
        gn = (grads.reshape(grads.size(0), -1) + 1e-16).norm(2, dim=1)
        return ((gn - self.norm) ** 2).mean() * self.lam

    def train_on_batch(self, batch_idx, real):
        log = {}
        for p in self.net_d.parameters():
            p.requires_grad = True
        self.opt_d.zero_grad(set_to_none=True)
        with torch.no_grad():
            fake = self.sample(real.shape[0], real.device)
        err_real, err_fake = self.net_d(real).mean(), self.net_d(fake).mean()
        loss_d = err_fake - err_real
        if self.lam > 0:
            gp = self.grad_penalty(real, fake)
            loss_d = loss_d + gp
            log["err_d_gp"] = gp.detach()
        loss_d.backward()
        self.opt_d.step()
        log.update(err_d_real=-err_real.detach(), err_d_fake=err_fake.detach(), err_d=loss_d.detach())
        if batch_idx % self.d_iter == 0:
            for p in self.net_d.parameters():
                p.requires_grad = False
            self.opt_g.zero_grad(set_to_none=True)
            err_g = -self.net_d(self.sample(real.shape[0], real.device)).mean()
            err_g.backward()
            self.opt_g.step()
            self._last_err_g = err_g.detach()
        log["err_g"] = self._last_err_g
        return log
