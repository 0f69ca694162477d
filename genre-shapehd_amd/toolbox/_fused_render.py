"""Fused render_spherical.  Present only when libgenre_hip.so exports
genre_render_spherical_forward / _backward."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .calc_prob.calc_prob._ext import _loader


def available():
    return _loader().has_symbol("genre_render_spherical_forward")


class RenderSphericalFused(Function):
    @staticmethod
    def forward(ctx, vox, sph_res, z_res):
        assert vox.dim() == 5 and vox.is_cuda and vox.dtype == torch.float32
        L = _loader()
        n, nc = vox.shape[0], vox.shape[1]
        out = torch.empty((n, nc, sph_res, sph_res), dtype=vox.dtype, device=vox.device)
        cfg = torch.empty((0,), dtype=torch.int32, device=vox.device)  # placeholder, see _loader
        L._call_render_forward(vox, out, z_res)
        ctx.save_for_backward(vox)
        ctx.z_res = z_res
        ctx.sph_res = sph_res
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        vox, = ctx.saved_tensors
        L = _loader()
        grad_vox = torch.zeros_like(vox)
        L._call_render_backward(vox, grad_out.contiguous(), grad_vox, ctx.z_res)
        return grad_vox, None, None
