"""Fused render_spherical: one HIP kernel per direction instead of the
reference's grid_sample -> clamp -> CalcStopProb -> matmul -> prod -> add chain
(toolbox/spherical_proj.py:62-72).  Only `vox` is saved for backward; the backward kernel
recomputes the ray in registers."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .calc_prob.calc_prob._ext import _loader


def available():
    return _loader().has_symbol("genre_render_spherical_forward")


class RenderSphericalFused(Function):
    """apply(vox [N,NC,X,Y,Z], dirs64 [R,R,3] float64, depth_weight [ZR]) -> [N,NC,R,R]"""

    @staticmethod
    def forward(ctx, vox, dirs64, depth_weight):
        assert vox.dim() == 5 and vox.is_cuda and vox.dtype == torch.float32
        assert dirs64.dtype == torch.float64 and dirs64.dim() == 3 and dirs64.is_contiguous()
        lib = _loader().render_lib
        res = dirs64.shape[0]
        out = torch.empty((vox.shape[0], vox.shape[1], res, res), dtype=vox.dtype, device=vox.device)
        lib.render_spherical_forward(vox, dirs64.view(torch.float32), depth_weight, out)
        ctx.save_for_backward(vox, dirs64, depth_weight)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        vox, dirs64, depth_weight = ctx.saved_tensors
        lib = _loader().render_lib
        grad_vox = torch.empty(vox.shape, dtype=vox.dtype, device=vox.device)
        lib.render_spherical_backward(vox, dirs64.view(torch.float32), depth_weight, grad_out, grad_vox)
        return grad_vox, None, None
