"""SphericalBackProjection -- drop-in for
toolbox/cam_bp/cam_bp/functions/sperical_to_tdf.py:10-47 (file name keeps the reference's
spelling so its import line works).  Returns ``(tdf, cnt)``; backward takes the phony
gradient of ``cnt`` and returns ``(grad_depth, None, None)``.

Dropped on purpose: the two host-synchronising ``np.isnan(torch.sum(...))`` asserts and the
``pdb.set_trace()`` of the reference's backward (:37,:43-46) -- they stall the stream and make
the op un-capturable in a HIP graph; values are unchanged.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .._ext import cam_bp_lib


class SphericalBackProjection(Function):

    @staticmethod
    def forward(ctx, spherical, grid, res=128):
        assert spherical.dim() == 4
        assert grid.dim() == 5
        assert tuple(grid.shape) == tuple(spherical.shape) + (3,)
        assert spherical.is_cuda and grid.is_cuda
        n, nc = spherical.shape[0], spherical.shape[1]
        tdf = torch.empty((n, nc, res, res, res), dtype=spherical.dtype, device=spherical.device)
        cnt = torch.empty_like(tdf)
        cam_bp_lib.spherical_back_proj_forward(spherical, grid, tdf, cnt)
        ctx.save_for_backward(spherical.detach(), grid, cnt)
        ctx.depth_shape = spherical.shape
        ctx.mark_non_differentiable(cnt)
        return tdf, cnt

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output, grad_phony):
        assert grad_output.is_cuda
        spherical, grid, cnt = ctx.saved_tensors
        grad_depth = torch.empty(ctx.depth_shape, dtype=grad_output.dtype, device=grad_output.device)
        cam_bp_lib.spherical_back_proj_backward(spherical, grid, cnt, grad_output, grad_depth)
        return grad_depth, None, None
