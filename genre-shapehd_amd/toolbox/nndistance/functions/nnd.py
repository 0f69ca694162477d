"""NNDFunction and the nndistance helpers -- drop-ins for
toolbox/nndistance/functions/nnd.py:8-85 (Chamfer nearest-neighbour distance, squared L2,
both directions; int32 indices).

CUDA (HIP) tensors run the gfx950 kernels (my_lib.nnd_*_cuda); CPU tensors take the reference's
CPU entry points my_lib.nnd_forward / nnd_backward (:27-28,53-54), host code in csrc/nnd_host.hip
-- the dispatch is by the tensors' device, exactly as in the reference; a CUDA tensor never falls
back to the host.  Outputs are allocated on the inputs' device directly (the reference allocates
on the host and copies, :21-33).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .._ext import my_lib


class NNDFunction(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        assert xyz1.dim() == 3 and xyz2.dim() == 3
        assert xyz1.size(0) == xyz2.size(0)
        assert xyz1.size(2) == 3 and xyz2.size(2) == 3
        assert xyz1.is_cuda == xyz2.is_cuda
        assert xyz1.dtype == torch.float32 and xyz2.dtype == torch.float32, \
            'only FloatTensor are supported for NNDistance'
        assert xyz1.is_contiguous() and xyz2.is_contiguous()
        ctx.is_cuda = xyz1.is_cuda
        b, n, _ = xyz1.size()
        m = xyz2.size(1)
        dev = xyz1.device
        dist1 = torch.empty((b, n), dtype=torch.float32, device=dev)
        dist2 = torch.empty((b, m), dtype=torch.float32, device=dev)
        idx1 = torch.empty((b, n), dtype=torch.int32, device=dev)
        idx2 = torch.empty((b, m), dtype=torch.int32, device=dev)
        (my_lib.nnd_forward_cuda if ctx.is_cuda else my_lib.nnd_forward)(xyz1, xyz2, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    @once_differentiable
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        """takes placeholder grads for the two index outputs, like the reference (:41-43)"""
        assert graddist1.is_cuda == ctx.is_cuda and graddist2.is_cuda == ctx.is_cuda
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1 = graddist1.contiguous()
        graddist2 = graddist2.contiguous()
        assert graddist1.dtype == torch.float32 and graddist2.dtype == torch.float32, \
            'only FloatTensor are supported for NNDistance'
        gradxyz1 = torch.empty_like(xyz1)
        gradxyz2 = torch.empty_like(xyz2)
        (my_lib.nnd_backward_cuda if ctx.is_cuda else my_lib.nnd_backward)(xyz1, xyz2, gradxyz1, gradxyz2, graddist1,
                                                                            graddist2, idx1, idx2)
        return gradxyz1, gradxyz2


def nndistance_w_idx(xyz1, xyz2):
    return NNDFunction.apply(xyz1.contiguous(), xyz2.contiguous())


def nndistance(xyz1, xyz2):
    # [B,3,n] inputs are transposed to [B,n,3] (:73-76)
    if xyz1.size(2) != 3:
        xyz1 = xyz1.transpose(1, 2)
    if xyz2.size(2) != 3:
        xyz2 = xyz2.transpose(1, 2)
    dist1, dist2, _, _ = NNDFunction.apply(xyz1.contiguous(), xyz2.contiguous())
    return dist1, dist2


def nndistance_score(xyz1, xyz2, eps=1e-10):
    dist1, dist2 = nndistance(xyz1, xyz2)
    return torch.sqrt(dist1 + eps).mean(1) + torch.sqrt(dist2 + eps).mean(1)
