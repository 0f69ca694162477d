"""CalcStopProb -- drop-in for toolbox/calc_prob/calc_prob/functions/calc_prob.py:9-29.

forward: stop_prob[z] = p[z] * prod_{k<z}(1 - p[k]) along the last dim.
backward: the reference forms ``stop_prob * grad_in`` in Python (:27) and passes it to
calc_prob_backward; here that product is folded into the kernel
(``calc_prob_backward_fused``), same fp32 product, one pass less over three 16 MiB tensors.
The outputs are ``torch.empty``: the native op writes every element, so the reference's
``zero_()`` fills (:15-16, :25-26) are gone.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .._ext import calc_prob_lib


class CalcStopProb(Function):
    @staticmethod
    def forward(ctx, prob_in):
        assert prob_in.dim() == 5
        assert prob_in.dtype == torch.float32
        assert prob_in.is_cuda
        stop_prob = torch.empty_like(prob_in)
        calc_prob_lib.calc_prob_forward(prob_in, stop_prob)
        ctx.save_for_backward(prob_in, stop_prob)
        return stop_prob

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_in):
        prob_in, stop_prob = ctx.saved_tensors
        grad_out = torch.empty_like(prob_in)
        calc_prob_lib.calc_prob_backward_fused(prob_in, stop_prob, grad_in, grad_out)
        return grad_out
