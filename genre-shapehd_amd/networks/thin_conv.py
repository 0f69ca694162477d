"""Weight gradients of the 3-D convolutions at 64^3 and 128^3 as blocked GEMMs -- the layers on which MIOpen has no usable
weight-gradient solver on gfx950 (ROCm 7.2).  Measured (profiles/r04a_train_*_kernel_stats.txt, fp32, one MI355X; the
find-db entries of the same run):

    layer (reference: networks/networks.py, networks/uresnet.py)     weight gradient, stock            here
    ConvTranspose3d(32 -> 1, k4 s2 p1) 64^3 -> 128^3, batch 8      283 ms  ConvDirectNaiveConvWrw     GEMM
      (MarrNet-2 / ShapeHD decoder output, marrnet2.py:88-111)      = 77 % of the whole ShapeHD step
    ConvTranspose3d(64 -> 1), batch 8  (3-D GAN generator output)   283 ms  ConvDirectNaiveConvWrw
    ConvTranspose3d(40 -> 1), batch 4  (Unet_3D dec6)               132 ms  ConvDirectNaiveConvWrw
    Conv3d(2 -> 20, k8 s2 p3) 128^3 -> 64^3, batch 4 (Unet_3D enc1) 155 ms  ConvHipImplicitGemm3DGroupWrwXdlops
    ConvTranspose3d(64 -> 32, k4 s2 p1) 32^3 -> 64^3, batch 8       40 ms  ConvHipImplicitGemm3DGroupWrwXdlops (CK batched GEMM)
    ConvTranspose3d(64 -> 64), batch 8 (generator)                   41 ms  same
    Conv3d(20 -> 40, k4 s2 p1) 64^3 -> 32^3, batch 4 (Unet_3D enc2)  20 ms  same
    ConvTranspose3d(80 -> 20, k8 s2 p3) 32^3 -> 64^3 (Unet_3D dec5)  20 ms  same

MIOpen's find step ranks a *naive reference kernel* first for them (the implicit-GEMM and GEMM solvers are slower still):
with one output (or two input) channels the weight gradient is a [C x taps] matrix reduced over 2-17 M voxels, a shape
none of the tiled solvers is built for.  It is a plain GEMM once the thin side is unfolded:

    transposed:  gw[ci, co, k] = sum_{n,i} x[n, ci, i] * gy_pad[n, co, i*s + k]
    regular:     gw[co, ci, k] = sum_{n,o} gy[n, co, o] * x_pad[n, ci, o*s + k]

i.e. (channels x voxels) @ (voxels x thin*taps) with the unfolded operand a strided VIEW of the padded tensor (materialised
once per call by the contraction: <= 0.55 GB at the shapes above, in chunks of taps beyond that) and the voxels cut
into blocks that run as a batched GEMM (a single [32 x 64] output tile would occupy one workgroup of the chip).  Forward and
data gradient stay MIOpen's (its solvers for those directions are fine: 0.3-4 ms).

The modules are drop-in subclasses: same parameters, same state_dict keys, same forward values; float64 / CPU tensors take
the stock path.  The data gradient can be differentiated a second time (the WGAN-GP gradient penalty), and that second
differentiation meets the GEMM weight gradient again; the weight gradient itself is first order."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
import torch.nn.functional as F

import contextlib

_INPUT_GRAD_ONLY = False    # see input_grad_only()


@contextlib.contextmanager
def input_grad_only():
    """Inside this context the backward of the two Functions below returns the DATA gradient only.  `ctx.needs_input_grad` is
    fixed when a Python Function runs forward, so `autograd.grad(score, x_hat, create_graph=True)` -- the first-order pass of
    the WGAN-GP penalty, which wants the critic's input gradient and nothing else (models/wgangp.py:131-147) -- would
    otherwise run every layer's unfold + GEMM weight gradient and record a graph node for it, only to drop the result
    (ADVICE r4).  The second differentiation (the penalty's backward, outside this context) still meets the GEMM weight
    gradients of the recorded data-gradient operators."""
    global _INPUT_GRAD_ONLY
    prev, _INPUT_GRAD_ONLY = _INPUT_GRAD_ONLY, True
    try:
        yield
    finally:
        _INPUT_GRAD_ONLY = prev


_BLOCK = 4096               # voxels per GEMM block (K of one batch entry)
_MAX_UNFOLD = 160 << 20     # elements of the unfolded operand materialised at once (0.64 GB of fp32)


def _blocked_contract(a, b):
    """a [N, A, P], b [N, B, P] (b may be a non-contiguous view) -> [A, B] = sum_{n,p} a[n,:,p] b[n,:,p]^T, the voxel axis cut into
    blocks that become GEMM batch entries"""
    n, ca, p = a.shape
    cb = b.shape[1]
    blk = _BLOCK
    while p % blk:
        blk //= 2
    a4 = a.reshape(n, ca, p // blk, blk)
    b4 = b.reshape(n, cb, p // blk, blk)
    part = torch.einsum("nacp,nbcp->ncab", a4, b4)              # batch = (n, block)
    return part.sum((0, 1))


def transposed_weight_grad(x, gy, weight_shape, stride, padding):
    """weight gradient of conv_transpose3d(x, w, stride, padding): x [N,Ci,I..], gy [N,Co,O..] -> [Ci,Co,k,k,k]"""
    ci, co, kd, kh, kw = weight_shape
    n = x.shape[0]
    i3 = x.shape[2:]
    s, p = stride, padding
    # gy_pad[o + p] = gy[o]; tap k of input voxel i reads gy_pad[i*s + k]
    need = [(i3[d] - 1) * s[d] + weight_shape[2 + d] for d in range(3)]
    pads = []
    for d in (2, 1, 0):
        pads += [p[d], max(need[d] - p[d] - gy.shape[2 + d], 0)]
    gp = F.pad(gy.to(x.dtype), pads)                                   # (autocast: an fp16 gy against an fp32 x)
    xf = x.reshape(n, ci, -1)
    out = x.new_zeros((ci, co, kd, kh, kw))
    # the unfolded operand is materialised in pieces of <= _MAX_UNFOLD elements: first over taps, then -- when one plane of
    # taps of the whole batch is still too large (batch 32 at 128^3) -- over batch items too
    per_item = co * xf.shape[2] * kh * kw
    nb = max(1, min(n, _MAX_UNFOLD // max(1, per_item)))
    step = max(1, min(kd, _MAX_UNFOLD // max(1, nb * per_item)))
    for n0 in range(0, n, nb):
        n1 = min(n, n0 + nb)
        for k0 in range(0, kd, step):
            k1 = min(kd, k0 + step)
            # view [N', Co, I0, I1, I2, kd', kh, kw]
            v = gp[n0:n1, :, k0:k0 + (i3[0] - 1) * s[0] + (k1 - k0)].unfold(2, k1 - k0, s[0]).unfold(3, kh, s[1]).unfold(4, kw, s[2])
            v = v[:, :, :i3[0], :i3[1], :i3[2]]
            g = v.permute(0, 1, 5, 6, 7, 2, 3, 4).reshape(n1 - n0, co * (k1 - k0) * kh * kw, -1)      # materialised here
            r = _blocked_contract(xf[n0:n1], g)                                                       # [Ci, Co*k'*kh*kw]
            out[:, :, k0:k1] += r.reshape(ci, co, k1 - k0, kh, kw)
    return out


def regular_weight_grad(x, gy, weight_shape, stride, padding):
    """weight gradient of conv3d(x, w, stride, padding): x [N,Ci,I..], gy [N,Co,O..] -> [Co,Ci,k,k,k]"""
    co, ci, kd, kh, kw = weight_shape
    n = x.shape[0]
    o3 = gy.shape[2:]
    s, p = stride, padding
    need = [(o3[d] - 1) * s[d] + weight_shape[2 + d] for d in range(3)]
    pads = []
    for d in (2, 1, 0):
        pads += [p[d], max(need[d] - p[d] - x.shape[2 + d], 0)]
    xp = F.pad(x, pads)
    gf = gy.to(x.dtype).reshape(n, co, -1)
    out = x.new_zeros((co, ci, kd, kh, kw))
    per_item = ci * gf.shape[2] * kh * kw
    nb = max(1, min(n, _MAX_UNFOLD // max(1, per_item)))
    step = max(1, min(kd, _MAX_UNFOLD // max(1, nb * per_item)))
    for n0 in range(0, n, nb):
        n1 = min(n, n0 + nb)
        for k0 in range(0, kd, step):
            k1 = min(kd, k0 + step)
            v = xp[n0:n1, :, k0:k0 + (o3[0] - 1) * s[0] + (k1 - k0)].unfold(2, k1 - k0, s[0]).unfold(3, kh, s[1]).unfold(4, kw, s[2])
            v = v[:, :, :o3[0], :o3[1], :o3[2]]
            u = v.permute(0, 1, 5, 6, 7, 2, 3, 4).reshape(n1 - n0, ci * (k1 - k0) * kh * kw, -1)
            r = _blocked_contract(gf[n0:n1], u)                                                       # [Co, Ci*k'*kh*kw]
            out[:, :, k0:k1] += r.reshape(co, ci, k1 - k0, kh, kw)
    return out


class _TransposedWgradFn(Function):
    """the weight gradient itself as an opaque op: under create_graph=True (the WGAN-GP penalty differentiates the critic's INPUT
    gradient a second time, not this) no graph is recorded through the unfold / GEMM and nothing but x and gy is kept"""

    @staticmethod
    def forward(ctx, x, gy, weight_shape, stride, padding):
        return transposed_weight_grad(x, gy, weight_shape, stride, padding)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        raise NotImplementedError("thin_conv: the weight gradient is not differentiable a second time (use nn.ConvTranspose3d)")


class _RegularWgradFn(Function):
    @staticmethod
    def forward(ctx, x, gy, weight_shape, stride, padding):
        return regular_weight_grad(x, gy, weight_shape, stride, padding)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        raise NotImplementedError("thin_conv: the weight gradient is not differentiable a second time (use nn.Conv3d)")


class _ThinConvTranspose3dFn(Function):
    """y = conv_transpose3d(x, w, b).  The backward is written with differentiable operators -- the data gradient is the adjoint
    convolution THROUGH _ThinConv3dFn, so that a second differentiation (create_graph=True: the gradient penalty of the 3-D
    WGAN-GP runs the critic's input gradient through autograd once more, models/wgangp.py:131-147) meets the GEMM weight
    gradient again instead of MIOpen's; the weight gradient is opaque (first order only)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, output_padding=(0, 0, 0)):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, output_padding, bias is not None)
        return F.conv_transpose3d(x, weight, bias, stride, padding, output_padding)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, padding, output_padding, has_bias = ctx.cfg
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:                                   # adjoint of a transposed convolution: a convolution
            gx = _ThinConv3dFn.apply(gy, weight, None, stride, padding)
        if ctx.needs_input_grad[1] and not _INPUT_GRAD_ONLY:
            gw = _TransposedWgradFn.apply(x, gy, tuple(weight.shape), stride, padding)
        if has_bias and ctx.needs_input_grad[2] and not _INPUT_GRAD_ONLY:
            gb = gy.sum((0, 2, 3, 4))
        return gx, gw, gb, None, None, None


class _ThinConv3dFn(Function):
    """y = conv3d(x, w, b); see _ThinConvTranspose3dFn (its mirror image)"""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, bias is not None)
        return F.conv3d(x, weight, bias, stride, padding)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, padding, has_bias = ctx.cfg
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            opad = tuple(x.shape[2 + d] - ((gy.shape[2 + d] - 1) * stride[d] - 2 * padding[d] + weight.shape[2 + d]) for d in range(3))
            gx = _ThinConvTranspose3dFn.apply(gy, weight, None, stride, padding, opad)
        if ctx.needs_input_grad[1] and not _INPUT_GRAD_ONLY:
            gw = _RegularWgradFn.apply(x, gy, tuple(weight.shape), stride, padding)
        if has_bias and ctx.needs_input_grad[2] and not _INPUT_GRAD_ONLY:
            gb = gy.sum((0, 2, 3, 4))
        return gx, gw, gb, None, None


_BIG = 64 ** 3              # voxels per sample (of the larger side) from which MIOpen's weight-gradient solvers fall off


def _custom_path(x, mod, voxels):
    """the GEMM weight gradient serves fp32 training on the GPU at >= 64^3; everything else (CPU, float64, inference, small
    volumes) takes the stock operator"""
    if getattr(mod, "force_custom", False):
        return True
    return (x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and mod.weight.requires_grad
            and voxels >= _BIG and not getattr(mod, "force_stock", False) and mod.dilation == (1, 1, 1) and mod.groups == 1)


class ThinConvTranspose3d(nn.ConvTranspose3d):
    """nn.ConvTranspose3d whose weight gradient is the blocked GEMM above when its OUTPUT has >= 64^3 voxels"""

    def forward(self, x, output_size=None):
        vox = x.shape[2] * x.shape[3] * x.shape[4] * self.stride[0] * self.stride[1] * self.stride[2] if x.dim() == 5 else 0
        if output_size is not None or self.output_padding != (0, 0, 0) or not _custom_path(x, self, vox):
            return super().forward(x, output_size)
        return _ThinConvTranspose3dFn.apply(x, self.weight, self.bias, self.stride, self.padding)


class ThinConv3d(nn.Conv3d):
    """nn.Conv3d whose weight gradient is the blocked GEMM above when its INPUT has >= 64^3 voxels"""

    def forward(self, x):
        vox = x.shape[2] * x.shape[3] * x.shape[4] if x.dim() == 5 else 0
        if self.padding_mode != "zeros" or not _custom_path(x, self, vox):
            return super().forward(x)
        return _ThinConv3dFn.apply(x, self.weight, self.bias, self.stride, self.padding)
