"""GenRe caller glue around the geometric ops (first step).

The reference's models call the toolbox ops with a few full-volume elementwise passes in between
(models/depth_pred_with_sph_inpaint.py:120-126, models/genre_full_model.py:122-143).  This module
re-hosts exactly those two glue sections with the passes folded into the native ops, so that the
refiner input [N,2,128^3] is written once.  The networks themselves (net1, net2, Unet_3D) are stock
torch.nn and stay with the caller.

    geo = GenReGeometry().cuda()
    pred_abs_depth = geo.get_abs_depth(pred['depth'], pred['depth_minmax'], input_struct.silhou)   # :131-142
    proj_depth, sph_in = geo.depth_to_spherical(pred_abs_depth)       # :120-126  -> net2
    refine_input, cnt  = geo.refiner_input(pred_sph_full, proj_depth) # :122-127,134-143 -> Unet_3D
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .toolbox.cam_bp.cam_bp._ext import cam_bp_lib, _loader
from .toolbox.cam_bp.cam_bp.modules.camera_backprojection_module import Camera_back_projection_layer
from .toolbox.spherical_proj import gen_sph_grid, render_spherical, sph_pad

LO, HI = 1e-5, 1 - 1e-5


class AbsDepth(Function):
    """(pred_depth [N,1,H,W], depth_minmax [N,2], silhou [N,1,H,W], scale_25d) -> abs depth [N,1,W,H], i.e.
    depth_pred_with_sph_inpaint.py:131-142:

        d = to_abs_depth(1 - pred_depth / scale, depth_minmax.detach());  d[silhou.detach() / scale < 0.5] = 0
        d = flip(d.permute(0, 1, 3, 2), [2])

    one native pass instead of the reference's six elementwise / copy kernels; gradient w.r.t. pred_depth only
    (the reference detaches the other two, :135,137)."""

    @staticmethod
    def forward(ctx, pred_depth, depth_minmax, silhou, scale_25d=100.0):
        assert pred_depth.dim() == 4 and pred_depth.is_cuda and pred_depth.dtype == torch.float32
        n, c, h, w = pred_depth.shape
        out = torch.empty((n, c, w, h), dtype=pred_depth.dtype, device=pred_depth.device)
        mm = depth_minmax.detach().reshape(n, 2)
        _loader().glue_lib.abs_depth_forward(pred_depth, mm, silhou.detach(), out, float(scale_25d))
        ctx.save_for_backward(mm, silhou.detach())
        ctx.scale = float(scale_25d)
        ctx.shape = pred_depth.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        mm, silhou = ctx.saved_tensors
        grad_pred = torch.empty(ctx.shape, dtype=grad_out.dtype, device=grad_out.device)
        _loader().glue_lib.abs_depth_backward(grad_out, mm, silhou, grad_pred, ctx.scale)
        return grad_pred, None, None, None


class RefinerInput(Function):
    """(pred_sph_full [N,1,H,W], grid [N,1,h,w,3], proj_depth [N,1,R,R,R], margin) ->
    (refine_input [N,2,R,R,R], cnt [N,1,R,R,R]) with

        channel 0 = (-tdf + 1/R) * R * clamp(cnt,0,1),  (tdf, cnt) = SphericalBackProjection(1 - crop(sph), grid, R)
        channel 1 = clamp(proj_depth / 50, 1e-5, 1-1e-5)

    i.e. genre_full_model.py:134-143 and :125-126.  Channel 0 is written by the native op straight into
    the output (its normalise pass applies the post-transform), channel 1 by one elementwise kernel with
    `out=`; the reference's torch.cat copy and four full-volume elementwise passes disappear."""

    @staticmethod
    def forward(ctx, pred_sph_full, grid, proj_depth, margin):
        n, _, h, w = pred_sph_full.shape
        res = proj_depth.shape[2]
        inv = (1 - pred_sph_full[:, :, margin:h - margin, margin:w - margin]).contiguous()
        out = torch.empty((n, 2, res, res, res), dtype=proj_depth.dtype, device=proj_depth.device)
        cnt = torch.empty((n, 1, res, res, res), dtype=proj_depth.dtype, device=proj_depth.device)
        cam_bp_lib.spherical_back_proj_forward_shifted(inv, grid, out[:, 0:1], cnt)
        torch.clamp(proj_depth / 50, LO, HI, out=out[:, 1:2])
        ctx.save_for_backward(inv, grid, cnt, proj_depth)
        ctx.margin = margin
        ctx.sph_shape = pred_sph_full.shape
        ctx.mark_non_differentiable(cnt)
        return out, cnt

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out, grad_cnt):
        inv, grid, cnt, proj_depth = ctx.saved_tensors
        m = ctx.margin
        grad_sph = grad_proj = None
        if ctx.needs_input_grad[0]:
            gd = torch.empty_like(inv)
            cam_bp_lib.spherical_back_proj_backward_shifted(inv, grid, cnt, grad_out[:, 0:1], gd)
            grad_sph = torch.zeros(ctx.sph_shape, dtype=gd.dtype, device=gd.device)
            grad_sph[:, :, m:ctx.sph_shape[2] - m, m:ctx.sph_shape[3] - m] = -gd          # d(1 - crop)/d sph
        if ctx.needs_input_grad[2]:
            t = proj_depth / 50
            grad_proj = grad_out[:, 1:2] * ((t >= LO) & (t <= HI)).to(grad_out.dtype) / 50
        return grad_sph, None, grad_proj, None


class GenReGeometry(nn.Module):
    def __init__(self, padding_margin=16, res=128, batch_minor=False):
        """batch_minor: inference batches of >= 16 images keep the projected volume image-minor in memory, the
        layout in which the fused renderer's forward is fastest (see Camera_back_projection_layer)"""
        super().__init__()
        self.margin = padding_margin
        self.res = res
        self.proj_depth = Camera_back_projection_layer(res, batch_minor=batch_minor)
        self.render_spherical = render_spherical()
        self.register_buffer('grid', gen_sph_grid(res))                  # genre_full_model.py:108

    def get_abs_depth(self, pred_depth, depth_minmax, silhou, scale_25d=100.0):
        """depth_pred_with_sph_inpaint.py:131-142 (scale_25d: marrnetbase.py:17) -> depth map for proj_depth"""
        return AbsDepth.apply(pred_depth, depth_minmax, silhou, scale_25d)

    def depth_to_spherical(self, pred_abs_depth):
        """depth_pred_with_sph_inpaint.py:120-129 -> (out_1['proj_depth'], out_1['pred_sph_partial'])"""
        proj = self.proj_depth(pred_abs_depth)                           # 1 - 128*tdf, shift folded into the op
        # == sph_pad(render(clamp(proj*50, 1e-5, 1-1e-5)), margin): clamp and padding both folded into the renderer
        sph_in = self.render_spherical(proj, pre_scale=50.0, pad=self.margin)
        return proj * 50, sph_in

    def refiner_input(self, pred_sph_full, proj_depth):
        """genre_full_model.py:122-127,134-143 -> (refine_input [N,2,R,R,R], cnt)"""
        grid = self.grid.expand(pred_sph_full.shape[0], -1, -1, -1, -1)
        return RefinerInput.apply(pred_sph_full, grid, proj_depth, self.margin)
