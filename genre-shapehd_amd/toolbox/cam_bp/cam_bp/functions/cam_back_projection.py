"""CameraBackProjection -- drop-in for the reference's autograd Function
(toolbox/cam_bp/cam_bp/functions/cam_back_projection.py:9-46): same signature,
same returned gradients ``(grad_depth, grad_fl, grad_camdist, None)``.

Differences that do not change results: outputs are ``torch.empty`` (the native
op writes every element, so the reference's two ``zero_()`` passes and the
``+ 1/res`` pass, :22-24, disappear), and ``cnt`` is kept both on ``ctx`` (as the
reference does, :28) and in ``saved_tensors``.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .._ext import cam_bp_lib


def leader_halo(res, fl, cam_dist):
    """csrc/cam_bp.hip: leader_halo restated -- how many pixels apart two points of one voxel can project for a camera (fl,
    cam_dist) looking at the unit cube from (-cam_dist, 0, 0); -1: the camera is too close for the bound.  The product does not
    call it (the library decides: cam_bp_lib.forward_plan); tests/test_cam_leader_host.py pins the bound against it."""
    x_min = float(cam_dist) - 0.5
    if not (x_min > 0.05) or not (fl > 0):
        return -1
    b = float(fl) * ((1.0 / res) / x_min + 0.5 * (1.0 / res) / (x_min * x_min))
    return int(b * (1.0 + 1e-4) + 1e-3)


class CameraBackProjection(Function):

    @staticmethod
    def forward(ctx, depth_t, fl, cam_dist, res=128):
        assert depth_t.dim() == 4
        n, nc = depth_t.shape[0], depth_t.shape[1]
        assert fl.dim() == 2 and tuple(fl.shape) == (n, nc)
        assert cam_dist.dim() == 2 and tuple(cam_dist.shape) == (n, nc)
        assert depth_t.is_cuda and fl.is_cuda and cam_dist.is_cuda
        tdf = torch.empty((n, nc, res, res, res), dtype=depth_t.dtype, device=depth_t.device)
        cnt = torch.empty_like(tdf)
        cam_bp_lib.back_projection_forward(depth_t, cam_dist, fl, tdf, cnt)
        ctx.save_for_backward(depth_t, fl, cam_dist, cnt)
        ctx.cnt_forward = cnt
        ctx.depth_shape = depth_t.shape
        return tdf

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        assert grad_output.is_cuda
        depth_t, fl, cam_dist, cnt = ctx.saved_tensors
        n, nc = ctx.depth_shape[0], ctx.depth_shape[1]
        grad_depth = torch.empty(ctx.depth_shape, dtype=grad_output.dtype, device=grad_output.device)
        grad_fl = torch.empty((n, nc), dtype=grad_output.dtype, device=grad_output.device)
        grad_camdist = torch.empty_like(grad_fl)
        # note the fl / cam_dist order flip between forward and backward (back_projection.h:1-2)
        cam_bp_lib.back_projection_backward(depth_t, fl, cam_dist, cnt, grad_output,
                                            grad_depth, grad_camdist, grad_fl)
        return grad_depth, grad_fl, grad_camdist, None


class ShiftedCameraBackProjection(Function):
    """CameraBackProjection followed by Camera_back_projection_layer.shift_tdf
    (camera_backprojection_module.py:25-28), 1 - res*tdf, evaluated inside the native op: same
    values, one full-volume elementwise pass less in each direction.  Used by the layer."""

    @staticmethod
    def forward(ctx, depth_t, fl, cam_dist, res=128, batch_minor=False, const=None, hint=None):
        """const = (fl, cam_dist) as Python floats when the two tensors are filled with those constants (the layer's
        default call, camera_backprojection_module.py:16-21): the forward then takes the by-value entry point (no
        loads of the camera in front of the brick screen); the tensors are still what the backward reads.
        hint: an empty dict the op fills with its occupancy words for the renderer -- hint["words"] (int32) and hint["cell"]
        (None: the leader pass's words per group of 32 images and renderer brick, image-minor volumes; cx*10000 + cy*100 + cz:
        the brick kernel's words per image and cell, dense volumes) -- when the by-value entry serves the call"""
        assert depth_t.dim() == 4
        n, nc = depth_t.shape[0], depth_t.shape[1]
        assert fl.dim() == 2 and tuple(fl.shape) == (n, nc)
        assert cam_dist.dim() == 2 and tuple(cam_dist.shape) == (n, nc)
        assert depth_t.is_cuda and fl.is_cuda and cam_dist.is_cuda
        if batch_minor and nc == 1:
            # same logical [n,1,res,res,res] tensors, image index fastest in memory: the layout the fused renderer's
            # batch-minor kernels want (the native op is stride-generic, its fill treats any dense tensor as flat)
            strides = (1, n * res ** 3, res * res * n, res * n, n)
            out = torch.empty_strided((n, nc, res, res, res), strides, dtype=depth_t.dtype, device=depth_t.device)
            cnt = torch.empty_strided((n, nc, res, res, res), strides, dtype=depth_t.dtype, device=depth_t.device)
        else:
            out = torch.empty((n, nc, res, res, res), dtype=depth_t.dtype, device=depth_t.device)
            cnt = torch.empty_like(out)
        # the by-value entry runs (a) the single-launch brick kernel on dense NCXYZ outputs whose z rows are float4-aligned, or
        # (b) fill + the deterministic, atomic-free leader pass on other layouts, for cameras whose voxels project to <= 4 pixels.
        # Which one -- or neither (res = 30, 126, ...; the scatter / gather modes; a very close camera: the tensor entry then) --
        # is the LIBRARY's decision, asked of it for the very tensors it will write (cam_bp_lib.forward_plan; ADVICE r3 / r5:
        # nothing is mirrored here)
        plan = cam_bp_lib.forward_plan(out, cnt, const[1], const[0]) if const is not None else cam_bp_lib.PLAN_NONE
        if plan != cam_bp_lib.PLAN_NONE:
            words = None
            if hint is not None and nc == 1:
                if plan == cam_bp_lib.PLAN_LEADER:
                    from ...._fused_render import new_brick_words
                    words = new_brick_words(n, res, depth_t.device)
                    hint["cell"] = None
                else:
                    cx, cy, cz = cam_bp_lib.cam_cell()
                    words = torch.empty((n * nc, -(-res // cx), -(-res // cy), -(-res // cz)), dtype=torch.int32,
                                        device=depth_t.device)
                    hint["cell"] = cx * 10000 + cy * 100 + cz
                hint["words"] = words
            # (leader pass: cnt is kept for THIS Function's backward alone, which reads it at the voxel of every in-grid pixel and
            # nowhere else -- the leader pass then writes only those elements and half of the fill disappears)
            cam_bp_lib.back_projection_forward_const(depth_t, const[1], const[0], out, cnt, shifted=True, tile_live=words,
                                                     sparse_cnt=plan == cam_bp_lib.PLAN_LEADER)
        else:
            cam_bp_lib.back_projection_forward_shifted(depth_t, cam_dist, fl, out, cnt)
        ctx.save_for_backward(depth_t, fl, cam_dist, cnt)
        ctx.depth_shape = depth_t.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        depth_t, fl, cam_dist, cnt = ctx.saved_tensors
        n, nc = ctx.depth_shape[0], ctx.depth_shape[1]
        grad_depth = torch.empty(ctx.depth_shape, dtype=grad_output.dtype, device=grad_output.device)
        grad_fl = torch.empty((n, nc), dtype=grad_output.dtype, device=grad_output.device)
        grad_camdist = torch.empty_like(grad_fl)
        # the renderer's backward says which images' gradient is identically zero (the clamp in front of it blocked every voxel:
        # GenRe's own chain) -- toolbox/_fused_render.py: attach_zero_hint
        from ...._fused_render import zero_hint_of
        zh = zero_hint_of(grad_output)
        if zh is not None:
            cam_bp_lib.back_projection_backward_hinted(depth_t, fl, cam_dist, cnt, grad_output, grad_depth, grad_camdist,
                                                       grad_fl, *zh, shifted=True)
        else:
            cam_bp_lib.back_projection_backward_shifted(depth_t, fl, cam_dist, cnt, grad_output,
                                                        grad_depth, grad_camdist, grad_fl)
        return grad_depth, grad_fl, grad_camdist, None, None, None, None
