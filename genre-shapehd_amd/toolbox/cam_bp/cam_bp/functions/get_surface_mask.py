"""get_surface_mask / get_vox_surface_cnt -- drop-ins for
toolbox/cam_bp/cam_bp/functions/get_surface_mask.py:8-40 (plain functions, no autograd)."""
import torch

from .._ext import cam_bp_lib


def _as_param(v, n, nc, like):
    """float -> [n,nc] tensor on the depth's device (the reference builds
    torch.FloatTensor(n, nc).cuda() and fills it, :28-35)."""
    if isinstance(v, (float, int)):
        return torch.full((n, nc), float(v), dtype=torch.float32, device=like.device)
    return v


def get_vox_surface_cnt(depth_t, fl, cam_dist, res=128):
    assert depth_t.dim() == 4
    n, nc = depth_t.shape[0], depth_t.shape[1]
    assert fl.dim() == 2 and tuple(fl.shape) == (n, nc)
    assert cam_dist.dim() == 2 and tuple(cam_dist.shape) == (n, nc)
    assert depth_t.is_cuda and fl.is_cuda and cam_dist.is_cuda
    tdf = torch.empty((n, nc, res, res, res), dtype=depth_t.dtype, device=depth_t.device)
    cnt = torch.empty_like(tdf)
    cam_bp_lib.back_projection_forward(depth_t, cam_dist, fl, tdf, cnt)
    return cnt


def get_surface_mask(depth_t, fl=784.4645406, cam_dist=2.0, res=128):
    n, nc = depth_t.size(0), depth_t.size(1)
    fl = _as_param(fl, n, nc, depth_t)
    cam_dist = _as_param(cam_dist, n, nc, depth_t)
    cnt = get_vox_surface_cnt(depth_t, fl, cam_dist, res)
    mask = torch.empty_like(cnt)
    cam_bp_lib.get_surface_mask(depth_t, cam_dist, fl, cnt, mask)
    surface_vox = torch.clamp(cnt, min=0.0, max=1.0)
    return surface_vox, mask
