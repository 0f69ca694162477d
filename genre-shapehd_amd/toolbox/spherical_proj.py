"""render_spherical, sph_pad, gen_sph_grid -- drop-ins for toolbox/spherical_proj.py:6-72.

``render_spherical`` keeps the reference's buffers (``grid`` [res,res,z_res,3] and
``depth_weight`` [z_res], :59-60) under the same names and shapes so state_dicts of the
reference load.  Two forward paths, same result within fp32 rounding:

* ``fused=True`` (default when the library exports it): one HIP kernel does the trilinear
  sampling, clamp, stop-probability scan, depth expectation and background term, reading the
  128^3 volume once and writing only the [N,1,res,res] map (the reference materialises four
  16 MiB intermediates and reads the 48 MiB grid buffer per image).
* ``fused=False``: the reference's op sequence (:63-71) on stock PyTorch with
  ``CalcStopProb`` in the middle.  ``grid_sample`` is called with ``align_corners=True``,
  which is what PyTorch 0.4.1 (environment.yml:14) computed.
"""
import numpy as np
import torch

from .calc_prob.calc_prob.functions.calc_prob import CalcStopProb
from .calc_prob.calc_prob._ext import calc_prob_lib   # noqa: F401  (fail early if the library is missing)


def _unit_dirs(res):
    """[res,res,3] float64 unit directions: phi = odd multiples of 180/(2 res) degrees,
    theta = multiples of 360/res degrees (spherical_proj.py:8-16)."""
    phi = np.linspace(0, 180, res * 2 + 1)[1::2] * np.pi / 180
    theta = np.linspace(0, 360, res + 1)[:-1] * np.pi / 180
    dirs = np.empty((res, res, 3))
    dirs[:, :, 0] = np.sin(phi)[:, None] * np.cos(theta)[None, :]
    dirs[:, :, 1] = np.sin(phi)[:, None] * np.sin(theta)[None, :]
    dirs[:, :, 2] = np.cos(phi)[:, None]
    return dirs


def gen_sph_grid(res=128):
    return torch.from_numpy(_unit_dirs(res).reshape(1, 1, res, res, 3)).float()


def sph_pad(sph_tensor, padding_margin=16):
    """replicate-pad all four sides, then make the theta (last) axis circular
    (spherical_proj.py:21-28; like the reference this assumes a square map)."""
    pm = padding_margin
    out = torch.nn.functional.pad(sph_tensor, (pm, pm, pm, pm), mode='replicate')
    _, _, h, w = out.shape
    out[:, :, :, 0:pm] = out[:, :, :, w - 2 * pm:w - pm]
    out[:, :, :, h - pm:] = out[:, :, :, pm:2 * pm]
    return out


class render_spherical(torch.nn.Module):
    def __init__(self, sph_res=128, z_res=256, fused=None):
        super().__init__()
        self.sph_res = sph_res
        self.z_res = z_res
        self.fused = fused
        self.gen_grid()
        self.calc_stop_prob = CalcStopProb.apply

    def gen_grid(self):
        res, z_res = self.sph_res, self.z_res
        # sample k of ray (i,j) sits at 2*dir*(1 - k/(z_res-1)) in grid_sample coordinates (:51-56)
        alpha = np.linspace(0, 1, z_res).reshape(1, 1, z_res, 1)
        grid = (_unit_dirs(res) * 2)[:, :, np.newaxis, :] * (1 - alpha)
        self.register_buffer('depth_weight', torch.linspace(0, 1, z_res))
        self.register_buffer('grid', torch.from_numpy(grid).float())
        # float64 unit directions for the fused kernel, which regenerates `grid` bit-exactly from
        # them; non-persistent so the state_dict keeps exactly the reference's two buffers
        self.register_buffer('_dirs64', torch.from_numpy(_unit_dirs(res)).contiguous(), persistent=False)

    def _use_fused(self, vox):
        if self.fused is not None:
            return self.fused
        from . import _fused_render
        # the brick kernels take rays of up to 256 samples in float4 groups; other shapes use the op sequence
        return (_fused_render.available() and vox.is_cuda and vox.dtype == torch.float32
                and self.z_res <= 256 and self.z_res % 4 == 0 and self.sph_res * self.sph_res < (1 << 24))

    def forward(self, vox, pre_scale=None, pad=0):
        """vox [N,C,X,Y,Z] -> [N,C,res,res].  Extensions: `pre_scale=s` renders
        clamp(vox * s, 1e-5, 1 - 1e-5) -- the expression GenRe feeds this module
        (depth_pred_with_sph_inpaint.py:124, s = 50) -- without materialising it on the fused path;
        `pad=m` returns sph_pad(map, m) (:126), written directly by the fused kernel."""
        if self._use_fused(vox) and 0 <= 2 * int(pad) <= self.sph_res:
            from . import _fused_render
            return _fused_render.RenderSphericalFused.apply(vox, self._dirs64, self.depth_weight,
                                                            0.0 if pre_scale is None else float(pre_scale), int(pad))
        if pad:
            return sph_pad(self.forward(vox, pre_scale), int(pad))
        if pre_scale is not None:
            vox = torch.clamp(vox * pre_scale, 1e-5, 1 - 1e-5)
        grid = self.grid.expand(vox.shape[0], -1, -1, -1, -1)
        vox = vox.permute(0, 1, 4, 3, 2)
        prob_sph = torch.nn.functional.grid_sample(vox, grid, mode='bilinear', padding_mode='zeros',
                                                   align_corners=True)
        prob_sph = torch.clamp(prob_sph, 1e-5, 1 - 1e-5)
        sph_stop_prob = self.calc_stop_prob(prob_sph)
        exp_depth = torch.matmul(sph_stop_prob, self.depth_weight)
        back_groud_prob = torch.prod(1.0 - prob_sph, dim=4)
        return exp_depth + back_groud_prob
