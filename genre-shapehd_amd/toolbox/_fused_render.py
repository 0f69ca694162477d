"""Fused render_spherical: HIP kernels instead of the reference's
grid_sample -> clamp -> CalcStopProb -> matmul -> prod -> add chain
(toolbox/spherical_proj.py:62-72).  Only `vox` is saved for backward; the backward recomputes
the rays and accumulates the trilinear adjoint brick by brick in LDS (no global atomics), using
a geometry-only sample list built once here."""
import numpy as np
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .calc_prob.calc_prob._ext import _loader

BRICK = 16          # must match kBrick in csrc/sph_render.hip
SPLIT = 512         # bricks with more 16-sample chunks than this are split over several workgroups
_TABLES = {}


def available():
    return _loader().has_symbol("genre_render_spherical_forward")


def build_brick_tables(X, Y, Z, dirs64, z_res, split=SPLIT):
    """Which samples touch which 16^3 voxel brick -- depends only on the geometry.

    dirs64: float64 [R,R,3] unit directions (spherical_proj.py:43-49).  Returns numpy int32
    (brick_table [rows,4] = (brick id, begin, end, mode), heaviest row first, mode 1 = the brick is
    split over several rows; chunk_list [S] with entries (ray << 12) | (k0 << 4) | (len - 1) = up to
    16 consecutive samples k0..k0+len-1 of one ray).  Sample positions and voxel coordinates are computed with exactly
    the fp64/fp32 operation sequence of the kernel (csrc/sph_render.hip: sample_pos, locate),
    so membership is exact."""
    R = dirs64.shape[0]
    assert z_res <= 256 and R * R < (1 << 20)
    d2 = dirs64.reshape(-1, 3).astype(np.float64) * 2.0
    step = 1.0 / (z_res - 1) if z_res > 1 else 0.0
    alpha = np.arange(z_res, dtype=np.float64) * step
    alpha[-1] = 1.0
    a = 1.0 - alpha
    nbx, nby, nbz = -(-X // BRICK), -(-Y // BRICK), -(-Z // BRICK)
    one, two = np.float32(1), np.float32(2)
    axes = []
    anyin = None
    for ax, size in enumerate((X, Y, Z)):
        g = (d2[:, None, ax] * a[None, :]).astype(np.float32)                # [R*R, ZR]
        i0 = np.floor(((g + one) / two) * np.float32(size - 1)).astype(np.int32)
        inside = (i0 >= -1) & (i0 < size)
        anyin = inside if anyin is None else (anyin & inside)
        b0, v0 = i0 >> 4, i0 >= 0
        b1 = (i0 + 1) >> 4
        v1 = (i0 + 1 <= size - 1) & ((b1 != b0) | ~v0)
        axes.append(((b0, v0), (b1, v1)))
    keys = []
    sample_id = np.arange(R * R * z_res, dtype=np.int64).reshape(R * R, z_res)
    for cx in axes[0]:
        for cy in axes[1]:
            for cz in axes[2]:
                m = anyin & cx[1] & cy[1] & cz[1]
                if not m.any():
                    continue
                brick = (cx[0][m].astype(np.int64) * nby + cy[0][m]) * nbz + cz[0][m]
                keys.append((brick << 32) | sample_id[m])
    keys = np.sort(np.concatenate(keys)) if keys else np.zeros((0,), np.int64)
    bricks = (keys >> 32).astype(np.int64)
    sid = keys & 0xFFFFFFFF
    q, k = sid // z_res, sid % z_res
    # runs of consecutive k of one ray inside one brick, cut into chunks of <= 16 samples
    new_run = np.ones(len(keys), bool)
    new_run[1:] = (bricks[1:] != bricks[:-1]) | (q[1:] != q[:-1]) | (k[1:] != k[:-1] + 1)
    run_start = np.maximum.accumulate(np.where(new_run, np.arange(len(keys)), 0))
    pos = np.arange(len(keys)) - run_start
    starts = np.flatnonzero(pos % 16 == 0)
    lens = np.diff(np.append(starts, len(keys)))
    assert lens.min() >= 1 and lens.max() <= 16
    sample_list = ((q[starts] << 12) | (k[starts] << 4) | (lens - 1)).astype(np.uint32).view(np.int32)
    bricks = bricks[starts]
    nb = nbx * nby * nbz
    begin = np.searchsorted(bricks, np.arange(nb), side="left")
    end = np.searchsorted(bricks, np.arange(nb), side="right")
    rows = []
    for b in range(nb):
        n = int(end[b] - begin[b])
        if n <= split:
            rows.append((b, int(begin[b]), int(end[b]), 0))
        else:
            parts = -(-n // split)
            size = -(-n // parts)
            for s0 in range(int(begin[b]), int(end[b]), size):
                rows.append((b, s0, min(s0 + size, int(end[b])), 1))
    rows.sort(key=lambda r: -(r[2] - r[1]))
    return np.asarray(rows, np.int32).reshape(-1, 4), sample_list


def _tables_for(vox, dirs64, z_res):
    key = (tuple(vox.shape[2:]), dirs64.shape[0], z_res, str(vox.device))
    t = _TABLES.get(key)
    if t is None:
        table, samples = build_brick_tables(vox.shape[2], vox.shape[3], vox.shape[4], dirs64.cpu().numpy(), z_res)
        t = (torch.from_numpy(table).to(vox.device), torch.from_numpy(samples).to(vox.device))
        _TABLES[key] = t
    return t


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
        z_res = depth_weight.shape[0]
        grad_vox = torch.empty(vox.shape, dtype=vox.dtype, device=vox.device)
        table, samples = _tables_for(vox, dirs64, z_res)
        rays = vox.shape[0] * vox.shape[1] * dirs64.shape[0] * dirs64.shape[1]
        scratch = torch.empty((rays * z_res + 4,), dtype=torch.float32, device=vox.device)
        lib.render_spherical_backward(vox, dirs64.view(torch.float32), depth_weight, grad_out, grad_vox,
                                      scratch, table, samples)
        return grad_vox, None, None
