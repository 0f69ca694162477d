"""Host-side checks of the atomic-free camera forward (csrc/cam_bp.hip: cam_leader_kernel, leader_halo).  No GPU needed.

 1. THE WINDOW BOUND.  The kernel finds the pixels that share a voxel by scanning a (2H+1)^2 pixel window, H = leader_halo(res,
    fl, cam_dist) computed on the host from |du| <= f (dy / x_min + |y|_max dx / x_min^2).  If two pixels of one voxel ever lay
    further apart, a contributor would silently be missing from that voxel's sum.  Here every pair of pixels that the
    reference's fp32 arithmetic (back_projection_kernel.cu:215-275, restated in numpy float32) puts into the same voxel is
    enumerated for adversarial depth maps -- a surface that grazes voxel boundaries, random depths, the image corners -- and
    cameras from the default to the closest the bound admits: no pair exceeds H on either axis, and H is not slack by more than
    one pixel for the default camera.
 2. THE ALGORITHM.  A numpy emulation of the leader pass (window scan in row-major order, fp32 sum from the prefill, first
    contributor writes) reproduces the oracle's serial evaluation bit for bit -- tdf and cnt on every voxel; the GPU test
    test_image_minor_camera_forward_is_deterministic_and_bit_identical_to_the_serial_reference then pins the kernel itself."""
import numpy as np
import pytest

import genre_shapehd_amd  # noqa: F401
from genre_shapehd_amd.toolbox.cam_bp.cam_bp.functions.cam_back_projection import leader_halo
from test_cam_brick_screens import pixel_voxels

F = np.float32


def _depths(H, res, f, cd, seed):
    """depth maps that put many pixels into shared voxels: a tilted plane through the cube, a sphere, random depths"""
    rng = np.random.default_rng(seed)
    h, w = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(H, dtype=np.float64), indexing="ij")
    u_h, u_w = h - (H - 1) / 2, w - (H - 1) / 2
    norm = np.sqrt(u_h ** 2 + u_w ** 2 + f ** 2)
    plane = (cd + 0.3 * u_w / (H / 2) * 0.4 + 0.1 * u_h / (H / 2)) * norm / f            # axial depth cd + tilt -> ray depth
    sphere = np.where(u_h ** 2 + u_w ** 2 < (0.3 * f / cd) ** 2, cd - 0.25, 0.0) * norm / f
    rnd = rng.uniform(cd - 0.5, cd + 0.5, (H, H))
    flat = np.full((H, H), cd - 0.4999) * norm / f                                         # hugging the near face: the widest footprint
    return [a.astype(F) for a in (plane, sphere, rnd, flat)]


@pytest.mark.parametrize("H,res,f,cd", [(256, 128, 418.3, 2.2), (64, 32, 100.0, 2.0), (96, 48, 150.0, 1.5), (100, 50, 200.0, 3.0),
                                        (128, 128, 300.0, 1.2), (64, 16, 80.0, 0.8)])
def test_pixels_of_one_voxel_lie_within_the_window(H, res, f, cd):
    halo = leader_halo(res, f, cd)
    assert halo >= 0
    worst = 0
    for d in _depths(H, res, f, cd, seed=H + res):
        ix, iy, iz = pixel_voxels(d, F(f), F(cd), res)
        key = np.where(ix >= 0, (ix * res + iy) * res + iz, -1).ravel()
        hh, ww = np.divmod(np.arange(H * H), H)
        order = np.argsort(key, kind="stable")
        k, hh, ww = key[order], hh[order], ww[order]
        start = np.flatnonzero(np.r_[True, k[1:] != k[:-1]])
        for a, b in zip(start, np.r_[start[1:], k.size]):
            if k[a] < 0 or b - a < 2:
                continue
            worst = max(worst, hh[a:b].max() - hh[a:b].min(), ww[a:b].max() - ww[a:b].min())
    assert worst <= halo, (worst, halo)
    if (H, res, f, cd) == (256, 128, 418.3, 2.2):
        assert halo == 2 and worst >= 1                                   # the default camera: +-2 pixels, and they are needed


def _leader_pass(d, f, cd, res, halo):
    """numpy emulation of cam_leader_kernel for one image: (tdf, cnt) float32 [res,res,res]"""
    H = d.shape[0]
    ix, iy, iz = pixel_voxels(d, F(f), F(cd), res)
    key = np.where(ix >= 0, (ix * res + iy) * res + iz, -1)
    h = np.arange(H, dtype=F)[:, None]
    w = np.arange(H, dtype=F)[None, :]
    u_h, u_w = h - (F(H) - F(1)) / F(2), w - (F(H) - F(1)) / F(2)
    with np.errstate(all="ignore"):
        norm = np.sqrt((u_h * u_h + u_w * u_w + F(f) * F(f)).astype(F)).astype(F)
        dd = (d * (F(f) / norm).astype(F)).astype(F)
        gx, gy, gz = (dd - F(cd)).astype(F), ((-dd * u_w).astype(F) / F(f)).astype(F), ((-dd * u_h).astype(F) / F(f)).astype(F)
        c = lambda i: (((i.astype(np.float64) + 0.5) / res) - 0.5).astype(F)      # noqa: E731  (res is a power of two in this test)
        a, b, e = (gx - c(ix)).astype(F), (gy - c(iy)).astype(F), (gz - c(iz)).astype(F)
        dist = np.sqrt(((a * a).astype(F) + (b * b).astype(F)).astype(F) + (e * e).astype(F)).astype(F)
    prefill, bias = F(1.0 / res), F(1.0) / F(res)
    tdf = np.full((res, res, res), prefill, F)
    cnt = np.zeros((res, res, res), F)
    for y, x in zip(*np.nonzero(key >= 0)):
        k0 = key[y, x]
        s, n, leader = prefill, F(0), True
        for dy in range(-halo, halo + 1):
            for dx in range(-halo, halo + 1):
                yy, xx = y + dy, x + dx
                if 0 <= yy < H and 0 <= xx < H and key[yy, xx] == k0:
                    if dy < 0 or (dy == 0 and dx < 0):
                        leader = False
                    s = F(s + dist[yy, xx])
                    n = F(n + F(1))
        if leader:
            tdf[ix[y, x], iy[y, x], iz[y, x]] = F(F(s - bias) / n)
            cnt[ix[y, x], iy[y, x], iz[y, x]] = n
    return tdf, cnt


def test_the_leader_pass_reproduces_the_serial_reference_bit_for_bit(oracle):
    H, res, f, cd = 64, 32, 100.0, 2.0
    halo = leader_halo(res, f, cd)
    for d in _depths(H, res, f, cd, seed=3)[:3]:
        tdf, cnt = _leader_pass(d, f, cd, res, halo)
        tdf_o, cnt_o = oracle.back_projection_forward(d[None, None], np.full((1, 1), cd, F), np.full((1, 1), f, F), res)
        assert cnt_o.max() >= 2
        assert np.array_equal(cnt, cnt_o[0, 0])
        assert np.array_equal(tdf, tdf_o[0, 0])
