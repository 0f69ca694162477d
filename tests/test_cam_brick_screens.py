"""Host-side property test of the screens of the single-launch camera kernels (csrc/cam_bp.hip: cam_brick_kernel,
cam_gather_kernel).  Both kernels find the points of a voxel brick by looking only at the depth pixels under the brick's
image FOOTPRINT (project_box), drop whole bricks whose footprint depth range cannot reach them (slab_live) and drop
single pixels with a cheap plane-depth test before the reference's arithmetic runs.  All three are conservative
approximations: if any of them ever rejected a pixel whose point the reference puts into the brick, that point would
silently be missing from the volume.  Here the three screens are restated in numpy float32 (same operation order; the
device's 1-ulp rcp / rsqrt are replaced by correctly rounded ones, which is inside the margins the screens carry) and
checked against the reference's per-pixel arithmetic (back_projection_kernel.cu:215-275) for random cameras, image and
grid sizes, including cameras close to and inside the grid.  No GPU needed; the GPU tests then pin the kernels' outputs."""
import numpy as np
import pytest

F = np.float32
QX, QY, QZ = 8, 8, 32           # cam_brick_kernel's brick
GX, GY, GZ = 16, 8, 64          # cam_gather_kernel's brick


def pixel_voxels(d, f, cam_dist, res):
    """the reference's fp32 sequence per pixel: voxel index triple, or -1 where the point is skipped / out of grid"""
    H, W = d.shape
    h = np.arange(H, dtype=F)[:, None]
    w = np.arange(W, dtype=F)[None, :]
    u_h = h - (F(H) - F(1)) / F(2)
    u_w = w - (F(W) - F(1)) / F(2)
    with np.errstate(all="ignore"):
        norm = np.sqrt((u_h * u_h + u_w * u_w + f * f).astype(F)).astype(F)
        cos = (f / norm).astype(F)
        dd = (d * cos).astype(F)
        gy = ((-dd * u_w).astype(F) / f).astype(F)
        gz = ((-dd * u_h).astype(F) / f).astype(F)
        gx = (dd - cam_dist).astype(F)

        def vox(g):
            a = ((g + F(0.5)).astype(F) * F(res)).astype(F)
            t = np.where(np.isfinite(a), a, F(-1)).astype(np.int64)          # (int)a truncates
            return np.where(a < 0, t - 1, t)
        ix, iy, iz = vox(gx), vox(gy), vox(gz)
    ok = (d >= 0) & (ix >= 0) & (ix < res) & (iy >= 0) & (iy < res) & (iz >= 0) & (iz < res)
    return np.where(ok, ix, -1), np.where(ok, iy, -1), np.where(ok, iz, -1)


def project_box(H, W, xlo, xhi, ylo, yhi, zlo, zhi, cam_dist, f, margin):
    """Win of csrc/cam_bp.hip: inclusive pixel window (h0, h1, w0, w1) and amax2"""
    Xn, Xf = F(xlo + cam_dist), F(xhi + cam_dist)
    ch, cw = (F(H) - F(1)) / F(2), (F(W) - F(1)) / F(2)
    if not (Xn > F(1e-3)) or not (f > 0):
        return 0, H - 1, 0, W - 1, F(ch * ch + cw * cw + F(1))
    a, b = F(f * (F(1) / Xn)), F(f * (F(1) / Xf))
    ws = [F(-ylo * a), F(-ylo * b), F(-yhi * a), F(-yhi * b)]
    hs = [F(-zlo * a), F(-zlo * b), F(-zhi * a), F(-zhi * b)]
    uw_lo, uw_hi, uh_lo, uh_hi = min(ws), max(ws), min(hs), max(hs)
    wl, wh = np.ceil(F(uw_lo + cw - margin)), np.floor(F(uw_hi + cw + margin))
    hl, hh = np.ceil(F(uh_lo + ch - margin)), np.floor(F(uh_hi + ch + margin))
    w0, w1 = int(max(wl, 0)), int(min(wh, W - 1))
    h0, h1 = int(max(hl, 0)), int(min(hh, H - 1))
    mw = F(max(abs(uw_lo), abs(uw_hi)) + margin)
    mh = F(max(abs(uh_lo), abs(uh_hi)) + margin)
    return h0, h1, w0, w1, F(mw * mw + mh * mh)


def check_image(d, f, cam_dist, res, brick):
    H, W = d.shape
    f, cam_dist = F(f), F(cam_dist)
    ix, iy, iz = pixel_voxels(d, f, cam_dist, res)
    BX, BY, BZ = brick
    r = F(1) / F(res)
    hit = ix >= 0
    keys = np.unique(np.stack([ix[hit] // BX, iy[hit] // BY, iz[hit] // BZ], 1), axis=0)
    hh, ww = np.nonzero(hit)
    pb = np.stack([ix[hit] // BX, iy[hit] // BY, iz[hit] // BZ], 1)
    u_h_all = np.arange(H, dtype=F) - (F(H) - F(1)) / F(2)
    u_w_all = np.arange(W, dtype=F) - (F(W) - F(1)) / F(2)
    for bx, by, bz in keys:
        x0, y0, z0 = bx * BX, by * BY, bz * BZ
        x1, y1, z1 = min(x0 + BX, res), min(y0 + BY, res), min(z0 + BZ, res)
        bxlo, bxhi = F(F(x0) * r - F(0.5)), F(F(x1) * r - F(0.5))
        h0, h1, w0, w1, amax2 = project_box(H, W, bxlo, bxhi, F(F(y0) * r - F(0.5)), F(F(y1) * r - F(0.5)),
                                            F(F(z0) * r - F(0.5)), F(F(z1) * r - F(0.5)), cam_dist, f, F(1.0))
        mine = (pb[:, 0] == bx) & (pb[:, 1] == by) & (pb[:, 2] == bz)
        ph, pw = hh[mine], ww[mine]
        # (1) every pixel whose point lands in the brick lies inside the brick's footprint
        assert (ph >= h0).all() and (ph <= h1).all() and (pw >= w0).all() and (pw <= w1).all(), \
            ("footprint", (bx, by, bz), (h0, h1, w0, w1), ph.min(), ph.max(), pw.min(), pw.max())
        # (2) the footprint's depth range keeps the brick alive (slab_live)
        win = d[h0:h1 + 1, w0:w1 + 1]
        pos = win[win > 0]
        dmin, dmax = (pos.min(), pos.max()) if pos.size else (F(3e38), F(0))
        any_zero = bool(((win <= 0) & ~(win < 0)).any())
        eps = F(1e-4)
        cmin = F(f / np.sqrt(F(f * f + amax2)))
        band_lo, band_hi = F(dmin * cmin - eps), F(dmax + eps)
        exotic = (not f > 0) or (not F(bxlo + cam_dist) > F(1e-3))
        zero_hits = any_zero and F(bxlo + cam_dist - eps) <= 0 and F(bxhi + cam_dist + eps) >= 0
        special = exotic or zero_hits
        live = special or (dmax > 0 and band_hi >= F(bxlo + cam_dist) and band_lo <= F(bxhi + cam_dist))
        assert live, ("slab_live", (bx, by, bz), dmin, dmax, band_lo, band_hi, bxlo + cam_dist, bxhi + cam_dist)
        # (3) the per-pixel plane-depth screen keeps every one of them (skipped for `special` bricks, as in the kernel)
        if not special:
            uh, uw = u_h_all[ph], u_w_all[pw]
            xp = (d[ph, pw] * f * (F(1) / np.sqrt((uh * uh + uw * uw + f * f).astype(F))).astype(F)).astype(F) - cam_dist
            assert ((xp >= F(bxlo - F(1e-5))) & (xp <= F(bxhi + F(1e-5)))).all(), ("screen", (bx, by, bz))
    return int(hit.sum()), len(keys)


CAMERAS = [  # (H, W, res, fl, cam_dist, depth lo, depth hi)
    (256, 256, 128, 418.3, 2.2, 1.7, 2.7),       # the reference's camera (configs[1])
    (256, 256, 128, 784.4645406, 2.0, 1.5, 2.5),  # get_surface_mask.py defaults
    (64, 64, 32, 100.0, 2.0, 1.4, 2.6),
    (96, 80, 48, 150.0, 1.5, 0.9, 2.1),           # non-square image
    (100, 100, 50, 200.0, 3.0, 2.4, 3.6),         # non-power-of-two grid (division path of the centres)
    (37, 37, 20, 60.0, 0.9, 0.3, 1.5),            # camera close to the grid
    (128, 128, 64, 90.0, 0.7, 0.15, 1.3),         # wide field of view, camera just outside
    (32, 32, 16, 40.0, 0.3, 0.0, 0.8),            # camera INSIDE the grid (window = whole image)
    (200, 200, 128, 900.0, 4.0, 3.4, 4.6),        # long lens
]


@pytest.mark.parametrize("brick", [(QX, QY, QZ), (GX, GY, GZ)], ids=["brick_8x8x32", "gather_16x8x64"])
@pytest.mark.parametrize("cam", CAMERAS, ids=[f"{c[0]}x{c[1]}_res{c[2]}_fl{c[3]:g}_cd{c[4]:g}" for c in CAMERAS])
def test_screens_never_drop_a_point(cam, brick):
    H, W, res, fl, cd, lo, hi = cam
    rng = np.random.default_rng(hash(cam) % (2 ** 32))
    total = 0
    for trial in range(3):
        d = rng.uniform(lo, hi, (H, W)).astype(F)
        if trial == 1:                                   # smooth surface + holes + skipped pixels
            yy, xx = np.mgrid[0:H, 0:W]
            d = (0.5 * (lo + hi) + 0.3 * (hi - lo) * np.sin(xx / 7.0) * np.cos(yy / 9.0)).astype(F)
        d[rng.random((H, W)) < 0.2] = 0.0
        d[rng.random((H, W)) < 0.05] = -1.0
        n, nb = check_image(d, fl, cd, res, brick)
        total += n
    assert total > 0, "no point landed in the grid: the case does not test anything"


def test_divmod_px_estimate_is_within_one():
    """divmod_px: r = (int)((float)t * (1.0f / d)) is within one of t // d for t < 2^22 (one correction step suffices)"""
    rng = np.random.default_rng(0)
    t = np.concatenate([rng.integers(0, 1 << 22, 200000), np.arange(0, 4096), (1 << 22) - 1 - np.arange(0, 4096)])
    for dd in (1, 2, 3, 7, 19, 20, 23, 64, 255, 256, 257, 1000, 4095, 65536):
        inv = F(1) / F(dd)
        r = (t.astype(F) * inv).astype(F).astype(np.int64)
        assert (np.abs(r - t // dd) <= 1).all(), dd
        q = t - r * dd
        r2 = np.where(q < 0, r - 1, np.where(q >= dd, r + 1, r))
        assert (r2 == t // dd).all(), dd


def test_power_of_two_centre_is_bit_identical():
    """centre_f / centre_d: (i + 0.5) / R == (i + 0.5) * 2^-k bit for bit when R = 2^k"""
    for k in range(0, 12):
        R = 1 << k
        i = np.arange(R, dtype=F)
        n = i + F(0.5)
        assert np.array_equal((n / F(R)).astype(F), (n * F(2.0 ** -k)).astype(F))
        nd = i.astype(np.float64) + 0.5
        assert np.array_equal(nd / float(R), nd * 2.0 ** -k)
