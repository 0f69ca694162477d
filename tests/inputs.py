"""Seeded synthetic inputs for the hot path.

Pure numpy; used by the parity tests, the golden-vector generator
(tests/golden/make_golden.py), __graft_entry__.smoke() and bench.py.
"""
import numpy as np

FL = 418.3        # camera_backprojection_module.py:12 defaults
CAM_DIST = 2.2
RES = 128


def sphere_depth(H=256, W=256, radius=0.4, fl=FL, cam_dist=CAM_DIST, noise_seed=None,
                 noise_sigma=1e-3, centre=(0.0, 0.0, 0.0)):
    """Ray depth of a sphere seen from (-cam_dist,0,0) looking along +x; background 0
    (what get_abs_depth produces, depth_pred_with_sph_inpaint.py:139).  float32 [1,1,H,W].
    Pixel (h,w) looks along (fl, -(w-(W-1)/2), -(h-(H-1)/2)) (back_projection_kernel.cu:231-242)."""
    h = np.arange(H, dtype=np.float64)[:, None] - (H - 1) / 2.0
    w = np.arange(W, dtype=np.float64)[None, :] - (W - 1) / 2.0
    norm = np.sqrt(h * h + w * w + fl * fl)
    dx, dy, dz = fl / norm, -w / norm, -h / norm
    ox, oy, oz = -cam_dist - centre[0], -centre[1], -centre[2]
    b = dx * ox + dy * oy + dz * oz
    c = ox * ox + oy * oy + oz * oz - radius * radius
    disc = b * b - c
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.0))
    depth = np.where(hit & (t > 0), t, 0.0)
    if noise_seed is not None:
        rng = np.random.default_rng(noise_seed)
        depth = np.where(depth > 0, depth + rng.normal(0.0, noise_sigma, depth.shape), 0.0)
    return depth.astype(np.float32)[None, None]


def random_depth(H=256, W=256, seed=5, lo=1.8, hi=2.6, keep=0.3, negative_bg=False):
    """Uniform depth in [lo,hi] on a random `keep` fraction of pixels; background 0 (or -1):
    stresses out-of-grid rejection and the d<0 skip.  float32 [1,1,H,W]."""
    rng = np.random.default_rng(seed)
    d = rng.uniform(lo, hi, (H, W))
    m = rng.random((H, W)) < keep
    bg = -1.0 if negative_bg else 0.0
    return np.where(m, d, bg).astype(np.float32)[None, None]


def batch_depth(n, H=256, W=256, seed=100):
    """n different noisy spheres (radius/centre jittered per item).  float32 [n,1,H,W]."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        r = 0.25 + 0.2 * rng.random()
        c = tuple((rng.random(3) - 0.5) * 0.2)
        out.append(sphere_depth(H, W, radius=r, centre=c, noise_seed=seed + 1 + i)[0])
    return np.stack(out).astype(np.float32)


def cam_params(n, nc=1, fl=FL, cam_dist=CAM_DIST):
    return (np.full((n, nc), fl, np.float32), np.full((n, nc), cam_dist, np.float32))


def uniform_prob(shape=(1, 1, 128, 128, 256), seed=3):
    return np.random.default_rng(seed).uniform(1e-5, 1 - 1e-5, shape).astype(np.float32)


def binary_prob(shape=(1, 1, 128, 128, 256), seed=6, p_hit=0.02):
    """Near-binary field like clamp(tdf*50): mostly 1e-5, occasional 1-1e-5 and mid values."""
    rng = np.random.default_rng(seed)
    u = rng.random(shape)
    p = np.full(shape, 1e-5, np.float32)
    p[u < p_hit] = np.float32(1 - 1e-5)
    mid = (u >= p_hit) & (u < 2 * p_hit)
    p[mid] = rng.uniform(1e-5, 1 - 1e-5, int(mid.sum())).astype(np.float32)
    return np.clip(p, np.float32(1e-5), np.float32(1 - 1e-5)).astype(np.float32)


def gen_sph_grid_np(res=128):
    """Unit directions of toolbox/spherical_proj.py:6-18 (float64 maths, cast to fp32);
    [1,1,res,res,3]."""
    phi = np.linspace(0, 180, res * 2 + 1)[1::2] * np.pi / 180
    theta = np.linspace(0, 360, res + 1)[:-1] * np.pi / 180
    g = np.zeros((res, res, 3))
    g[:, :, 2] = np.cos(phi)[:, None]
    g[:, :, 0] = np.sin(phi)[:, None] * np.cos(theta)[None, :]
    g[:, :, 1] = np.sin(phi)[:, None] * np.sin(theta)[None, :]
    return g.reshape(1, 1, res, res, 3).astype(np.float32)


def sph_depth_map(res=128, seed=7, radius=0.35, bumps=0.05):
    """A spherical depth map (distance from the origin along each direction), values in
    (0, 0.5) so most points land inside the cube; a few are set negative (skipped by K5)
    and a few > 0.9 (out of grid).  float32 [1,1,res,res]."""
    rng = np.random.default_rng(seed)
    d = radius + bumps * rng.standard_normal((res, res))
    u = rng.random((res, res))
    d[u < 0.02] = -0.1
    d[(u >= 0.02) & (u < 0.04)] = 0.95
    return d.astype(np.float32)[None, None]


def clouds(b=1, n=2048, m=2048, seed1=0, seed2=1):
    """config #1: uniform [0,1)^3 clouds, default_rng(0)/(1)."""
    x1 = np.random.default_rng(seed1).random((b, n, 3), dtype=np.float32)
    x2 = np.random.default_rng(seed2).random((b, m, 3), dtype=np.float32)
    return x1, x2


def genre_offclamp_volumes(oracle, n, seed=40, margin=3e-5, solid_margin=None):
    """GenRe-class near-binary occupancy volumes whose values stay OFF the clamp bounds of render_spherical:
    clamp(shift_tdf(cam_bp(depth)) * 50) (depth_pred_with_sph_inpaint.py:120-124) of n different noisy spheres with
    the empty level lifted to margin*(1+u) and the solid level lowered to 1 - margin*(1+u'), u seeded per voxel.
    Every trilinear sample of such a field lies strictly inside (1e-5, 1-1e-5) (except the few rays that graze the
    zero-padded faces of the cube), so the clamp derivative is continuous there and gradients can be compared --
    while the transmittance still underflows behind the surface and dL/dp spans many orders of magnitude, which
    is what the kernels have to survive.  `solid_margin` lowers the solid level further (1 - solid_margin*(1+u')):
    with 1 - p ~ 3e-5 a ONE-ulp difference in a sampled value (6e-8) changes 1 - p by 2e-3 relative, so two correct
    fp32 implementations that add the eight trilinear terms in a different order already differ by ~1e-3 in the
    gradient; solid_margin = 0.02 keeps the field near-binary but conditions the gradient well enough for a 1e-5
    comparison.  float32 [n,1,128,128,128]."""
    rng = np.random.default_rng(seed)
    d = batch_depth(n, seed=seed + 1)
    fl, cd = cam_params(n)
    out = np.empty((n, 1, RES, RES, RES), np.float32)
    for i in range(n):
        tdf, _ = oracle.back_projection_forward(d[i:i + 1], cd[i:i + 1], fl[i:i + 1])
        occ = np.clip((1 - RES * tdf) * 50, 0.0, 1.0)
        lo = margin * (1 + rng.random(occ.shape))
        hi = 1 - (margin if solid_margin is None else solid_margin) * (1 + rng.random(occ.shape))
        out[i] = (lo + occ * (hi - lo)).astype(np.float32)[0]
    return out
