"""Randomised-shape parity (seeded, hypothesis-free so the GPU run is deterministic): many small,
ragged problems per op against the oracle -- edge sizes the fixed cases do not hit (rays of 1..700
samples, 1..9 points, odd grids, NC > 1, strides)."""
import numpy as np
import pytest
import torch

import inputs

pytestmark = pytest.mark.gpu


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_calc_prob_random_shapes(genre, oracle, dev):
    from genre_shapehd_amd.toolbox.calc_prob.calc_prob._ext import calc_prob_lib
    rng = np.random.default_rng(2024)
    for _ in range(40):
        shape = (int(rng.integers(1, 3)), int(rng.integers(1, 3)), int(rng.integers(1, 7)), int(rng.integers(1, 7)),
                 int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 128, 255, 256, 257, 260, 511, 512, 700])))
        p = (inputs.binary_prob(shape, seed=int(rng.integers(1 << 30))) if rng.random() < 0.5
             else inputs.uniform_prob(shape, seed=int(rng.integers(1 << 30))))
        s_o = oracle.calc_prob_forward(p)
        g = rng.standard_normal(shape).astype(np.float32)
        gp_o = oracle.calc_prob_backward(p, s_o * g)
        pt = t(p, dev)
        s = torch.empty_like(pt)
        calc_prob_lib.calc_prob_forward(pt, s)
        assert np.abs(s.cpu().numpy() - s_o).max() <= 1e-5, shape
        out = torch.empty_like(pt)
        calc_prob_lib.calc_prob_backward_fused(pt, s, t(g, dev), out)
        rel = np.abs(out.cpu().numpy() - gp_o) / np.maximum(1.0, np.abs(gp_o))
        assert rel.max() <= 1e-5, (shape, rel.max())


def test_nnd_random_shapes(genre, oracle, dev):
    rng = np.random.default_rng(7)
    for _ in range(40):
        b = int(rng.integers(1, 4))
        n = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 63, 64, 65, 127, 128, 129, 500]))
        m = int(rng.choice([1, 2, 3, 4, 5, 9, 31, 32, 33, 64, 100, 257]))
        if rng.random() < 0.3:          # lattice points: many exact ties
            x1 = rng.integers(0, 3, (b, n, 3)).astype(np.float32)
            x2 = rng.integers(0, 3, (b, m, 3)).astype(np.float32)
        else:
            x1 = rng.standard_normal((b, n, 3)).astype(np.float32)
            x2 = rng.standard_normal((b, m, 3)).astype(np.float32)
        d1o, d2o, i1o, i2o = oracle.nnd_forward(x1, x2)
        d1, d2, i1, i2 = genre.nndistance_w_idx(t(x1, dev), t(x2, dev))
        assert np.array_equal(i1.cpu().numpy(), i1o) and np.array_equal(i2.cpu().numpy(), i2o), (b, n, m)
        assert np.array_equal(d1.cpu().numpy(), d1o) and np.array_equal(d2.cpu().numpy(), d2o), (b, n, m)


def test_cam_bp_random_small(genre, oracle, dev):
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    rng = np.random.default_rng(99)
    for _ in range(20):
        N, NC = int(rng.integers(1, 3)), int(rng.integers(1, 3))
        H = int(rng.choice([8, 16, 33, 64]))
        res = int(rng.choice([8, 16, 24, 32]))
        cdv = float(rng.uniform(1.2, 3.0))
        d = rng.uniform(cdv - 0.6, cdv + 0.6, (N, NC, H, H)).astype(np.float32)
        d[rng.random(d.shape) < 0.3] = 0.0
        d[rng.random(d.shape) < 0.1] = -2.0
        fl = rng.uniform(0.4 * H, 2.0 * H, (N, NC)).astype(np.float32)
        cd = np.full((N, NC), cdv, np.float32)
        tdf_o, cnt_o = oracle.back_projection_forward(d, cd, fl, res)
        tdf = torch.empty((N, NC, res, res, res), device=dev)
        cnt = torch.empty_like(tdf)
        cam_bp_lib.back_projection_forward(t(d, dev), t(cd, dev), t(fl, dev), tdf, cnt)
        assert np.array_equal(cnt.cpu().numpy(), cnt_o), (N, NC, H, res)
        assert np.abs(tdf.cpu().numpy() - tdf_o).max() <= 1e-5
        mask_o = oracle.get_surface_mask(d, cd, fl, cnt_o)
        mask = torch.empty_like(cnt)
        cam_bp_lib.get_surface_mask(t(d, dev), t(cd, dev), t(fl, dev), cnt, mask)
        assert np.array_equal(mask.cpu().numpy(), mask_o), (N, NC, H, res)
        g = rng.standard_normal(cnt_o.shape).astype(np.float32)
        gd = torch.empty((N, NC, H, H), device=dev)
        gc = torch.empty((N, NC), device=dev)
        gf = torch.empty((N, NC), device=dev)
        cam_bp_lib.back_projection_backward(t(d, dev), t(fl, dev), t(cd, dev), cnt, t(g, dev), gd, gc, gf)
        for n in range(N):
            for c in range(NC):
                sl = (slice(n, n + 1), slice(c, c + 1))
                r = oracle.back_projection_backward(d[sl], fl[sl], cd[sl], cnt_o[sl], g[sl], with_double=True)
                assert np.abs(gd[sl].cpu().numpy() - r[0]).max() <= 1e-5 * max(1.0, np.abs(r[0]).max())
                assert abs(gf[n, c].item() - r[4].item()) <= 1e-5 * max(1.0, abs(r[4].item()))
                assert abs(gc[n, c].item() - r[3].item()) <= 1e-5 * max(1.0, abs(r[3].item()))


def test_spherical_random_small(genre, oracle, dev):
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    rng = np.random.default_rng(5150)
    for _ in range(12):
        N, R, res = int(rng.integers(1, 4)), int(rng.choice([8, 16, 31])), int(rng.choice([8, 16, 32]))
        s = rng.uniform(-0.1, 0.7, (N, 1, R, R)).astype(np.float32)
        dirs = rng.standard_normal((1, 1, R, R, 3))
        dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
        grid = np.broadcast_to(dirs.astype(np.float32), (N, 1, R, R, 3))
        tdf_o, cnt_o = oracle.spherical_back_proj_forward(s, grid, res)
        gt = t(dirs.astype(np.float32), dev).expand(N, -1, -1, -1, -1)
        tdf = torch.empty((N, 1, res, res, res), device=dev)
        cnt = torch.empty_like(tdf)
        cam_bp_lib.spherical_back_proj_forward(t(s, dev), gt, tdf, cnt)
        assert np.array_equal(cnt.cpu().numpy(), cnt_o)
        assert np.abs(tdf.cpu().numpy() - tdf_o).max() <= 1e-5
        g = rng.standard_normal(cnt_o.shape).astype(np.float32)
        gd_o = oracle.spherical_back_proj_backward(s, grid, cnt_o, g)
        gd = torch.empty((N, 1, R, R), device=dev)
        cam_bp_lib.spherical_back_proj_backward(t(s, dev), gt, cnt, t(g, dev), gd)
        rel = np.abs(gd.cpu().numpy() - gd_o) / np.maximum(1.0, np.abs(gd_o))
        assert rel.max() <= 1e-5, rel.max()
