"""Host logic of the segment forward (toolbox/_seg_tables.py; csrc/sph_render_seg.hip): the tables against a brute-force
enumeration of the samples, and the segment algebra -- (P, S) per segment chained per ray from the closed-form prefix -- against
the reference's op sequence (toolbox/spherical_proj.py:62-72) evaluated per ray in float64.  Runs without a GPU."""
import numpy as np
import pytest
import torch

import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _seg_tables as S
from test_tables import brute_force


@pytest.mark.parametrize("res,sph,zr,max_seg,split", [(16, 8, 12, 4, 64), (24, 12, 32, 16, 128), (33, 10, 20, 5, 64),
                                                      (40, 16, 64, 16, 64)])
def test_segments_partition_the_in_volume_samples(res, sph, zr, max_seg, split):
    mod = G.render_spherical(sph_res=sph, z_res=zr, fused=False)
    dirs = mod._dirs64.numpy()
    t = S.build_seg_tables(res, res, res, dirs, zr, mod.depth_weight.numpy(), max_seg=max_seg, split=split, bwd_split=2 * split)
    cells, inside = brute_force(res, res, res, dirs, zr)
    RR = sph * sph
    assert np.array_equal(inside, np.arange(zr)[None, :] >= t["kin"][:, None])
    segs = t["segs"].astype(np.int64)
    q, k0, L, line = segs[:, 0], segs[:, 1] & 255, segs[:, 1] >> 8, segs[:, 2]
    nb = -(-res // S.BRICK)
    brick = ((segs[:, 3] & 1023) * nb + ((segs[:, 3] >> 10) & 1023)) * nb + (segs[:, 3] >> 20)      # column 3: bx | by << 10 | bz << 20
    assert (L >= 1).all() and (L <= max_seg).all()
    # every in-volume sample in exactly one segment; all samples of a segment have their (clamped) base corner in its brick
    seen = np.zeros((RR, zr), int)
    for i in range(len(segs)):
        ks = np.arange(k0[i], k0[i] + L[i])
        seen[q[i], ks] += 1
        b = [np.clip(cells[ax][q[i], ks], 0, res - 1) // S.BRICK for ax in range(3)]
        assert ((b[0] * nb + b[1]) * nb + b[2] == brick[i]).all()
    assert np.array_equal(seen, inside.astype(int))
    # scratch lines: segment s (sample order) of ray q owns line s * RR + q
    assert len(np.unique(line)) == len(line)
    assert np.array_equal(np.bincount(q, minlength=RR), t["ray_nseg"])
    order = np.lexsort((k0, q))
    s_in_ray = np.concatenate([np.arange(n) for n in t["ray_nseg"]])
    assert np.array_equal(line[order], s_in_ray * RR + q[order])
    assert t["smax"][0] == t["ray_nseg"].max()
    # per scratch line: the depth weights of its segment's first and last sample
    dwn = mod.depth_weight.numpy()
    assert t["line_w"].shape == (t["smax"][0] * RR, 2)
    assert np.array_equal(t["line_w"][line, 0], dwn[k0]) and np.array_equal(t["line_w"][line, 1], dwn[k0 + L - 1])
    # rows: every brick in at least one, every segment in exactly one, lengths non-increasing (lane 0 of a wave is the longest)
    rows = t["seg_rows"]
    assert set(rows[:, 0].tolist()) == set(range(nb ** 3))
    covered = np.zeros(len(segs), int)
    for b, beg, end, packed in rows:
        assert packed == (b // (nb * nb)) | ((b // nb) % nb) << 10 | (b % nb) << 20      # the kernel divides nothing
        covered[beg:end] += 1
        assert (brick[beg:end] == b).all() and end - beg <= -(-split // 64) * 64
        assert (np.diff(L[beg:end]) <= 0).all()
    assert (covered == 1).all()
    # the backward's rows: the same segments in longer pieces, bit 30 of the last column set on every row of a split brick
    brows = t["bwd_rows"]
    covered[:] = 0
    per_brick = np.bincount(brows[:, 0], minlength=nb ** 3)
    for b, beg, end, packed in brows:
        packed = int(packed) & 0xffffffff
        assert (packed & 0x3fffffff) == (b // (nb * nb)) | ((b // nb) % nb) << 10 | (b % nb) << 20
        assert ((packed >> 30) & 1) == (per_brick[b] > 1)
        first = bool(packed >> 31)
        assert first == (per_brick[b] > 1 and beg == brows[brows[:, 0] == b][:, 1].min())      # exactly one first row per split brick
        covered[beg:end] += 1
        assert (brick[beg:end] == b).all()
    assert (covered == 1).all()
    flags = (brows[:, 3].astype(np.int64) >> 30) & 1
    assert (np.diff(flags) <= 0).all()                                   # the rows of split bricks form the head of the table


def test_segment_algebra_reproduces_the_ray_integral():
    res, sph, zr = 24, 12, 32
    mod = G.render_spherical(sph_res=sph, z_res=zr, fused=False)
    dirs = mod._dirs64.numpy()
    dw = mod.depth_weight.numpy()
    t = S.build_seg_tables(res, res, res, dirs, zr, dw, max_seg=5, split=64)
    rng = np.random.default_rng(0)
    vox = torch.from_numpy(rng.uniform(0, 1, (1, 1, res, res, res)).astype(np.float32))
    grid = mod.grid[None]
    p = torch.nn.functional.grid_sample(vox.permute(0, 1, 4, 3, 2), grid, mode="bilinear", padding_mode="zeros",
                                        align_corners=True)                    # spherical_proj.py:63-65
    p = torch.clamp(p, 1e-5, 1 - 1e-5)[0, 0].reshape(sph * sph, zr).double().numpy()
    T = np.cumprod(np.concatenate([np.ones((sph * sph, 1)), 1 - p[:, :-1]], 1), 1)      # transmittance before sample k
    want = (T * p * dw[None, :].astype(np.float64)).sum(1) + np.prod(1 - p, 1)          # :67-71
    # through the tables, as the kernels do it: (P, S) per segment with S relative to the depth weight of the segment's first
    # sample (line_w), chained per ray from the prefix of the samples before the volume
    RR = sph * sph
    ps = np.full((t["smax"][0] * RR, 2), np.nan)
    for q, kl, line, _ in t["segs"].astype(np.int64):
        k0, L = kl & 255, kl >> 8
        Ts, Ss = 1.0, 0.0
        for k in range(k0, k0 + L):
            Ss += Ts * p[q, k] * (float(dw[k]) - float(dw[k0]))
            Ts *= 1 - p[q, k]
        ps[line] = (Ts, Ss)
    got = np.empty(RR)
    R_front = np.empty(RR)
    for q in range(RR):
        Tq, Sq = t["ray_pre"][q]
        n = t["ray_nseg"][q]
        for s in range(n):
            P, Sg = ps[s * RR + q]
            Sq += Tq * (Sg + float(t["line_w"][s * RR + q, 0]) * (1 - P))
            Tq *= P
        got[q] = Sq + Tq
        R = 1.0                                                          # the backward's chain: R in front of every segment
        for s in range(n - 1, -1, -1):
            P, Sg = ps[s * RR + q]
            wf = float(t["line_w"][s * RR + q, 0])
            R = wf + Sg + P * (R - wf)
        R_front[q] = R
    assert np.abs(got - want).max() < 1e-12
    # R in front of the first in-volume sample = (the ray's value - what the samples before the volume contribute) / T there
    assert np.abs(R_front - (want - t["ray_pre"][:, 1]) / t["ray_pre"][:, 0]).max() < 1e-10


def test_fixed_point_bound_of_the_segment_backward():
    """csrc/sph_render_seg.hip: seg_scatter_kernel scales its 64-bit fixed-point tile by a BOUND of the image's |dL/dp| that needs no
    pass over the samples: max over rays of |g T| in front of the ray's first segment, times the span of the depth weights and 1.
    dL/dp_k = g T_k (w_k - R_{k+1}) with T_k <= T in front and R a convex combination of depth weights and 1 -- checked here on
    random rays (any probabilities in the clamp's range, any weights), and so is the difference recurrence the kernel carries:
    d_{k-1} = (w_{k-1} - w_k) + (1 - p_k) d_k"""
    rng = np.random.default_rng(3)
    for trial in range(200):
        n = int(rng.integers(2, 300))
        p = np.clip(rng.uniform(-0.2, 1.2, n) ** rng.choice([1, 3]), 1e-5, 1 - 1e-5)
        w = rng.uniform(-2, 5, n) if trial % 3 else np.linspace(0.5, 1.5, n)
        g, t_pre = rng.standard_normal() * 10 ** rng.uniform(-3, 3), rng.uniform(0.01, 1.0)
        T = t_pre * np.concatenate(([1.0], np.cumprod(1 - p[:-1])))
        R = np.empty(n + 1)
        R[n] = 1.0
        for k in range(n - 1, -1, -1):
            R[k] = p[k] * w[k] + (1 - p[k]) * R[k + 1]
        dldp = g * T * (w - R[1:])
        # finite differences of the ray's value  pre + sum_k T_k p_k w_k + T_end: the formula is the derivative
        def value(pp):
            Tq = t_pre * np.concatenate(([1.0], np.cumprod(1 - pp)))
            return (Tq[:-1] * pp * w).sum() + Tq[-1]
        k = int(rng.integers(0, n))
        e = np.zeros(n); e[k] = 1e-7
        assert abs((value(p + e) - value(p - e)) / 2e-7 * g - dldp[k]) <= 1e-5 * max(1.0, abs(dldp).max())
        span = max(w.max(), 1.0) - min(w.min(), 1.0)
        assert np.abs(dldp).max() <= abs(g * t_pre) * span * (1 + 1e-12)
        d = w[n - 1] - R[n]
        for k in range(n - 1, 0, -1):
            assert abs(d - (w[k] - R[k + 1])) <= 1e-9 * (1 + abs(d))
            d = (w[k - 1] - w[k]) + (1 - p[k]) * d
