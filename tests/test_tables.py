"""Host logic of the fused renderer: the geometry-only sample lists (toolbox/_fused_render.py) against a brute-force
enumeration with the same position arithmetic restated independently here (spherical_proj.py:39-56 for the sample
positions, ATen's align_corners=True un-normalisation for the cell).  Runs without a GPU."""
import numpy as np
import pytest

import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render as F


def brute_force(X, Y, Z, dirs64, z_res):
    """per sample: base cell (x0,y0,z0) and whether any of its 8 corners lies inside the volume"""
    R = dirs64.shape[0]
    alpha = np.linspace(0, 1, z_res)                                            # spherical_proj.py:52
    grid = ((dirs64 * 2)[:, :, None, :] * (1 - alpha)[None, None, :, None]).astype(np.float32)   # :50-56
    cells = []
    for ax, size in enumerate((X, Y, Z)):
        ix = ((grid[..., ax] + np.float32(1)) / np.float32(2)) * np.float32(size - 1)
        cells.append(np.floor(ix).astype(np.int64))
    inside = np.ones(cells[0].shape, bool)
    for c, size in zip(cells, (X, Y, Z)):
        inside &= (c >= -1) & (c < size)
    return [c.reshape(R * R, z_res) for c in cells], inside.reshape(R * R, z_res)


@pytest.mark.parametrize("res,sph,zr", [(16, 8, 12), (24, 12, 32), (33, 10, 20)])
def test_brick_tables_cover_exactly_the_in_volume_samples(res, sph, zr):
    mod = G.render_spherical(sph_res=sph, z_res=zr, fused=False)
    dirs = mod._dirs64.numpy()
    t = F.build_brick_tables(res, res, res, dirs, zr)
    cells, inside = brute_force(res, res, res, dirs, zr)
    # kin: the in-volume samples of a ray are a suffix starting at kin
    kin = t["kin"]
    assert np.array_equal(inside, np.arange(zr)[None, :] >= kin[:, None])
    # forward list: every in-volume sample exactly once, under the brick of its clamped base corner
    fw = t["fwd_chunks"].view(np.uint32)
    q, k = (fw >> 8).astype(np.int64), (fw & 255).astype(np.int64)
    assert len(fw) == inside.sum() and len(np.unique(q * zr + k)) == len(fw) and inside[q, k].all()
    nb = -(-res // F.BRICK)
    want = ((np.clip(cells[0][q, k], 0, res - 1) >> 4) * nb + (np.clip(cells[1][q, k], 0, res - 1) >> 4)) * nb \
        + (np.clip(cells[2][q, k], 0, res - 1) >> 4)
    rows = t["fwd_table"]
    got = np.empty(len(fw), np.int64)
    covered = np.zeros(len(fw), int)
    for b, beg, end, _ in rows:
        got[beg:end] = b
        covered[beg:end] += 1
    assert (covered == 1).all() and np.array_equal(got, want)
    # backward list: a sample is listed under brick B iff one of its in-volume corners lies in B
    bw = t["bwd_chunks"].view(np.uint32)
    pairs = set()
    for b, beg, end, _ in t["bwd_table"]:
        qq, kk = (bw[beg:end] >> 8).astype(np.int64), (bw[beg:end] & 255).astype(np.int64)
        pairs.update(zip([int(b)] * (end - beg), (qq * zr + kk).tolist()))
    expect = set()
    qs, ks = np.nonzero(inside)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                x, y, z = cells[0][qs, ks] + dx, cells[1][qs, ks] + dy, cells[2][qs, ks] + dz
                ok = (x >= 0) & (x < res) & (y >= 0) & (y < res) & (z >= 0) & (z < res)
                bid = ((x[ok] >> 4) * nb + (y[ok] >> 4)) * nb + (z[ok] >> 4)
                expect.update(zip(bid.tolist(), (qs[ok] * zr + ks[ok]).tolist()))
    assert pairs == expect


def test_subbrick_table_lists_every_touching_sample():
    res, sph, zr = 20, 10, 24
    mod = G.render_spherical(sph_res=sph, z_res=zr, fused=False)
    dirs = mod._dirs64.numpy()
    t = F.build_subbrick_table(res, res, res, dirs, zr, split=64)
    cells, inside = brute_force(res, res, res, dirs, zr)
    ns = -(-res >> F.SUB)
    words = t["sub_list"].view(np.uint32)
    pairs = set()
    seen_sub = set()
    for sb, beg, end, shared in t["sub_rows"]:
        assert end - beg <= 64 or not shared
        seen_sub.add(int(sb))
        qq, kk = (words[beg:end] >> 8).astype(np.int64), (words[beg:end] & 255).astype(np.int64)
        pairs.update(zip([int(sb)] * (end - beg), (qq * zr + kk).tolist()))
    assert seen_sub == set(range(ns ** 3))                                   # every sub-brick has a row (gets written)
    expect = set()
    qs, ks = np.nonzero(inside)
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                x, y, z = cells[0][qs, ks] + dx, cells[1][qs, ks] + dy, cells[2][qs, ks] + dz
                ok = (x >= 0) & (x < res) & (y >= 0) & (y < res) & (z >= 0) & (z < res)
                sid = ((x[ok] >> F.SUB) * ns + (y[ok] >> F.SUB)) * ns + (z[ok] >> F.SUB)
                expect.update(zip(sid.tolist(), (qs[ok] * zr + ks[ok]).tolist()))
    assert pairs == expect


def test_row_splitting_keeps_order_and_flags():
    mod = G.render_spherical(sph_res=12, z_res=32, fused=False)
    t = F.build_brick_tables(32, 32, 32, mod._dirs64.numpy(), 32, split=200, split_fwd=300)
    for table, lim in ((t["bwd_table"], 200), (t["fwd_table"], 300)):
        n = table[:, 2] - table[:, 1]
        assert (n <= lim).all() and (np.diff(n) <= 0).all()                   # heaviest first
    bt = t["bwd_table"]
    counts = np.bincount(bt[:, 0], minlength=8)
    for b in np.nonzero(counts > 1)[0]:
        assert (bt[bt[:, 0] == b, 3] == 1).all()                              # split bricks accumulate with atomics
    for b in np.nonzero(counts == 1)[0]:
        assert (bt[bt[:, 0] == b, 3] == 0).all()
