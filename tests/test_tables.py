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


@pytest.mark.parametrize("res,sph,zr,pull", [(20, 10, 24, (4, 8, 8)), (33, 8, 40, (8, 8, 8))])
def test_bm_listing_names_every_touching_sample_once_per_brick(res, sph, zr, pull):
    """batch-minor tile renderer (toolbox/_bm_tables.py): a brick's backward entries list exactly the samples with a
    weighted corner inside it, the ownership bits mark exactly those corners, segments partition the rays"""
    from genre_shapehd_amd.toolbox import _bm_tables as B
    mod = G.render_spherical(sph_res=sph, z_res=zr, fused=False)
    dirs = mod._dirs64.numpy()
    t = B.build_bm_tables(res, res, res, dirs, zr, mod.depth_weight.numpy(), split_f=50, split_b=40, pull=pull)
    cells, inside = brute_force(res, res, res, dirs, zr)
    assert np.array_equal(inside, np.arange(zr)[None, :] >= t["kin"][:, None])
    nbr = tuple(-(-res // b) for b in pull)                                  # the backward's (pull) bricks
    bs = pull
    segs = t["segs"]
    # segments: consecutive samples of one ray, every in-volume sample in exactly one, per ray in order
    # column 0 of a segment = its line in the per-segment scratch buffers = its position in RAY order
    ray_of = np.searchsorted(t["ray_ptr"], segs[:, 0], side="right") - 1
    seen = np.zeros(inside.shape, int)
    for q, (_, k0, L, slot0) in zip(ray_of, segs):
        seen[q, k0:k0 + L] += 1
    assert np.array_equal(seen, inside.astype(int))
    assert sorted(segs[:, 0].tolist()) == list(range(segs.shape[0]))
    for q in range(sph * sph):
        ids = t["ray_seg"][t["ray_ptr"][q]:t["ray_ptr"][q + 1]]
        assert (segs[ids, 0] == np.arange(t["ray_ptr"][q], t["ray_ptr"][q + 1])).all() and (np.diff(segs[ids, 1]) > 0).all()
        assert segs[ids, 2].sum() == inside[q].sum()
    # expected (brick, sample, corner) triples: corners inside the volume; the -1 corner of a low-side sample is padding
    expect = {}
    qs, ks = np.nonzero(inside)
    for c in range(8):
        d = (c & 1, (c >> 1) & 1, (c >> 2) & 1)
        xyz = [cells[a][qs, ks] + d[a] for a in range(3)]
        ok = np.ones(len(qs), bool)
        for a in range(3):
            ok &= (xyz[a] >= 0) & (xyz[a] < res)
        bid = ((xyz[0][ok] // bs[0]) * nbr[1] + xyz[1][ok] // bs[1]) * nbr[2] + xyz[2][ok] // bs[2]
        for b_, q_, k_ in zip(bid.tolist(), qs[ok].tolist(), ks[ok].tolist()):
            expect.setdefault((b_, q_ * zr + k_), set()).add(c)
    # the rows of split bricks (flag 1) are the head of the backward's table: bm_zero_shared_kernel stops at the first other row
    flags = np.asarray([B.row_flag(w) for w in t["bwd_rows"][:, 3]])
    assert (np.diff((flags == 1).astype(int)) <= 0).all()
    got = {}
    for b_, e0, e1, shared in t["bwd_rows"]:
        if B.row_flag(shared) == B.SKIP:
            continue
        for s_, slot0_, pk, rs in t["ent"][e0:e1]:
            _, k0, L, slot0 = segs[t["ray_seg"][s_]]                     # s_: the segment's scratch line (ray order)
            q = ray_of[t["ray_seg"][s_]]
            i0, i1 = pk & 63, (pk >> 6) & 63
            assert ((pk >> 12) & 63, (pk >> 18) & 255, slot0_) == (L, k0, slot0)
            assert 0 <= i0 < i1 <= L
            for i in range(i0, i1):
                own = int(t["rec_b"][rs + i - i0, 1])
                key = (int(b_), int(q) * zr + int(k0) + i)
                assert key not in got
                got[key] = own
    assert set(got) == set(expect)
    # ownership bits <-> corners.  A low-side sample (base -1) is re-based on voxel 0: its real corner moves from
    # bit 1 to bit 0 of that axis and the other one carries weight 0 and is not owned
    for key, own in got.items():
        q, k = divmod(key[1], zr)
        low = [cells[a][q, k] == -1 for a in range(3)]
        want = 0
        for c in expect[key]:
            bits = [(c >> a) & 1 for a in range(3)]
            bits = [0 if low[a] else bits[a] for a in range(3)]
            want |= 1 << ((bits[0] | (bits[1] << 1)) + 4 * bits[2])
        assert own == want, (key, own, want)


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


def test_zero_gradient_hint_is_guarded_by_the_tensor_version():
    """toolbox/_fused_render.py: attach_zero_hint / zero_hint_of -- the words a producer hangs on a gradient tensor are honoured
    only as long as nobody wrote to the tensor since (ATen bumps the version; raw C-ABI writes drop the attribute in _loader._call)"""
    import torch
    from genre_shapehd_amd.toolbox import _fused_render as F
    g = torch.zeros(4, 1, 2, 2, 2)
    words = torch.zeros(5, dtype=torch.int32)
    assert F.zero_hint_of(g) is None
    F.attach_zero_hint(g, words, 1, 1, 1)
    h = F.zero_hint_of(g)
    assert h is not None and h[0] is words and h[1:] == (1, 1, 1)
    g.add_(1.0)
    assert F.zero_hint_of(g) is None


def test_provably_blocked_volumes_are_recognised_on_the_host(monkeypatch):
    """toolbox/_fused_render.py: provably_blocked / lazy_zero_grad -- the producer's value range ({fill} u [0.13, 1]) hangs on the
    volume; when both ends land outside clamp(x * pre_scale, 1e-5, 1 - 1e-5) the renderer's gradient is identically zero and is
    returned as a stride-0 view of one zero with the zero-gradient words.  Not without the hint, not after a write, not for a
    pre_scale that lets occupied voxels through, not when switched off."""
    import torch
    from genre_shapehd_amd.toolbox import _fused_render as F
    monkeypatch.setenv("GENRE_LAZY_ZERO_GRAD", "1")
    v = torch.zeros(2, 1, 4, 4, 4)
    assert not F.provably_blocked(v, 50.0)
    F.attach_hint(v, torch.zeros(1, dtype=torch.int32), 128)
    assert F.provably_blocked(v, 50.0) and F.provably_blocked(v, 8.0)
    assert not F.provably_blocked(v, 5.0) and not F.provably_blocked(v, 0.9) and not F.provably_blocked(v, 0.0)
    monkeypatch.setenv("GENRE_LAZY_ZERO_GRAD", "0")
    assert not F.provably_blocked(v, 50.0)
    monkeypatch.setenv("GENRE_LAZY_ZERO_GRAD", "1")
    v.mul_(2.0)
    assert not F.provably_blocked(v, 50.0)
    # the bound itself: an occupied voxel holds 1 - res * (mean distance to its centre of points inside it) >= 1 - sqrt(3)/2
    assert F._VMIN_SHIFTED < 1 - np.sqrt(3) / 2
    g = F.lazy_zero_grad((3, 1, 4, 4, 4), torch.device("cpu"))
    assert g.shape == (3, 1, 4, 4, 4) and g.stride() == (0, 0, 0, 0, 0) and float(g.abs().sum()) == 0.0
    words, stride, off, group = F.zero_hint_of(g)
    assert int(words[(2 // group) * stride + off]) == 0
