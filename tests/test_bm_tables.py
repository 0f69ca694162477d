"""CPU checks of the batch-minor tile renderer's host side (genre-shapehd_amd/toolbox/_bm_tables.py): the tables, walked by
a plain-numpy emulation of the kernels' data flow (tests/bm_emulation.py), must reproduce the reference chain
(toolbox/spherical_proj.py:62-72 on CPU torch + the C oracle's calc_prob) forward and backward."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import bm_emulation as E

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod():
    # the table builder is pure numpy: load it without importing the package (which needs libgenre_hip.so)
    spec = importlib.util.spec_from_file_location("_bm_tables", os.path.join(ROOT, "genre-shapehd_amd", "toolbox", "_bm_tables.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _field(res, seed):
    rng = np.random.default_rng(seed)
    ax = (np.arange(res) + 0.5) / res - 0.5
    r2 = ax[:, None, None] ** 2 + (ax[None, :, None] - 0.07) ** 2 + (ax[None, None, :] + 0.04) ** 2
    return (0.003 + 0.9 * (r2 < 0.05) + rng.uniform(0, 0.02, (res,) * 3)).astype(np.float32)


@pytest.mark.parametrize("res,sph,zr,pre_scale,split,pull", [(16, 8, 32, 0.0, None, (8, 8, 8)), (20, 10, 48, 0.0, 64, (4, 8, 8)),
                                                              (13, 6, 24, 3.0, 40, (8, 8, 8)), (16, 8, 32, 2.5, None, (4, 8, 8)),
                                                              (21, 12, 40, 0.0, 150, (4, 8, 8))])
def test_tables_reproduce_the_reference_chain(res, sph, zr, pre_scale, split, pull, oracle, monkeypatch):
    from oracle.torch_oracle import RenderSphericalCPU, unit_dirs
    m = _mod()
    monkeypatch.setattr(m, "ROW_ORDER", "xcd" if res == 20 else "heaviest")      # both row orders are exercised
    dw = np.linspace(0, 1, zr).astype(np.float32)
    dw = torch.linspace(0, 1, zr).numpy()
    kw = dict(pull=pull) if split is None else dict(split_f=split, split_b=split, pull=pull)
    t = m.build_bm_tables(res, res, res, unit_dirs(sph), zr, dw, **kw)
    vox = _field(res, res)
    if pre_scale:
        vox = (vox / np.float32(pre_scale) * np.float32(1.3)).astype(np.float32)       # some voxels clamp, some do not
    out, PS, stash, mask = E.forward(m, t, vox, pre_scale)
    vt = torch.from_numpy(vox[None, None]).requires_grad_(True)
    vin = vt if not pre_scale else torch.clamp(vt * pre_scale, 1e-5, 1 - 1e-5)
    ref = RenderSphericalCPU(oracle, sph, zr)(vin)
    assert np.abs(out - ref.detach().numpy().reshape(-1)).max() <= 1e-5
    g = np.random.default_rng(1).standard_normal(sph * sph).astype(np.float32)
    ref.backward(torch.from_numpy(g).reshape(ref.shape))
    grad = E.backward(m, t, vox.shape, PS, stash, mask, g, dw, pre_scale)
    gr = vt.grad[0, 0].numpy()
    assert (np.abs(grad - gr) / np.maximum(1, np.abs(gr))).max() <= 2e-5
    # structure: segments partition the in-volume samples, rows cover all bricks
    assert t["segs"][:, 2].sum() + m.SLOT_PAD == t["rec_f"].shape[0] and t["segs"][:, 2].max() <= m.MAXSEG
    assert not t["rec_f"][-m.SLOT_PAD:].any()
    pk = t["ent"][:, 2]
    assert (((pk >> 6) & 63) - (pk & 63)).sum() + m.REC_PAD == t["rec_b"].shape[0]
    nb = -(-res // m.BX) * -(-res // m.BY) * -(-res // m.BZ)
    pnb = -(-res // pull[0]) * -(-res // pull[1]) * -(-res // pull[2])
    fr, br = t["fwd_rows"], t["bwd_rows"]
    assert set(fr[fr[:, 3] != m.SKIP, 0]) == set(range(nb)) and set(br[br[:, 3] != m.SKIP, 0]) == set(range(pnb))


def test_tables_are_cached_on_disk(tmp_path, monkeypatch):
    """geometry tables depend on the geometry only: built once, then read back from $GENRE_TABLE_CACHE"""
    import genre_shapehd_amd as G
    from genre_shapehd_amd.toolbox import _fused_render as F
    monkeypatch.setenv("GENRE_TABLE_CACHE", str(tmp_path))
    mod = G.render_spherical(sph_res=8, z_res=16, fused=False)
    args = ((16, 1, 12, 12, 12), torch.device("cpu"), mod._dirs64, mod.depth_weight)
    monkeypatch.setattr(F, "_TABLES", {})
    a = F.bm_tables_for(*args)
    files = sorted(p.name for p in tmp_path.iterdir())
    assert len(files) == 1 and files[0].startswith("bm_") and files[0].endswith(".npz")
    monkeypatch.setattr(F, "_TABLES", {})
    calls = []
    monkeypatch.setattr(F._bm_tables_module(), "build_bm_tables", lambda *x, **k: calls.append(1))
    b = F.bm_tables_for(*args)                                           # served from the file: the builder is not called
    assert not calls and set(a) == set(b)
    for k in a:
        assert a[k] == b[k] if isinstance(a[k], int) else torch.equal(a[k], b[k]), k


def test_table_cache_is_keyed_on_the_contents_of_depth_weight(monkeypatch, tmp_path):
    """a depth_weight buffer that is REWRITTEN with the same values (DDP's buffer broadcast, load_state_dict: its
    _version moves) or re-allocated elsewhere maps to the same device tables -- no rebuild, no leak -- and different
    values never alias an old entry; the cache is bounded (toolbox/_fused_render.py: bm_tables_for)"""
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd.toolbox import _fused_render as F
    from oracle.torch_oracle import unit_dirs
    monkeypatch.setenv("GENRE_TABLE_CACHE", str(tmp_path))
    monkeypatch.setattr(F, "_TABLES", {})
    monkeypatch.setattr(F, "_CONTENT", {})
    dirs = torch.from_numpy(unit_dirs(6))
    dw = torch.linspace(0, 1, 24)
    shape = (32, 1, 13, 13, 13)
    t1 = F.bm_tables_for(shape, "cpu", dirs, dw)
    dw.copy_(torch.linspace(0, 1, 24))                                   # same values, new version
    assert F.bm_tables_for(shape, "cpu", dirs, dw) is t1
    assert F.bm_tables_for(shape, "cpu", dirs, dw.clone()) is t1         # same values, another address
    t2 = F.bm_tables_for(shape, "cpu", dirs, dw * 0.5)
    assert t2 is not t1 and len(F._TABLES) == 2
    for k in range(3 * F._MAX_TABLES):
        F.bm_tables_for(shape, "cpu", dirs, dw * (0.9 - 0.01 * k))
    assert len(F._TABLES) <= F._MAX_TABLES
