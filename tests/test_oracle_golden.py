"""The oracle restatement against the committed golden fixtures (tests/golden/hotpath_golden.npz,
produced by tests/golden/make_golden.py from the reference's own host-compiled code).  Runs
anywhere -- this is what carries the parity pin to machines without /root/reference."""
import os

import numpy as np
import pytest

import inputs

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "hotpath_golden.npz"))

CAM = {"sphere": lambda: inputs.sphere_depth(),
       "sphere_noise": lambda: inputs.sphere_depth(noise_seed=2),
       "random30": lambda: inputs.random_depth(seed=5),
       "random_negbg": lambda: inputs.random_depth(seed=6, negative_bg=True)}


def dense(prefix, shape, empty):
    vol = np.full(int(np.prod(shape)), empty, np.float32)
    cnt = np.zeros(int(np.prod(shape)), np.float32)
    nz = GOLD[prefix + "/nz"]
    vol[nz] = GOLD[prefix + "/tdf"]
    cnt[nz] = GOLD[prefix + "/cnt"].astype(np.float32)
    return vol.reshape(shape), cnt.reshape(shape)


@pytest.mark.parametrize("name", list(CAM))
def test_camera(name, oracle):
    d = CAM[name]()
    fl, cd = inputs.cam_params(1)
    tdf, cnt = oracle.back_projection_forward(d, cd, fl)
    gt, gc = dense("cam/" + name, tdf.shape, GOLD[f"cam/{name}/empty_value"][0])
    assert np.array_equal(cnt, gc) and np.array_equal(tdf, gt)
    assert tdf.astype(np.float64).sum() == GOLD[f"cam/{name}/tdf_sum"]
    g = np.random.default_rng(4).standard_normal(cnt.shape).astype(np.float32)
    gd, gcam, gfl = oracle.back_projection_backward(d, fl, cd, cnt, g)
    assert np.array_equal(gd, GOLD[f"cam/{name}/grad_depth"])
    assert np.array_equal(gcam, GOLD[f"cam/{name}/grad_camdist"])
    assert np.array_equal(gfl, GOLD[f"cam/{name}/grad_fl"])


@pytest.mark.parametrize("name", ["sphere_noise", "random_negbg"])
def test_mask(name, oracle):
    d = CAM[name]()
    fl, cd = inputs.cam_params(1, fl=784.4645406, cam_dist=2.0)
    _, cnt = oracle.back_projection_forward(d, cd, fl)
    assert np.array_equal(np.flatnonzero(cnt).astype(np.int32), GOLD[f"mask/{name}/cnt_nz"])
    mask = oracle.get_surface_mask(d, cd, fl, cnt)
    assert np.array_equal(np.packbits(mask.ravel().astype(np.uint8)), GOLD[f"mask/{name}/bits"])


@pytest.mark.parametrize("batch", [1, 2])
def test_spherical(batch, oracle):
    s = np.concatenate([inputs.sph_depth_map(seed=7 + i) for i in range(batch)])
    g = np.broadcast_to(inputs.gen_sph_grid_np(), (batch, 1, 128, 128, 3))
    tdf, cnt = oracle.spherical_back_proj_forward(s, g)
    gt, gc = dense(f"sph/b{batch}", tdf.shape, 0.0)
    assert np.array_equal(cnt, gc) and np.array_equal(tdf, gt)
    gi = np.random.default_rng(4).standard_normal(tdf.shape).astype(np.float32)
    assert np.array_equal(oracle.spherical_back_proj_backward(s, g, cnt, gi), GOLD[f"sph/b{batch}/grad_depth"])


def test_calc_prob(oracle):
    for name, p in (("uniform", inputs.uniform_prob((1, 1, 8, 8, 256))),
                    ("binary", inputs.binary_prob((1, 1, 8, 8, 256))),
                    ("odd37", inputs.uniform_prob((1, 2, 3, 5, 37), seed=14))):
        s = oracle.calc_prob_forward(p)
        assert np.array_equal(s, GOLD[f"cp/{name}/stop"])
        g = np.random.default_rng(9).standard_normal(p.shape).astype(np.float32)
        assert np.array_equal(oracle.calc_prob_backward(p, s * g), GOLD[f"cp/{name}/grad"])


@pytest.mark.parametrize("name,gen", [("uniform_full", "uniform_prob"), ("binary_full", "binary_prob")])
def test_calc_prob_full_size(name, gen, oracle):
    s = oracle.calc_prob_forward(getattr(inputs, gen)())
    assert np.array_equal(s.astype(np.float64).sum(-1).astype(np.float32), GOLD[f"cp/{name}/ray_sums"])
    assert np.array_equal(s[0, 0, ::16, ::16], GOLD[f"cp/{name}/sample"])


@pytest.mark.parametrize("name,cfg", [("cfg1", (1, 2048, 2048, 0, 1)), ("ragged", (3, 777, 1301, 11, 12)),
                                      ("tiny", (2, 1, 5, 21, 22))])
def test_nnd(name, cfg, oracle):
    x1, x2 = inputs.clouds(*cfg)
    d1, d2, i1, i2 = oracle.nnd_forward(x1, x2)
    for key, v in (("d1", d1), ("d2", d2), ("i1", i1), ("i2", i2)):
        assert np.array_equal(v, GOLD[f"nnd/{name}/{key}"]), key
    gd1 = np.random.default_rng(77).standard_normal(d1.shape).astype(np.float32)
    gd2 = np.random.default_rng(78).standard_normal(d2.shape).astype(np.float32)
    g1, g2 = oracle.nnd_backward(x1, x2, gd1, gd2, i1, i2)
    assert np.array_equal(g1, GOLD[f"nnd/{name}/g1"]) and np.array_equal(g2, GOLD[f"nnd/{name}/g2"])


def test_config1_nnd_cpu_reference_path(oracle):
    """BASELINE configs[0]: Chamfer on two random 2048-pt clouds, CPU reference path.
    Cross-check the oracle against an independent numpy float32 argmin."""
    x1, x2 = inputs.clouds()
    d1, d2, i1, i2 = oracle.nnd_forward(x1, x2)
    diff = x1[0][:, None, :] - x2[0][None, :, :]
    sq = diff * diff
    dd = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
    assert np.array_equal(i1[0], dd.argmin(1).astype(np.int32))
    assert np.array_equal(i2[0], dd.argmin(0).astype(np.int32))
    assert np.array_equal(d1[0], dd.min(1)) and np.array_equal(d2[0], dd.min(0))
