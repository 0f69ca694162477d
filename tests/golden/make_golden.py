#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own code (oracle/_ref: the kernel bodies of
back_projection_kernel.cu / calc_prob_kernel.cu / nnd_cuda.cu and my_lib.c as shipped,
host-compiled by oracle/build_ref.py from /root/reference).

Run in the build container only (needs /root/reference):
    python oracle/build_ref.py && python tests/golden/make_golden.py
Inputs are NOT stored: they are regenerated from seeds by tests/inputs.py.  Volumes are stored
sparsely (flat indices of non-empty voxels + values) plus float64 checksums.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs  # noqa: E402
from oracle.oracle import Reference  # noqa: E402

CAM_CASES = {
    "sphere": dict(fn="sphere_depth", kw={}),
    "sphere_noise": dict(fn="sphere_depth", kw=dict(noise_seed=2)),
    "random30": dict(fn="random_depth", kw=dict(seed=5)),
    "random_negbg": dict(fn="random_depth", kw=dict(seed=6, negative_bg=True)),
}


def sparse(vol, cnt):
    nz = np.flatnonzero(cnt)
    return nz.astype(np.int32), cnt.ravel()[nz].astype(np.uint16), vol.ravel()[nz]


def main():
    R = Reference()
    out = {}
    # ---- cam_bp forward / backward / mask ------------------------------------------------
    fl, cd = inputs.cam_params(1)
    for name, spec in CAM_CASES.items():
        d = getattr(inputs, spec["fn"])(**spec["kw"])
        tdf, cnt = R.back_projection_forward(d, cd, fl)
        nz, cv, tv = sparse(tdf, cnt)
        out[f"cam/{name}/nz"] = nz
        out[f"cam/{name}/cnt"] = cv
        out[f"cam/{name}/tdf"] = tv
        out[f"cam/{name}/empty_value"] = tdf.ravel()[np.flatnonzero(cnt == 0)[:1]]
        out[f"cam/{name}/tdf_sum"] = np.array(tdf.astype(np.float64).sum())
        g = np.random.default_rng(4).standard_normal(cnt.shape).astype(np.float32)
        gd, gc, gf = R.back_projection_backward(d, fl, cd, cnt, g)
        out[f"cam/{name}/grad_depth"] = gd
        out[f"cam/{name}/grad_camdist"] = gc
        out[f"cam/{name}/grad_fl"] = gf
    flm, cdm = inputs.cam_params(1, fl=784.4645406, cam_dist=2.0)
    for name in ("sphere_noise", "random_negbg"):
        spec = CAM_CASES[name]
        d = getattr(inputs, spec["fn"])(**spec["kw"])
        _, cnt = R.back_projection_forward(d, cdm, flm)
        mask = R.get_surface_mask(d, cdm, flm, cnt)
        out[f"mask/{name}/bits"] = np.packbits(mask.ravel().astype(np.uint8))
        out[f"mask/{name}/cnt_nz"] = np.flatnonzero(cnt).astype(np.int32)
    # ---- spherical back-projection ----------------------------------------------------------
    g0 = inputs.gen_sph_grid_np()
    for batch in (1, 2):
        s = np.concatenate([inputs.sph_depth_map(seed=7 + i) for i in range(batch)])
        gb = np.broadcast_to(g0, (batch, 1, 128, 128, 3))
        tdf, cnt = R.spherical_back_proj_forward(s, gb)
        nz, cv, tv = sparse(tdf, cnt)
        out[f"sph/b{batch}/nz"] = nz
        out[f"sph/b{batch}/cnt"] = cv
        out[f"sph/b{batch}/tdf"] = tv
        gi = np.random.default_rng(4).standard_normal(tdf.shape).astype(np.float32)
        out[f"sph/b{batch}/grad_depth"] = R.spherical_back_proj_backward(s, gb, cnt, gi)
    # ---- calc_prob ----------------------------------------------------------------------------
    for name, p in (("uniform", inputs.uniform_prob((1, 1, 8, 8, 256))),
                    ("binary", inputs.binary_prob((1, 1, 8, 8, 256))),
                    ("odd37", inputs.uniform_prob((1, 2, 3, 5, 37), seed=14))):
        s = R.calc_prob_forward(p)
        g = np.random.default_rng(9).standard_normal(p.shape).astype(np.float32)
        out[f"cp/{name}/stop"] = s
        out[f"cp/{name}/grad"] = R.calc_prob_backward(p, s * g)
    for name, p in (("uniform_full", inputs.uniform_prob()), ("binary_full", inputs.binary_prob())):
        s = R.calc_prob_forward(p)
        out[f"cp/{name}/ray_sums"] = s.astype(np.float64).sum(-1).astype(np.float32)
        out[f"cp/{name}/sample"] = s[0, 0, ::16, ::16].copy()
    # ---- nndistance: CPU path as shipped (my_lib.c) and the CUDA kernel bodies ------------------
    for name, (b, n, m, s1, s2) in (("cfg1", (1, 2048, 2048, 0, 1)), ("ragged", (3, 777, 1301, 11, 12)),
                                    ("tiny", (2, 1, 5, 21, 22))):
        x1, x2 = inputs.clouds(b, n, m, s1, s2)
        d1, d2, i1, i2 = R.nnd_forward(x1, x2, "cpu")
        k = R.nnd_forward(x1, x2, "cuda")
        assert all(np.array_equal(a, bb) for a, bb in zip((d1, d2, i1, i2), k)), "CPU vs CUDA-body mismatch"
        gd1 = np.random.default_rng(77).standard_normal(d1.shape).astype(np.float32)
        gd2 = np.random.default_rng(78).standard_normal(d2.shape).astype(np.float32)
        g1, g2 = R.nnd_backward(x1, x2, gd1, gd2, i1, i2, "cpu")
        for key, v in (("d1", d1), ("d2", d2), ("i1", i1), ("i2", i2), ("g1", g1), ("g2", g2)):
            out[f"nnd/{name}/{key}"] = v
    path = os.path.join(HERE, "hotpath_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB,", len(out), "arrays")


if __name__ == "__main__":
    main()
