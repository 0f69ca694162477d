"""GPU parity: cam_bp family -- HIP kernels through the C ABI /
autograd Functions vs the CPU oracle on the same seeded inputs.

Bars: cnt (integer-valued fp32) bit-exact; tdf / gradients within 1e-5 absolute (fp32), the
tolerance north_star states; masks bit-exact."""
import numpy as np
import pytest
import torch

import inputs

pytestmark = pytest.mark.gpu
TOL = 1e-5


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def depth_cases():
    return {
        "sphere": inputs.sphere_depth(),
        "sphere_noise": inputs.sphere_depth(noise_seed=2),
        "random30": inputs.random_depth(seed=5),
        "random_negbg": inputs.random_depth(seed=6, negative_bg=True),
        "empty": np.zeros((1, 1, 256, 256), np.float32),
        "small_odd": inputs.sphere_depth(64, 64, noise_seed=3),
    }


@pytest.mark.parametrize("name", list(depth_cases()))
def test_camera_forward(name, genre, oracle, dev):
    d = depth_cases()[name]
    fl, cd = inputs.cam_params(1)
    tdf_o, cnt_o = oracle.back_projection_forward(d, cd, fl)
    tdf = genre.CameraBackProjection.apply(t(d, dev), t(fl, dev), t(cd, dev), 128)
    lib = genre.CameraBackProjection  # cnt via the plain function
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp.functions.get_surface_mask import get_vox_surface_cnt
    cnt = get_vox_surface_cnt(t(d, dev), t(fl, dev), t(cd, dev), 128)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o), "cnt must be exact"
    diff = np.abs(tdf.cpu().numpy() - tdf_o).max()
    assert diff <= TOL, diff
    # voxels hit by exactly one point have no summation-order freedom: bit-exact
    one = cnt_o == 1
    assert np.array_equal(tdf.cpu().numpy()[one], tdf_o[one])
    assert np.array_equal(tdf.cpu().numpy()[cnt_o == 0], tdf_o[cnt_o == 0])


def _odd_cases():
    rng = np.random.default_rng(17)
    out = []
    for (H, res, flv, cdv) in ((64, 32, 100.0, 2.0), (96, 48, 150.0, 1.5), (100, 50, 200.0, 3.0), (37, 20, 60.0, 0.9)):
        d = rng.uniform(cdv - 0.6, cdv + 0.6, (2, 1, H, H)).astype(np.float32)
        d[rng.random(d.shape) < 0.2] = 0.0
        d[rng.random(d.shape) < 0.05] = -1.0
        out.append((d, np.full((2, 1), flv, np.float32), np.full((2, 1), cdv, np.float32), res))
    rng = np.random.default_rng(23)                     # camera INSIDE the grid: zero-depth pixels land in a voxel
    d = rng.uniform(0.0, 0.8, (1, 1, 32, 32)).astype(np.float32)
    d[rng.random(d.shape) < 0.3] = 0.0
    out.append((d, np.full((1, 1), 40.0, np.float32), np.full((1, 1), 0.3, np.float32), 16))
    return out


def check_forward_cases(oracle, dev, exact, deterministic=False):
    """shared by the in-process test (batch-size default: the single-launch brick kernel for these 1- and 2-image
    cases) and the GENRE_CAMBP_MODE=scatter|brick|gather subprocesses: cnt exact always; tdf bit-exact on every voxel
    when `exact`, else <= 1e-5 and bit-exact where a voxel has one point; `deterministic`: two runs agree bit for bit"""
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    fl1, cd1 = inputs.cam_params(1)
    cases = [(d, fl1, cd1, 128) for d in depth_cases().values()] + _odd_cases()
    for d, fl, cd, res in cases:
        tdf_o, cnt_o = oracle.back_projection_forward(d, cd, fl, res)
        runs = []
        for _ in range(2):
            tdf = torch.empty((d.shape[0], 1, res, res, res), device=dev)
            cnt = torch.empty_like(tdf)
            cam_bp_lib.back_projection_forward(t(d, dev), t(cd, dev), t(fl, dev), tdf, cnt)
            runs.append((tdf.cpu().numpy(), cnt.cpu().numpy()))
        tdf, cnt = runs[0]
        assert np.array_equal(cnt, cnt_o), (d.shape, res)
        if exact:
            assert np.array_equal(tdf, tdf_o), (d.shape, res)
            assert np.array_equal(tdf, runs[1][0]) and np.array_equal(cnt, runs[1][1])     # deterministic
        else:
            assert np.abs(tdf - tdf_o).max() <= TOL
            assert np.array_equal(tdf[cnt_o <= 1], tdf_o[cnt_o <= 1])
            if deterministic:
                assert np.array_equal(tdf, runs[1][0]) and np.array_equal(cnt, runs[1][1])


def test_camera_forward_odd_sizes_and_cameras(genre, oracle, dev):
    """non-power-of-two grids / images, several cameras, camera inside the grid (default path: brick kernel at these
    batch sizes)"""
    check_forward_cases(oracle, dev, exact=False)


def _run_cases_in_subprocess(mode, exact, deterministic=False):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import torch\n"
        "from oracle.oracle import Oracle\n"
        "import genre_shapehd_amd\n"
        "import test_gpu_cam_bp as T\n"
        "T.check_forward_cases(Oracle(), torch.device('cuda:0'), exact=%r, deterministic=%r)\n"
        "print('ok')\n" % (root, os.path.join(root, "tests"), exact, deterministic))
    env = dict(os.environ, GENRE_CAMBP_MODE=mode)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])


def test_camera_forward_gather_bit_exact_subprocess(dev):
    """GENRE_CAMBP_MODE=gather: the single-launch gather kernel sums a voxel's points in the reference's serial
    pixel order -- tdf must equal the CPU oracle BIT FOR BIT on every voxel and repeat exactly"""
    _run_cases_in_subprocess("gather", True)


def test_camera_forward_scatter_pinned_subprocess(dev):
    """GENRE_CAMBP_MODE=scatter: fill + scatter + normalise on the 1- and 2-image cases the default now gives to the
    brick kernel"""
    _run_cases_in_subprocess("scatter", False)


def test_camera_forward_brick_pinned_subprocess(dev):
    """GENRE_CAMBP_MODE=brick: the single-launch LDS-brick kernel, pinned; its fp64 LDS sums make it deterministic"""
    _run_cases_in_subprocess("brick", False, deterministic=True)


def test_camera_forward_is_idempotent_on_dirty_outputs(genre, oracle, dev):
    """outputs are fully defined by the call (no dependence on what the buffers held)"""
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    d = inputs.sphere_depth(noise_seed=2)
    fl, cd = inputs.cam_params(1)
    tdf_o, cnt_o = oracle.back_projection_forward(d, cd, fl)
    tdf = torch.full((1, 1, 128, 128, 128), -7.0, device=dev)
    cnt = torch.full((1, 1, 128, 128, 128), 3.0, device=dev)
    for _ in range(2):
        cam_bp_lib.back_projection_forward(t(d, dev), t(cd, dev), t(fl, dev), tdf, cnt)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o)
    assert np.abs(tdf.cpu().numpy() - tdf_o).max() <= TOL


def test_camera_forward_batch_and_channels(genre, oracle, dev):
    d = inputs.batch_depth(3).reshape(3, 1, 256, 256)
    d = np.concatenate([d, d[::-1]], 1)                      # NC = 2
    fl = np.array([[418.3, 400.0], [430.0, 418.3], [418.3, 410.0]], np.float32)
    cd = np.array([[2.2, 2.1], [2.3, 2.2], [2.2, 2.25]], np.float32)
    tdf_o, cnt_o = oracle.back_projection_forward(d, cd, fl)
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp.functions.get_surface_mask import get_vox_surface_cnt
    tdf = genre.CameraBackProjection.apply(t(d, dev), t(fl, dev), t(cd, dev), 128)
    cnt = get_vox_surface_cnt(t(d, dev), t(fl, dev), t(cd, dev), 128)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o)
    assert np.abs(tdf.cpu().numpy() - tdf_o).max() <= TOL


def test_camera_forward_strided_inputs(genre, oracle, dev):
    """the reference kernels are stride-generic (back_projection_kernel.cu:650-672)"""
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    d = inputs.sphere_depth(noise_seed=2)
    fl, cd = inputs.cam_params(1)
    tdf_o, cnt_o = oracle.back_projection_forward(d, cd, fl)
    big = torch.zeros((1, 1, 256, 512), device=dev)
    big[..., ::2] = t(d, dev)
    dv = big[..., ::2]                                          # w stride 2
    assert not dv.is_contiguous()
    tdf = torch.empty((1, 1, 128, 128, 256), device=dev)[..., ::2]   # strided outputs too
    cnt = torch.empty((1, 1, 128, 128, 256), device=dev)[..., ::2]
    cam_bp_lib.back_projection_forward(dv, t(cd, dev), t(fl, dev), tdf, cnt)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o)
    assert np.abs(tdf.cpu().numpy() - tdf_o).max() <= TOL


@pytest.mark.parametrize("name", ["sphere", "sphere_noise", "random30"])
def test_camera_backward(name, genre, oracle, dev):
    d = depth_cases()[name]
    fl, cd = inputs.cam_params(1)
    _, cnt_o = oracle.back_projection_forward(d, cd, fl)
    g = np.random.default_rng(4).standard_normal(cnt_o.shape).astype(np.float32)
    gd_o, gc_o, gf_o, gc_d, gf_d = oracle.back_projection_backward(d, fl, cd, cnt_o, g, with_double=True)
    dt = t(d, dev).requires_grad_(True)
    flt = t(fl, dev).requires_grad_(True)
    cdt = t(cd, dev).requires_grad_(True)
    tdf = genre.CameraBackProjection.apply(dt, flt, cdt, 128)
    tdf.backward(t(g, dev))
    assert np.abs(dt.grad.cpu().numpy() - gd_o).max() <= TOL
    # per-pixel terms are bit-identical; only the summation differs.  The reference adds ~19k fp32
    # terms serially (its own rounding noise ~1e-5 relative); we reduce in fp64.  Check against the
    # fp64-accumulated oracle tightly and against the fp32 serial oracle at its own noise level.
    # (<= 256 per-block fp64 partial sums are combined with fp32 atomics: ~1e-6 relative, order-dependent)
    assert abs(flt.grad.item() - gf_d.item()) <= 5e-6 * max(1.0, abs(gf_d.item()))
    assert abs(cdt.grad.item() - gc_d.item()) <= 5e-6 * max(1.0, abs(gc_d.item()))
    assert abs(flt.grad.item() - gf_o.item()) <= 1e-4 * max(1.0, abs(gf_o.item()))
    assert abs(cdt.grad.item() - gc_o.item()) <= 1e-4 * max(1.0, abs(gc_o.item()))


def test_camera_backward_batch(genre, oracle, dev):
    """bwd at N>1 is defined per sample (the reference reads camdist out of bounds for n>0, F8)"""
    d = inputs.batch_depth(3).reshape(3, 1, 256, 256)
    fl = np.array([[418.3], [430.0], [400.0]], np.float32)
    cd = np.array([[2.2], [2.3], [2.1]], np.float32)
    _, cnt_o = oracle.back_projection_forward(d, cd, fl)
    g = np.random.default_rng(4).standard_normal(cnt_o.shape).astype(np.float32)
    dt = t(d, dev).requires_grad_(True)
    flt = t(fl, dev).requires_grad_(True)
    cdt = t(cd, dev).requires_grad_(True)
    genre.CameraBackProjection.apply(dt, flt, cdt, 128).backward(t(g, dev))
    for i in range(3):
        gd_o, gc_o, gf_o, gc_d, gf_d = oracle.back_projection_backward(
            d[i:i + 1], fl[i:i + 1], cd[i:i + 1], cnt_o[i:i + 1], g[i:i + 1], with_double=True)
        assert np.abs(dt.grad[i:i + 1].cpu().numpy() - gd_o).max() <= TOL
        assert abs(flt.grad[i].item() - gf_d.item()) <= 5e-6 * max(1.0, abs(gf_d.item()))
        assert abs(cdt.grad[i].item() - gc_d.item()) <= 5e-6 * max(1.0, abs(gc_d.item()))


@pytest.mark.parametrize("name", ["sphere", "sphere_noise", "random_negbg"])
def test_surface_mask(name, genre, oracle, dev):
    d = depth_cases()[name]
    fl, cd = inputs.cam_params(1, fl=784.4645406, cam_dist=2.0)     # get_surface_mask.py:25 defaults
    _, cnt_o = oracle.back_projection_forward(d, cd, fl)
    mask_o = oracle.get_surface_mask(d, cd, fl, cnt_o)
    surf, mask = genre.get_surface_mask(t(d, dev))
    assert np.array_equal(surf.cpu().numpy(), np.clip(cnt_o, 0, 1))
    assert np.array_equal(mask.cpu().numpy(), mask_o)


def test_layer_shift(genre, oracle, dev):
    """the layer folds shift_tdf (1 - 128*tdf) into the native op; it must equal the unfused
    composition bit for bit where the sum order is unique, and its gradient must be -128 x the plain one"""
    d = inputs.batch_depth(2)
    fl, cd = inputs.cam_params(2)
    tdf_o, cnt_o = oracle.back_projection_forward(d, cd, fl)
    layer = genre.Camera_back_projection_layer().to(dev)
    dt = t(d, dev).requires_grad_(True)
    out = layer(dt)                                              # fl=418.3, cam_dist=2.2, shift
    ref = (1 - np.float32(128) * tdf_o).astype(np.float32)
    assert np.abs(out.detach().cpu().numpy() - ref).max() <= 128 * TOL
    uniq = cnt_o <= 1
    assert np.array_equal(out.detach().cpu().numpy()[uniq], ref[uniq])
    assert np.array_equal(layer.shift_tdf(torch.from_numpy(tdf_o)).numpy(), ref)
    g = np.random.default_rng(4).standard_normal(tdf_o.shape).astype(np.float32)
    out.backward(t(g, dev))
    for i in range(2):
        gd_o, _, _ = oracle.back_projection_backward(d[i:i + 1], fl[i:i + 1], cd[i:i + 1], cnt_o[i:i + 1],
                                                     (np.float32(-128) * g[i:i + 1]).astype(np.float32))
        assert np.abs(dt.grad[i:i + 1].cpu().numpy() - gd_o).max() <= 128 * TOL
    out2 = layer(t(d, dev), shift=False)
    assert np.abs(out2.cpu().numpy() - tdf_o).max() <= TOL


# ---- spherical back-projection -------------------------------------------------------------
@pytest.mark.parametrize("batch", [1, 3])
def test_spherical_forward_backward(batch, genre, oracle, dev):
    s = np.concatenate([inputs.sph_depth_map(seed=7 + i) for i in range(batch)])
    g = inputs.gen_sph_grid_np()
    gb = np.broadcast_to(g, (batch, 1, 128, 128, 3))
    tdf_o, cnt_o = oracle.spherical_back_proj_forward(s, gb)
    gi = np.random.default_rng(4).standard_normal(tdf_o.shape).astype(np.float32)
    gd_o = oracle.spherical_back_proj_backward(s, gb, cnt_o, gi)
    st = t(s, dev).requires_grad_(True)
    grid = genre.gen_sph_grid(128).to(dev).expand(batch, -1, -1, -1, -1)      # batch stride 0
    assert np.array_equal(grid[0].cpu().numpy(), g[0]), "gen_sph_grid must match the reference table"
    tdf, cnt = genre.SphericalBackProjection.apply(st, grid, 128)
    assert np.array_equal(cnt.cpu().numpy(), cnt_o)
    assert np.abs(tdf.detach().cpu().numpy() - tdf_o).max() <= TOL
    one = cnt_o == 1
    assert np.array_equal(tdf.detach().cpu().numpy()[one], tdf_o[one])
    tdf.backward(t(gi, dev))
    diff = np.abs(st.grad.cpu().numpy() - gd_o)
    scale = np.maximum(1.0, np.abs(gd_o))
    assert (diff / scale).max() <= TOL, (diff / scale).max()


def test_shape_errors_raise(genre, dev):
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    d = torch.zeros((1, 1, 8, 8), device=dev)
    p = torch.zeros((1, 1), device=dev)
    v = torch.zeros((1, 1, 4, 4, 4), device=dev)
    with pytest.raises(RuntimeError, match="cnt"):
        cam_bp_lib.back_projection_forward(d, p, p, v, torch.zeros((1, 1, 4, 4, 5), device=dev))
    with pytest.raises(RuntimeError, match="camdist"):
        cam_bp_lib.back_projection_forward(d, torch.zeros((2, 1), device=dev), p, v, v.clone())
    with pytest.raises(RuntimeError, match="no CPU path"):
        cam_bp_lib.back_projection_forward(d.cpu(), p, p, v, v.clone())


def test_point_exactly_on_a_voxel_centre(genre, oracle, dev):
    """a single point whose distance to its voxel's centre is exactly 0 leaves a raw sum of (-)0: the normalise pass
    must still recognise the voxel as un-normalised -- in the shifted modes (1 - res*tdf of the camera layer,
    (-tdf + 1/res)*res of GenRe's spherical glue) such a voxel holds 1, not 0"""
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    res = 128
    c = (64 + 0.5) / res - 0.5                                            # centre of voxel 64: 0.00390625, exact in fp32
    sph = torch.full((1, 1, 4, 4), -1.0, device=dev)                      # every other direction is skipped (d < 0)
    sph[0, 0, 1, 2] = c
    grid = torch.ones((1, 1, 4, 4, 3), device=dev)                        # direction (1,1,1): the point is (c,c,c)
    for shifted in (False, True):
        out = torch.empty((1, 1, res, res, res), device=dev)
        cnt = torch.empty_like(out)
        (cam_bp_lib.spherical_back_proj_forward_shifted if shifted else cam_bp_lib.spherical_back_proj_forward)(
            sph, grid, out, cnt)
        tdf_o, cnt_o = oracle.spherical_back_proj_forward(sph.cpu().numpy(), grid.cpu().numpy(), res)
        assert cnt[0, 0, 64, 64, 64].item() == 1.0 and cnt.sum().item() == 1.0 and np.array_equal(cnt.cpu().numpy(), cnt_o)
        want = ((-tdf_o + 1.0 / res) * res * np.clip(cnt_o, 0, 1)) if shifted else tdf_o
        assert np.array_equal(out.cpu().numpy(), want.astype(np.float32))
        assert out[0, 0, 64, 64, 64].item() == (1.0 if shifted else 0.0)


def test_by_value_camera_entry_equals_the_tensor_entry(genre, oracle, dev):
    """genre_back_projection_forward_const (one focal length / camera distance for every image, passed by value -- what
    Camera_back_projection_layer fills its tensors with, camera_backprojection_module.py:16-21): bit-for-bit the
    tensor entry's output (plain and shifted), parity with the oracle, and the layer takes it when called with floats"""
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    fl1, cd1 = inputs.cam_params(1)
    cases = [(d, fl1, cd1, 128) for d in depth_cases().values()] + _odd_cases() + [(inputs.batch_depth(5), None, None, 128)]
    for d, fl, cd, res in cases:
        n = d.shape[0]
        flv, cdv = (418.3, 2.2) if fl is None else (float(fl[0, 0]), float(cd[0, 0]))
        flt = torch.full((n, 1), flv, device=dev)
        cdt = torch.full((n, 1), cdv, device=dev)
        if res % 4:                      # rows that are not float4-aligned: not the brick kernel's case (leader pass or refusal:
            continue                     # test_image_minor_camera_forward_is_deterministic_and_bit_identical_to_the_serial_reference)
        for shifted in (False, True):
            a, ca = torch.empty((n, 1, res, res, res), device=dev), torch.empty((n, 1, res, res, res), device=dev)
            b, cb = torch.empty_like(a), torch.empty_like(a)
            (cam_bp_lib.back_projection_forward_shifted if shifted else cam_bp_lib.back_projection_forward)(t(d, dev), cdt, flt, a, ca)
            cam_bp_lib.back_projection_forward_const(t(d, dev), cdv, flv, b, cb, shifted=shifted)
            assert torch.equal(a, b) and torch.equal(ca, cb), (d.shape, res, shifted)
        tdf_o, cnt_o = oracle.back_projection_forward(d, np.full((n, 1), cdv, np.float32), np.full((n, 1), flv, np.float32), res)
        assert np.array_equal(cb.cpu().numpy(), cnt_o)
        assert np.abs(b.cpu().numpy() - (1 - res * tdf_o)).max() <= res * TOL
    # the layer: floats -> by-value entry, same values and gradients as with tensors
    layer = genre.Camera_back_projection_layer().to(dev)
    d1 = t(inputs.sphere_depth(noise_seed=2), dev)
    x1, x2 = d1.clone().requires_grad_(True), d1.clone().requires_grad_(True)
    y1 = layer(x1)
    y2 = layer(x2, torch.full((1, 1), 418.3, device=dev), torch.full((1, 1), 2.2, device=dev))
    assert torch.equal(y1, y2)
    g = torch.randn_like(y1)
    y1.backward(g)
    y2.backward(g)
    assert torch.equal(x1.grad, x2.grad)
    # ... and a resolution whose rows are not float4-aligned takes the tensor entry instead of raising
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp.functions.cam_back_projection import ShiftedCameraBackProjection
    flt, cdt = torch.full((1, 1), 418.3, device=dev), torch.full((1, 1), 2.2, device=dev)
    for res in (30, 126):
        ya = ShiftedCameraBackProjection.apply(d1, flt, cdt, res, False, (418.3, 2.2))
        yb = ShiftedCameraBackProjection.apply(d1, flt, cdt, res, False, None)
        # (both are the three-launch path: global float atomics, so multi-hit voxels -- most of them at 30^3 -- agree to
        # rounding, not bit for bit)
        assert ya.shape == (1, 1, res, res, res) and (ya - yb).abs().max().item() <= res * TOL


@pytest.mark.parametrize("shifted", [False, True])
def test_image_minor_camera_forward_is_deterministic_and_bit_identical_to_the_serial_reference(shifted, oracle, dev):
    """cam_leader_kernel (round 5; csrc/cam_bp.hip): volumes WITHOUT contiguous z rows -- the image-minor volumes of the
    batch-minor renderer, or NCXYZ volumes whose rows are not float4-aligned -- with the camera passed by value get fill + a
    leader pass: every pixel sums the distances of the pixels of its (2 HALO + 1)^2 window that share its voxel, in row-major
    order = the reference's serial index order (back_projection_kernel.cu:215-275), and the first contributor writes.  No
    atomics: two runs agree bit for bit, every element is written, cnt AND tdf equal the oracle's serial evaluation bit for bit
    on every voxel, whatever the batch an image travels in (32, 19, 40, 3) and for every resolution whose voxels project to
    <= 4 pixels; cameras beyond that are refused with a message (the layer then passes tensors: the atomics path)."""
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp.functions.cam_back_projection import leader_halo
    from genre_shapehd_amd.toolbox import _fused_render
    assert leader_halo(128, 418.3, 2.2) == 2
    cases = [(inputs.batch_depth(n, seed=70 + n), 418.3, 2.2, 128, True) for n in (32, 19, 40, 3)]
    for d, fl, cd, res in _odd_cases():
        dd = np.concatenate([d] * 5)[:5]
        cases.append((dd, float(fl[0, 0]), float(cd[0, 0]), res, True))          # image-minor ...
        cases.append((d, float(fl[0, 0]), float(cd[0, 0]), res, False))           # ... and plain NCXYZ (rows not float4-aligned)
    served = 0
    for d, flv, cdv, res, bm in cases:
        n = d.shape[0]
        shape = (n, 1, res, res, res)
        halo = leader_halo(res, flv, cdv)
        new = (lambda: _fused_render.empty_batch_minor(shape, torch.float32, dev)) if bm else (lambda: torch.empty(shape, device=dev))
        if not bm and res % 4 == 0:
            continue                                                       # (dense aligned rows: the brick kernel's case)
        if not 0 <= halo <= 4:
            with pytest.raises(RuntimeError, match="by-value"):
                cam_bp_lib.back_projection_forward_const(t(d, dev), cdv, flv, new(), new(), shifted=shifted)
            continue
        runs = []
        for _ in range(2):
            b, cb = new(), new()
            b.fill_(float("nan")), cb.fill_(float("nan"))                  # every element must be written
            cam_bp_lib.back_projection_forward_const(t(d, dev), cdv, flv, b, cb, shifted=shifted)
            runs.append((b.clone(), cb.clone()))
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), (d.shape, res)
        k = min(n, 2)
        tdf_o, cnt_o = oracle.back_projection_forward(d[:k], np.full((k, 1), cdv, np.float32), np.full((k, 1), flv, np.float32), res)
        want = (np.float32(1) + np.float32(-res) * tdf_o) if shifted else tdf_o
        assert np.array_equal(runs[0][1][:k].cpu().numpy(), cnt_o), (d.shape, res)
        assert np.array_equal(runs[0][0][:k].cpu().numpy(), want.astype(np.float32)), (d.shape, res, halo)
        assert res != 128 or cnt_o.max() >= 2                               # multi-hit voxels are part of the claim
        served += 1
    assert served >= 5
    # the layer takes this path for image-minor batches when called with Python floats (its default call)
    layer = __import__("genre_shapehd_amd").Camera_back_projection_layer(batch_minor=True).to(dev)
    d = t(inputs.batch_depth(16, seed=3), dev)
    y1, y2 = layer(d), layer(d)
    assert y1.stride(0) == 1 and torch.equal(y1, y2)
    tdf_o, _ = oracle.back_projection_forward(d[:1].cpu().numpy(), np.full((1, 1), 2.2, np.float32), np.full((1, 1), 418.3, np.float32), 128)
    assert np.array_equal(y1[:1].cpu().numpy(), np.float32(1) + np.float32(-128) * tdf_o)


def test_image_minor_layer_backward_with_the_sparse_count(genre, dev):
    """The layer keeps cnt for its own backward only, which reads it at the voxel of every in-grid pixel and nowhere else; on
    image-minor volumes the leader pass therefore writes cnt only there (half of the fill is not written: csrc/cam_bp.hip,
    `shifted` bit 1).  Forward values and the gradient of a random upstream gradient equal the standard-layout layer's (whose
    cnt is dense): the gradient bit for bit -- same pixels, same voxels, same counts."""
    d = t(inputs.batch_depth(16, seed=5), dev)
    std, bm = genre.Camera_back_projection_layer().to(dev), genre.Camera_back_projection_layer(batch_minor=True).to(dev)
    g = torch.randn((16, 1, 128, 128, 128), device=dev)
    xa, xb = d.clone().requires_grad_(True), d.clone().requires_grad_(True)
    ya, yb = std(xa), bm(xb)
    assert yb.stride(0) == 1 and (ya - yb).abs().max().item() <= 128 * TOL
    ya.backward(g)
    yb.backward(g)
    assert xa.grad.abs().max().item() > 0 and torch.equal(xa.grad, xb.grad)


@pytest.mark.parametrize("batch_minor", [False, True])
def test_layer_values_stay_in_the_range_it_declares(batch_minor, genre, dev, monkeypatch):
    """Camera_back_projection_layer hangs its value range on the volume it returns (toolbox/_fused_render.py: attach_hint --
    an empty voxel holds the fill value, an occupied one 1 - res * tdf >= 1 - sqrt(3)/2 = 0.1339...: tdf is the mean distance of
    the points INSIDE the voxel from its centre), and the fused renderer builds on it (provably_blocked): checked here on depth
    maps that put many points into one voxel and points near voxel corners"""
    from genre_shapehd_amd.toolbox import _fused_render as F
    monkeypatch.setenv("GENRE_LAZY_ZERO_GRAD", "1")
    rng = np.random.default_rng(17)
    n = 32 if batch_minor else 6
    d = inputs.batch_depth(n, seed=23)
    d[1] = rng.uniform(1.75, 2.65, d[1].shape).astype(np.float32)               # noise through the whole cube: multi-hit voxels
    d[2] = np.float32(2.2) + np.float32(0.4) * np.sin(np.linspace(0, 40, 256, dtype=np.float32))[None, None, :] * np.ones((1, 256, 1), np.float32)
    d[3] = np.float32(1.7001)                                                    # a plane a hair behind the cube's near face
    layer = genre.Camera_back_projection_layer(batch_minor=batch_minor).to(dev)
    vol = layer(torch.from_numpy(d).to(dev))
    h = getattr(vol, "_genre_range_hint", None)
    assert h is not None and h[0] == 0.0 and h[1] == F._VMIN_SHIFTED
    occupied = vol[vol != h[0]]
    assert occupied.numel() > 1000
    assert occupied.min().item() >= 1 - np.sqrt(3) / 2 - 1e-5 > h[1] and occupied.max().item() <= 1.0
    assert F.provably_blocked(vol, 50.0) and not F.provably_blocked(vol, 5.0)
