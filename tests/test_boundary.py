"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/genre_hip.h declares, validation errors surface through genre_last_error() without
touching a GPU, the product has no CPU path and never reaches into oracle/."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "genre-shapehd_amd")
LIB = os.path.join(PKG, "csrc", "libgenre_hip.so")
HDR = os.path.join(ROOT, "include", "genre_hip.h")


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(genre_[a-z_0-9]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(LIB), "run __graft_entry__.build() first"
    lib = C.CDLL(LIB)
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), "libgenre_hip.so does not export %s" % s
    lib.genre_abi_version.restype = C.c_int
    assert lib.genre_abi_version() == 5


def test_every_symbol_cites_the_reference_interface():
    src = open(HDR).read()
    for ref in ("back_projection.c:9-17", "back_projection.c:20-28", "back_projection.c:30-38",
                "back_projection.c:40-48", "back_projection.c:49-57", "calc_prob.c:9-17",
                "calc_prob.c:18-26", "my_lib_cuda.c:9-27", "my_lib_cuda.c:30-54"):
        assert ref in src, ref


def test_validation_errors_do_not_launch(genre):
    """shape/dtype violations return 0 with a message before any HIP call (works without a GPU)"""
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import _loader
    L = _loader()
    T = L.GenreTensor

    def desc(shape, dtype=0, ptr=0x1000):
        d = T()
        d.data, d.ndim, d.dtype = ptr, len(shape), dtype
        st = 1
        for i in reversed(range(len(shape))):
            d.size[i], d.stride[i] = shape[i], st
            st *= shape[i]
        return d

    lib = L._lib
    depth, par = desc((1, 1, 8, 8)), desc((1, 1))
    vox, bad = desc((1, 1, 4, 4, 4)), desc((1, 1, 4, 4, 5))
    rc = lib.genre_back_projection_forward(C.byref(depth), C.byref(par), C.byref(par), C.byref(vox),
                                           C.byref(bad), None)
    assert rc == 0 and b"cnt" in lib.genre_last_error()
    rc = lib.genre_calc_prob_forward(C.byref(vox), C.byref(bad), None)
    assert rc == 0 and b"prob_out" in lib.genre_last_error()
    xyz, xyz_bad = desc((2, 7, 3)), desc((2, 7, 4))
    d1, i1 = desc((2, 7)), desc((2, 7), dtype=1)
    rc = lib.genre_nnd_forward(C.byref(xyz_bad), C.byref(xyz), C.byref(d1), C.byref(d1), C.byref(i1),
                               C.byref(i1), None)
    assert rc == 0 and b"xyz1" in lib.genre_last_error()
    rc = lib.genre_nnd_forward(C.byref(xyz), C.byref(xyz), C.byref(d1), C.byref(d1), C.byref(d1),
                               C.byref(i1), None)          # idx1 passed as fp32
    assert rc == 0 and b"idx1" in lib.genre_last_error()
    # glue and renderer extensions
    m, mm = desc((2, 1, 8, 6)), desc((2, 2))
    lib.genre_abs_depth_forward.argtypes = lib.genre_abs_depth_backward.argtypes = [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    rc = lib.genre_abs_depth_forward(C.byref(m), C.byref(mm), C.byref(m), C.byref(m), 100.0, None)   # out not transposed
    assert rc == 0 and b"transposed" in lib.genre_last_error()
    rc = lib.genre_abs_depth_forward(C.byref(m), C.byref(desc((2, 3))), C.byref(m), C.byref(desc((2, 1, 6, 8))), 100.0, None)
    assert rc == 0 and b"depth_minmax" in lib.genre_last_error()
    rc = lib.genre_abs_depth_forward(C.byref(m), C.byref(mm), C.byref(m), C.byref(desc((2, 1, 6, 8))), 0.0, None)
    assert rc == 0 and b"scale_25d" in lib.genre_last_error()
    lib.genre_render_spherical_forward.argtypes = [C.c_void_p] * 9 + [C.c_float, C.c_void_p]
    dirs, dw = desc((8, 8, 6)), desc((16,))
    rc = lib.genre_render_spherical_forward(C.byref(vox), C.byref(dirs), C.byref(dw), C.byref(desc((1, 1, 11, 11))),
                                            None, None, None, None, None, 0.0, None)    # odd padding
    assert rc == 0 and b"R+2p" in lib.genre_last_error()
    rc = lib.genre_render_spherical_forward(C.byref(vox), C.byref(dirs), C.byref(dw), C.byref(desc((1, 1, 12, 12))),
                                            None, None, None, None, None, 0.0, None)    # padded map without the tables
    assert rc == 0 and b"brick path" in lib.genre_last_error()
    # batch-minor renderer backward: one word per image group behind the clamp masks (since ABI 3; nothing is launched: the
    # call fails validation)
    def bm_vol(n, x, y, z):
        d = T()
        d.data, d.ndim, d.dtype = 0x1000, 5, 0
        for i, (sz, st) in enumerate(zip((n, 1, x, y, z), (1, n * x * y * z, y * z * n, z * n, n))):
            d.size[i], d.stride[i] = sz, st
        return d
    n, R, nseg, S = 32, 4, 40, 300
    gvox, gout = bm_vol(n, 8, 8, 8), desc((n, 1, R, R))
    segs, rptr, rseg, rpre = desc((nseg, 4), 1), desc((R * R + 1,), 1), desc((nseg,), 1), desc((R * R, 4))
    ent, rec, rows = desc((nseg, 4), 1), desc((S, 12), 1), desc((4, 4), 1)
    dwt, ps, tr, stash = desc((16,)), desc((nseg * 64,)), desc((nseg * 64,)), desc((S * 32,))
    lib.genre_render_bm_backward.argtypes = [C.c_void_p] * 14 + [C.c_float, C.c_int, C.c_void_p]
    args = [C.byref(a) for a in (gout, gvox, segs, rptr, rseg, rpre, ent, rec, rows, dwt, ps, tr, stash)]
    rc = lib.genre_render_bm_backward(*args, C.byref(desc((8 * 8 * 8,), 1)), 50.0, 488, None)      # masks without the group word
    assert rc == 0 and b"+ groups" in lib.genre_last_error()
    # ABI 4: the standard-layout renderer's pass words, the occupancy words between the camera forward and the batch-minor
    # forward -- wrong sizes / half-given pairs / the wrong producer are refused before anything is launched or cleared
    tab, chunks, kin = desc((1, 4), 1), desc((64,), 1), desc((64,), 1)
    v16 = desc((1, 1, 16, 16, 16))
    rc = lib.genre_render_spherical_forward(C.byref(v16), C.byref(dirs), C.byref(dw), C.byref(desc((1, 1, 8, 8))),
                                            C.byref(desc((8 * 8 * 16,))), C.byref(tab), C.byref(chunks), C.byref(kin),
                                            C.byref(desc((1,), 1)), 50.0, None)         # live: needs 1 * (1 + 1 brick) words
    assert rc == 0 and b"live" in lib.genre_last_error()
    lib.genre_render_bm_forward.argtypes = [C.c_void_p] * 13 + [C.c_float, C.c_void_p]
    fargs = [C.byref(a) for a in (gvox, gout, segs, rec, rows, rptr, rseg, rpre, ps)]
    rc = lib.genre_render_bm_forward(*fargs, None, None, C.byref(desc((1, 2, 1, 1), 1)), None, 0.0, None)   # words without constants
    assert rc == 0 and b"tile_live" in lib.genre_last_error()
    rc = lib.genre_render_bm_forward(*fargs, None, None, C.byref(desc((1, 2, 1, 1), 1)), C.byref(desc((nseg, 2))), 0.0, None)
    assert rc == 0 and b"ps_empty" in lib.genre_last_error()
    rc = lib.genre_render_bm_forward(*fargs, None, None, C.byref(desc((2, 2, 1, 1), 1)), C.byref(desc((nseg, 4))), 0.0, None)
    assert rc == 0 and b"tile_live" in lib.genre_last_error()                            # one slab per image group
    lib.genre_back_projection_forward_const.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int, C.c_void_p]
    rc = lib.genre_back_projection_forward_const(C.byref(depth), C.byref(vox), C.byref(vox), C.byref(desc((2, 1, 1, 1), 1)),
                                                 2.2, 418.3, 1, None)       # dense rows: the brick kernel's words are per image and cell
    assert rc == 0 and b"tile_live" in lib.genre_last_error() and b"N*NC, 1, 1, 1" in lib.genre_last_error()
    bmv = bm_vol(2, 4, 4, 4)
    rc = lib.genre_back_projection_forward_const(C.byref(desc((2, 1, 8, 8))), C.byref(bmv), C.byref(bmv), None, 0.6, 418.3, 1, None)
    assert rc == 0 and b"by-value" in lib.genre_last_error()                             # image-minor, but voxels project too wide
    # ABI 5: which implementation the by-value entry takes is the library's answer (nothing is mirrored in Python) ...
    lib.genre_cam_forward_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float]
    assert lib.genre_cam_cell() == 80832
    assert lib.genre_cam_forward_plan(C.byref(vox), C.byref(vox), 2.2, 418.3) == 1      # dense, float4 rows: brick kernel
    big_bm = bm_vol(32, 128, 128, 128)
    assert lib.genre_cam_forward_plan(C.byref(big_bm), C.byref(big_bm), 2.2, 418.3) == 2    # image-minor: fill + leader pass
    assert lib.genre_cam_forward_plan(C.byref(bmv), C.byref(bmv), 0.6, 418.3) == 0      # ... too close a camera: tensors
    odd = desc((1, 1, 126, 126, 126))
    assert lib.genre_cam_forward_plan(C.byref(odd), C.byref(odd), 2.2, 418.3) == 2       # rows not float4-aligned
    # ... and the segment forward's arguments: the occupancy pair, its cell grid, the scratch size
    lib.genre_render_seg_forward.argtypes = [C.c_void_p] * 14 + [C.c_float, C.c_int, C.c_void_p]
    srow, sseg, rn = desc((1, 4), 1), desc((10, 4), 1), desc((64,), 1)
    sargs = [C.byref(a) for a in (v16, dirs, dw, desc((1, 1, 8, 8)), srow, sseg, rn, desc((64, 4)), desc((3 * 64, 2)))]
    rc = lib.genre_render_seg_forward(*sargs, C.byref(desc((100,))), None, None, None, None, 0.0, 0, None)
    assert rc == 0 and b"ps_scratch" in lib.genre_last_error()                          # not a multiple of 2 * R*R
    pss = desc((3 * 64 * 2,))
    rc = lib.genre_render_seg_forward(*sargs[:-1], C.byref(desc((2 * 64, 2))), C.byref(pss), None, None, None, None, 0.0, 0, None)
    assert rc == 0 and b"line_w" in lib.genre_last_error()                              # one pair per scratch line
    rc = lib.genre_render_seg_forward(*sargs, C.byref(pss), None, C.byref(desc((1, 2, 2, 1), 1)), None, None, 0.0, 80832, None)
    assert rc == 0 and b"come together" in lib.genre_last_error()
    rc = lib.genre_render_seg_forward(*sargs, C.byref(pss), None, C.byref(desc((1, 2, 2, 2), 1)), C.byref(desc((10, 2))), None,
                                      0.0, 80832, None)
    assert rc == 0 and b"occ must be" in lib.genre_last_error()                         # 16^3 voxels in 8x8x32 cells: [1,2,2,1]
    rc = lib.genre_render_seg_forward(*sargs, C.byref(pss), None, C.byref(desc((1, 2, 2, 1), 1)), C.byref(desc((9, 2))), None,
                                      0.0, 80832, None)
    assert rc == 0 and b"ps_empty" in lib.genre_last_error()
    rc = lib.genre_render_seg_forward(*sargs, C.byref(pss), C.byref(desc((1,), 1)), None, None, None, 50.0, 0, None)
    assert rc == 0 and b"live" in lib.genre_last_error()
    rc = lib.genre_render_seg_forward(*sargs, C.byref(pss), None, None, None, C.byref(desc((64 * 16,))), 50.0, 0, None)
    assert rc == 0 and b"v_scratch with pre_scale" in lib.genre_last_error()            # saved samples need the live words
    # ... and its backward's: the state the forward left, one pair of depth weights per scratch line, the live words with pre_scale
    lib.genre_render_seg_backward.argtypes = [C.c_void_p] * 15 + [C.c_float, C.c_void_p]
    g88, gv16, hal = desc((1, 1, 8, 8)), desc((1, 1, 16, 16, 16)), desc((832,))
    bargs = [C.byref(a) for a in (v16, dirs, dw, g88, gv16, srow, sseg, rn, desc((64, 4)), desc((3 * 64, 2)), pss)]
    rc = lib.genre_render_seg_backward(*bargs, C.byref(pss), C.byref(desc((64 * 16,))), C.byref(hal), None, 0.0, None)
    assert rc == 0 and b"tr_scratch" in lib.genre_last_error()
    trs = desc((3 * 64 * 2 + 1,))                                                       # the lines + one word per image and block
    rc = lib.genre_render_seg_backward(*bargs, C.byref(trs), C.byref(desc((64 * 16 - 1,))), C.byref(hal), None, 0.0, None)
    assert rc == 0 and b"v_scratch" in lib.genre_last_error()                           # 16 floats per segment, segments in groups of 64
    rc = lib.genre_render_seg_backward(*bargs, C.byref(trs), C.byref(desc((64 * 16,))), C.byref(hal), None, 50.0, None)
    assert rc == 0 and b"live" in lib.genre_last_error()                                # which slots hold values
    rc = lib.genre_render_seg_backward(*bargs[:4], C.byref(desc((1, 1, 16, 16, 8))), *bargs[5:], C.byref(trs),
                                       C.byref(desc((64 * 16,))), C.byref(hal), None, 0.0, None)
    assert rc == 0 and b"grad_vox" in lib.genre_last_error()
    rc = lib.genre_render_seg_backward(*bargs, C.byref(trs), C.byref(desc((64 * 16,))), C.byref(desc((831,))), None, 0.0, None)
    assert rc == 0 and b"halo_scratch" in lib.genre_last_error()                        # 832 floats per image and row
    # empty problems succeed without launching anything
    e = desc((0, 1, 8, 8))
    ev = desc((0, 1, 4, 4, 4))
    ep = desc((0, 1))
    assert lib.genre_back_projection_forward(C.byref(e), C.byref(ep), C.byref(ep), C.byref(ev), C.byref(ev), None) == 1


def test_cuda_only_ops_refuse_cpu_tensors(genre):
    """cam_bp, calc_prob and the spherical back-projection are CUDA-only in the reference too (`assert ...is_cuda`,
    cam_back_projection.py:18-20, calc_prob.py:14): no CPU path, no fallback.  nndistance is the one op with a CPU
    entry point in the reference (my_lib.nnd_forward) -- tests/test_nnd_host.py; its CUDA entry still refuses host
    tensors"""
    with pytest.raises(AssertionError):
        genre.CalcStopProb.apply(torch.rand(1, 1, 2, 2, 8))
    with pytest.raises(AssertionError):
        genre.CameraBackProjection.apply(torch.rand(1, 1, 8, 8), torch.ones(1, 1), torch.ones(1, 1), 128)
    from genre_shapehd_amd.toolbox.nndistance._ext import my_lib
    x = torch.rand(1, 5, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        my_lib.nnd_forward_cuda(x, x, torch.empty(1, 5), torch.empty(1, 5), torch.empty(1, 5, dtype=torch.int32),
                                torch.empty(1, 5, dtype=torch.int32))


def test_reference_import_lines_work_unchanged(genre):
    """with genre-shapehd_amd/ (and its toolbox/) on sys.path the reference's own import lines
    resolve (genre_full_model.py:8-10,16; depth_pred_with_sph_inpaint.py:7-8; modules/nnd.py:2)"""
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "from toolbox.cam_bp.cam_bp.functions import CameraBackProjection, get_surface_mask, SphericalBackProjection\n"
        "from toolbox.cam_bp.cam_bp.modules.camera_backprojection_module import Camera_back_projection_layer\n"
        "from toolbox.calc_prob.calc_prob.functions.calc_prob import CalcStopProb\n"
        "from toolbox.spherical_proj import render_spherical, sph_pad, gen_sph_grid\n"
        "from nndistance.modules.nnd import NNDModule\n"
        "from nndistance.functions.nnd import nndistance, nndistance_w_idx, nndistance_score\n"
        "from nndistance.functions import nndistance as n2, nndistance_w_idx as n3, nndistance_score as n4\n"
        "from toolbox.calc_prob.calc_prob.functions import CalcStopProb as c2\n"
        "assert n2 is nndistance and c2 is CalcStopProb\n"
        "r = render_spherical()\n"
        "assert tuple(r.grid.shape) == (128, 128, 256, 3) and tuple(r.depth_weight.shape) == (256,)\n"
        "assert sorted(r.state_dict()) == ['depth_weight', 'grid']\n"
        "print('ok')\n" % (PKG, os.path.join(PKG, "toolbox")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under genre-shapehd_amd/ may import, link or
    execute it, and the native library must not link the oracle either."""
    pat = re.compile(r"oracle|liboracle|libref_|_ref/")
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not pat.search(text), "%s mentions the oracle" % os.path.join(dirpath, f)
    import subprocess
    ldd = subprocess.run(["ldd", LIB], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "libref" not in ldd


def test_sph_pad_and_grid_match_the_reference_definitions(genre):
    """sph_pad / gen_sph_grid are pure torch/numpy (spherical_proj.py:6-28): check them on CPU
    against an independent restatement"""
    import numpy as np
    import inputs
    assert np.array_equal(genre.gen_sph_grid(128).numpy(), inputs.gen_sph_grid_np(128))
    x = torch.arange(2 * 1 * 128 * 128, dtype=torch.float32).reshape(2, 1, 128, 128)
    y = genre.sph_pad(x, 16)
    assert y.shape == (2, 1, 160, 160)
    core = y[:, :, 16:144, 16:144]
    assert torch.equal(core, x)
    assert torch.equal(y[:, :, 16:144, 0:16], x[:, :, :, 112:128])       # circular in theta
    assert torch.equal(y[:, :, 16:144, 144:160], x[:, :, :, 0:16])
    assert torch.equal(y[:, :, 0:16, 16:144], x[:, :, 0:1, :].expand(-1, -1, 16, -1))   # replicate in phi
