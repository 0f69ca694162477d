"""ctypes binding of libgenre_hip.so (the C ABI declared in include/genre_hip.h).

This is the only place the Python host side touches native code.  There is NO
CPU fallback: if the library is missing (not built) the import fails loudly,
and every op refuses non-CUDA tensors.  The three objects exported at the
bottom -- ``cam_bp_lib``, ``calc_prob_lib``, ``my_lib`` -- carry the same
function names and argument orders as the reference's cffi modules
(toolbox/cam_bp/cam_bp/src/back_projection.h:1-5, calc_prob.h:1-2,
my_lib_cuda.h:1-4), so the autograd Functions read like the reference's.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (GENRE_HIP_LIB: an A/B variant of the library, tools/build_variants.sh -- measurement only; the default is the in-tree build)
LIB_PATH = os.environ.get("GENRE_HIP_LIB") or os.path.join(_HERE, "csrc", "libgenre_hip.so")
ABI_VERSION = 5
_MAX_DIMS = 5
_F32, _I32 = 0, 1


class GenreTensor(C.Structure):
    """struct genre_tensor of include/genre_hip.h"""
    _fields_ = [("data", C.c_void_p), ("ndim", C.c_int32), ("dtype", C.c_int32),
                ("size", C.c_int64 * _MAX_DIMS), ("stride", C.c_int64 * _MAX_DIMS)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "genre-shapehd_amd: %s is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C genre-shapehd_amd/csrc` (hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback for these ops." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.genre_abi_version.restype = C.c_int
    lib.genre_last_error.restype = C.c_char_p
    if lib.genre_abi_version() != ABI_VERSION:
        raise ImportError("libgenre_hip.so ABI %d != expected %d -- rebuild" % (lib.genre_abi_version(), ABI_VERSION))
    T, V = C.POINTER(GenreTensor), C.c_void_p
    scalars = {"genre_render_spherical_forward": [C.c_float], "genre_render_spherical_backward": [C.c_float],
               "genre_render_seg_forward": [C.c_float, C.c_int], "genre_render_seg_backward": [C.c_float],
               "genre_render_bm_forward": [C.c_float], "genre_render_bm_backward": [C.c_float, C.c_int],
               "genre_abs_depth_forward": [C.c_float], "genre_abs_depth_backward": [C.c_float],
               "genre_back_projection_forward_const": [C.c_float, C.c_float, C.c_int],
               "genre_back_projection_backward_hinted": [C.c_int64, C.c_int64, C.c_int, C.c_int]}
    for name, nargs in (("genre_back_projection_forward", 5), ("genre_back_projection_backward", 8),
                        ("genre_get_surface_mask", 5), ("genre_back_projection_forward_shifted", 5),
                        ("genre_back_projection_backward_shifted", 8), ("genre_back_projection_forward_const", 4),
                        ("genre_back_projection_backward_hinted", 9),
                        ("genre_spherical_back_proj_forward", 4),
                        ("genre_spherical_back_proj_backward", 5), ("genre_spherical_back_proj_forward_shifted", 4),
                        ("genre_spherical_back_proj_backward_shifted", 5), ("genre_calc_prob_forward", 2),
                        ("genre_calc_prob_backward", 3), ("genre_calc_prob_backward_fused", 4),
                        ("genre_nnd_forward", 6), ("genre_nnd_backward", 8),
                        ("genre_render_spherical_forward", 9), ("genre_render_spherical_backward", 11),
                        ("genre_render_seg_backward", 15),
                        ("genre_render_seg_forward", 14),
                        ("genre_render_bm_forward", 13), ("genre_render_bm_backward", 14),
                        ("genre_abs_depth_forward", 4), ("genre_abs_depth_backward", 4),
                        ("genre_nnd_forward_host", 6), ("genre_nnd_backward_host", 8)):
        fn = getattr(lib, name, None)
        if fn is None:
            raise ImportError("libgenre_hip.so does not export %s -- rebuild (make -C genre-shapehd_amd/csrc)" % name)
        fn.argtypes = [T] * nargs + scalars.get(name, []) + ([] if name.endswith("_host") else [V])
        fn.restype = C.c_int
    lib.genre_cam_forward_plan.argtypes = [T, T, C.c_float, C.c_float]
    lib.genre_cam_forward_plan.restype = C.c_int
    lib.genre_cam_cell.restype = C.c_int
    return lib


_lib = _load()


def _desc(t, what, host=False):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s: expected a torch.Tensor, got %r" % (what, type(t)))
    if host and t.is_cuda:
        raise RuntimeError("%s: a host entry point was handed a tensor on %s" % (what, t.device))
    if not host and not t.is_cuda:
        raise RuntimeError("%s: tensor is on %s; the MI355X ops take CUDA (HIP) tensors only -- "
                           "there is no CPU path" % (what, t.device))
    if t.dtype == torch.float32:
        dt = _F32
    elif t.dtype == torch.int32:
        dt = _I32
    else:
        raise RuntimeError("%s: dtype %s not supported (fp32 data, int32 indices)" % (what, t.dtype))
    if t.dim() > _MAX_DIMS:
        raise RuntimeError("%s: more than %d dims" % (what, _MAX_DIMS))
    d = GenreTensor()
    d.data = t.data_ptr()
    d.ndim = t.dim()
    d.dtype = dt
    for i in range(t.dim()):
        d.size[i] = t.size(i)
        d.stride[i] = t.stride(i)
    return d


_HINTS = ("_genre_brick_hint", "_genre_cell_hint", "_genre_zero_hint")   # words a producer hung on a tensor it wrote (toolbox/_fused_render.py)


def _call(name, *tensors, scalars=(), out=()):
    """Enqueue `name` on torch's current stream of the tensors' device; raise on failure
    (the reference raised via THError("aborting"), back_projection.c:13-15).  `out`: positions of the tensors the op WRITES --
    a raw write through the C ABI does not bump the version counter an occupancy hint is guarded by, so a hinted volume that is
    re-used as an output (the reference's caller-allocates convention, cam_back_projection.py:22-25) loses its hint here."""
    dev = tensors[0].device
    for i in out:
        d = getattr(tensors[i], "__dict__", None)
        if d:
            for a in _HINTS:
                d.pop(a, None)
    descs = []
    for k, t in enumerate(tensors):
        if t is None:                       # optional argument -> NULL
            descs.append(None)
            continue
        if t.device != dev:
            raise RuntimeError("%s: tensors are on different GPUs (%s vs %s)" % (name, dev, t.device))
        descs.append(_desc(t, "%s arg %d" % (name, k)))
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        ok = getattr(_lib, name)(*[None if d is None else C.byref(d) for d in descs], *scalars, C.c_void_p(stream))
    if ok != 1:
        raise RuntimeError("%s failed: %s" % (name, _lib.genre_last_error().decode()))
    return 1


def _call_host(name, *tensors):
    """the reference's CPU entry points (my_lib.nnd_forward / nnd_backward): host tensors, synchronous"""
    descs = [_desc(t, "%s arg %d" % (name, k), host=True) for k, t in enumerate(tensors)]
    if getattr(_lib, name)(*[C.byref(d) for d in descs]) != 1:
        raise RuntimeError("%s failed: %s" % (name, _lib.genre_last_error().decode()))
    return 1


def has_symbol(name):
    return hasattr(_lib, name)


class _CamBpLib:
    """stands in for toolbox/cam_bp/cam_bp/_ext/cam_bp_lib (back_projection.h:1-5)"""

    @staticmethod
    def back_projection_forward(depth, camdist, fl, voxel, cnt):
        return _call("genre_back_projection_forward", depth, camdist, fl, voxel, cnt, out=(3, 4))

    @staticmethod
    def back_projection_forward_const(depth, camdist, fl, voxel, cnt, shifted=False, tile_live=None, sparse_cnt=False):
        """extension: camdist / fl are Python floats (one camera for every image), passed by value.  tile_live (int32
        [groups, nbx, nby, nbz]; leader pass only): receives which bricks of which image group hold anything but the fill value;
        sparse_cnt (leader pass only): cnt is written where a point landed and left undefined elsewhere"""
        return _call("genre_back_projection_forward_const", depth, voxel, cnt, tile_live,
                     scalars=(C.c_float(camdist), C.c_float(fl), C.c_int((1 if shifted else 0) | (2 if sparse_cnt else 0))), out=(1, 2, 3))

    PLAN_NONE, PLAN_BRICK, PLAN_LEADER = 0, 1, 2

    @staticmethod
    def forward_plan(voxel, cnt, camdist, fl):
        """which implementation back_projection_forward_const takes for these outputs and this camera (the library's own
        decision -- nothing is mirrored in Python): PLAN_BRICK (dense NCXYZ; occupancy words per image and cam_cell()),
        PLAN_LEADER (other layouts; words per group of 32 images and renderer brick) or PLAN_NONE (pass tensors instead)"""
        return _lib.genre_cam_forward_plan(C.byref(_desc(voxel, "forward_plan voxel")), C.byref(_desc(cnt, "forward_plan cnt")),
                                           C.c_float(camdist), C.c_float(fl))

    @staticmethod
    def cam_cell():
        code = _lib.genre_cam_cell()
        return code // 10000, (code // 100) % 100, code % 100

    @staticmethod
    def back_projection_backward(depth, fl, camdist, cnt, grad_in, grad_depth, grad_camdist, grad_fl):
        return _call("genre_back_projection_backward", depth, fl, camdist, cnt, grad_in, grad_depth,
                     grad_camdist, grad_fl, out=(5, 6, 7))

    @staticmethod
    def back_projection_forward_shifted(depth, camdist, fl, voxel, cnt):
        """extension: writes 1 - res*tdf (Camera_back_projection_layer.shift_tdf folded in)"""
        return _call("genre_back_projection_forward_shifted", depth, camdist, fl, voxel, cnt, out=(3, 4))

    @staticmethod
    def back_projection_backward_shifted(depth, fl, camdist, cnt, grad_in, grad_depth, grad_camdist, grad_fl):
        return _call("genre_back_projection_backward_shifted", depth, fl, camdist, cnt, grad_in, grad_depth,
                     grad_camdist, grad_fl, out=(5, 6, 7))

    @staticmethod
    def back_projection_backward_hinted(depth, fl, camdist, cnt, grad_in, grad_depth, grad_camdist, grad_fl, zero_words,
                                        word_stride, word_offset, group, shifted=True):
        """back_projection_backward[_shifted] for a grad_in whose producer says which images' gradient is identically zero
        (word [(image // group) * word_stride + word_offset] == 0): their grad_depth is zeros, nothing is read"""
        return _call("genre_back_projection_backward_hinted", depth, fl, camdist, cnt, grad_in, grad_depth, grad_camdist,
                     grad_fl, zero_words, scalars=(C.c_int64(word_stride), C.c_int64(word_offset), C.c_int(group),
                                                   C.c_int(1 if shifted else 0)), out=(5, 6, 7))

    @staticmethod
    def get_surface_mask(depth, camdist, fl, cnt, mask):
        return _call("genre_get_surface_mask", depth, camdist, fl, cnt, mask, out=(4,))

    @staticmethod
    def spherical_back_proj_forward(depth, grid_in, voxel, cnt):
        return _call("genre_spherical_back_proj_forward", depth, grid_in, voxel, cnt, out=(2, 3))

    @staticmethod
    def spherical_back_proj_backward(depth, grid_in, cnt, grad_in, grad_depth):
        return _call("genre_spherical_back_proj_backward", depth, grid_in, cnt, grad_in, grad_depth, out=(4,))


    @staticmethod
    def spherical_back_proj_forward_shifted(depth, grid_in, voxel, cnt):
        """extension: writes (-tdf + 1/res) * res * clamp(cnt,0,1) (genre_full_model.py:139-142 folded in)"""
        return _call("genre_spherical_back_proj_forward_shifted", depth, grid_in, voxel, cnt, out=(2, 3))

    @staticmethod
    def spherical_back_proj_backward_shifted(depth, grid_in, cnt, grad_in, grad_depth):
        return _call("genre_spherical_back_proj_backward_shifted", depth, grid_in, cnt, grad_in, grad_depth, out=(4,))


class _CalcProbLib:
    """stands in for toolbox/calc_prob/calc_prob/_ext/calc_prob_lib (calc_prob.h:1-2)"""

    @staticmethod
    def calc_prob_forward(prob_in, prob_out):
        return _call("genre_calc_prob_forward", prob_in, prob_out, out=(1,))

    @staticmethod
    def calc_prob_backward(prob_in, stop_prob_weighted, grad_out):
        return _call("genre_calc_prob_backward", prob_in, stop_prob_weighted, grad_out, out=(2,))

    @staticmethod
    def calc_prob_backward_fused(prob_in, stop_prob, grad_in, grad_out):
        """extension: forms stop_prob*grad_in in-kernel (calc_prob.py:27 folded in)"""
        return _call("genre_calc_prob_backward_fused", prob_in, stop_prob, grad_in, grad_out, out=(3,))


class _RenderLib:
    """fused render_spherical (extension; fuses toolbox/spherical_proj.py:62-72)"""

    @staticmethod
    def render_spherical_forward(vox, dirs64_as_f32, depth_weight, out,
                                 v_scratch=None, fwd_table=None, fwd_chunks=None, kin=None, pre_scale=0.0, live=None):
        """with the four optional tensors: LDS-staged brick sampling + scan (v_scratch receives the raw
        sample values); without: one wave-per-ray gather kernel.  live (int32 [N*NC*(1 + bricks)], with pre_scale): receives
        the clamp's pass words per image and per 16^3 brick for the backward"""
        return _call("genre_render_spherical_forward", vox, dirs64_as_f32, depth_weight, out,
                     v_scratch, fwd_table, fwd_chunks, kin, live, scalars=(C.c_float(pre_scale),), out=(3, 4, 8))

    @staticmethod
    def render_spherical_backward(vox, dirs64_as_f32, depth_weight, grad_out, grad_vox,
                                  dp_scratch=None, brick_table=None, chunk_list=None, v_scratch=None, kin=None,
                                  pre_scale=0.0, live=None):
        """dp_scratch/brick_table/chunk_list given: brick-owned backward (no global atomics), re-using the
        forward's v_scratch when it is passed too; without: global-atomic scatter fallback.  live: the forward's pass words --
        what the pre_scale clamp blocks is written as zeros without being computed"""
        return _call("genre_render_spherical_backward", vox, dirs64_as_f32, depth_weight, grad_out, grad_vox,
                     dp_scratch, brick_table, chunk_list, v_scratch, kin, live, scalars=(C.c_float(pre_scale),), out=(4, 5))

    @staticmethod
    def render_seg_forward(vox, dirs64_as_f32, depth_weight, out, seg_rows, segs, ray_nseg, ray_pre_as_f32, line_w, ps_scratch,
                           pre_scale=0.0, live=None, occ=None, ps_empty=None, occ_cell=0, v_scratch=None):
        """segment renderer, standard layout (csrc/sph_render_seg.hip; tables: toolbox/_seg_tables.py): one (P, S) pair per
        segment instead of one value per sample.  occ + ps_empty + occ_cell: the producer's occupancy words -- tiles known to
        hold only the fill value are not read.  v_scratch: a backward will follow -- the raw sample values of the tiles a gradient
        can come back through are saved too"""
        return _call("genre_render_seg_forward", vox, dirs64_as_f32, depth_weight, out, seg_rows, segs, ray_nseg,
                     ray_pre_as_f32, line_w, ps_scratch, live, occ, ps_empty, v_scratch,
                     scalars=(C.c_float(pre_scale), C.c_int(occ_cell)), out=(3, 9, 10, 13))

    @staticmethod
    def render_seg_backward(vox, dirs64_as_f32, depth_weight, grad_out, grad_vox, bwd_rows, segs, ray_nseg, ray_pre_as_f32,
                            line_w, ps_scratch, tr_scratch, v_scratch, halo_scratch, pre_scale=0.0, live=None):
        """the backward of render_seg_forward from the state it left (ps_scratch, v_scratch, live): per-ray chains, then one
        pass over the segments that scatters dL/dp times the trilinear weights into the bricks' tiles"""
        return _call("genre_render_seg_backward", vox, dirs64_as_f32, depth_weight, grad_out, grad_vox, bwd_rows, segs,
                     ray_nseg, ray_pre_as_f32, line_w, ps_scratch, tr_scratch, v_scratch, halo_scratch, live,
                     scalars=(C.c_float(pre_scale),), out=(4, 11, 13))


    @staticmethod
    def render_bm_forward(vox, out, segs, rec_f, fwd_rows, ray_ptr, ray_seg, ray_pre_as_f32, ps_scratch,
                          p_stash=None, mask=None, pre_scale=0.0, tile_live=None, ps_empty=None):
        """batch-minor tile renderer (csrc/sph_render_bm.hip; tables: toolbox/_bm_tables.py).  p_stash (and mask when
        pre_scale != 0) given: the state the backward needs is saved.  tile_live + ps_empty: the producer's occupancy words and
        the geometry's constants -- tiles known to hold only the fill value are not read"""
        return _call("genre_render_bm_forward", vox, out, segs, rec_f, fwd_rows, ray_ptr, ray_seg, ray_pre_as_f32,
                     ps_scratch, p_stash, mask, tile_live, ps_empty, scalars=(C.c_float(pre_scale),), out=(1, 8, 9, 10))

    @staticmethod
    def render_bm_backward(grad_out, grad_vox, segs, ray_ptr, ray_seg, ray_pre_as_f32, ent, rec_b, bwd_rows,
                           depth_weight, ps_scratch, tr_scratch, p_stash, mask=None, pre_scale=0.0, pull_brick=488):
        return _call("genre_render_bm_backward", grad_out, grad_vox, segs, ray_ptr, ray_seg, ray_pre_as_f32, ent,
                     rec_b, bwd_rows, depth_weight, ps_scratch, tr_scratch, p_stash, mask,
                     scalars=(C.c_float(pre_scale), C.c_int(pull_brick)), out=(1, 11))



class _GlueLib:
    """GenRe caller glue folded into single passes (extension)"""

    @staticmethod
    def abs_depth_forward(pred_depth, depth_minmax, silhou, out, scale_25d=100.0):
        """depth_pred_with_sph_inpaint.py:131-142 (get_abs_depth) in one pass"""
        return _call("genre_abs_depth_forward", pred_depth, depth_minmax, silhou, out, scalars=(C.c_float(scale_25d),), out=(3,))

    @staticmethod
    def abs_depth_backward(grad_out, depth_minmax, silhou, grad_pred, scale_25d=100.0):
        return _call("genre_abs_depth_backward", grad_out, depth_minmax, silhou, grad_pred,
                     scalars=(C.c_float(scale_25d),), out=(3,))


class _MyLib:
    """stands in for nndistance/_ext/my_lib (my_lib.h:3-5, my_lib_cuda.h:1-4): the two CUDA entries run the HIP
    kernels, the two CPU entries the host code of csrc/nnd_host.hip (CPU tensors only; nothing falls back to them)."""

    @staticmethod
    def nnd_forward_cuda(xyz1, xyz2, dist1, dist2, idx1, idx2):
        return _call("genre_nnd_forward", xyz1, xyz2, dist1, dist2, idx1, idx2, out=(2, 3, 4, 5))

    @staticmethod
    def nnd_backward_cuda(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
        return _call("genre_nnd_backward", xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2, out=(2, 3))

    @staticmethod
    def nnd_forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
        """my_lib.h:3 -- the reference's CPU entry (host tensors)"""
        return _call_host("genre_nnd_forward_host", xyz1, xyz2, dist1, dist2, idx1, idx2)

    @staticmethod
    def nnd_backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
        """my_lib.h:5"""
        return _call_host("genre_nnd_backward_host", xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)


cam_bp_lib = _CamBpLib()
calc_prob_lib = _CalcProbLib()
my_lib = _MyLib()
render_lib = _RenderLib()
glue_lib = _GlueLib()
