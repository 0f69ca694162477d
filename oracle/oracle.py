"""ctypes/numpy bindings for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  Nothing under genre-shapehd_amd/ does (tests/
test_boundary.py greps for it).

Two back ends with the same Python surface:

* ``Oracle()``      -> oracle/liboracle.so, our plain-C restatement
                       (genre_oracle.c; every function cites reference file:line)
* ``Reference()``   -> oracle/_ref/libref_kernels.so + libref_mylib.so, the
                       reference's own kernel bodies / my_lib.c host-compiled by
                       oracle/build_ref.py (exists only where /root/reference was
                       available at build time, or travelled with the snapshot)

All arrays are C-contiguous numpy float32 (int32 for idx); shapes as in the
reference: depth [N,NC,H,W], voxel/cnt [N,NC,R,R,R], prob [N,NC,X,Y,Z],
xyz [B,n,3].
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_f = np.float32
FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int32)
DP = C.POINTER(C.c_double)
LP = C.POINTER(C.c_int64)


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(FP)


def _ip(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(IP)


def _c(a, dtype=_f):
    return np.ascontiguousarray(a, dtype=dtype)


def build_oracle():
    """Compile liboracle.so (gcc, <1 s).  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    return os.path.join(HERE, "liboracle.so")


def build_reference():
    """Compile oracle/_ref from /root/reference when that tree is present."""
    rc = subprocess.call(["python3", os.path.join(HERE, "build_ref.py")])
    return rc == 0


def reference_available():
    return all(os.path.exists(os.path.join(HERE, "_ref", n))
               for n in ("libref_kernels.so", "libref_mylib.so"))


def _grid_strides(grid):
    """element strides of a [N,NC,H,W,3] float32 array (may be a broadcast view)"""
    assert grid.dtype == np.float32 and grid.ndim == 5 and grid.shape[4] == 3
    return (C.c_int64 * 5)(*[s // 4 for s in grid.strides])


class _Base:
    """Shared numpy-level surface; subclasses supply the C symbols."""

    # ---- cam_bp ------------------------------------------------------------
    def back_projection_forward(self, depth, camdist, fl, res=128):
        depth, camdist, fl = _c(depth), _c(camdist), _c(fl)
        N, NC, H, W = depth.shape
        vox = np.empty((N, NC, res, res, res), _f)
        cnt = np.empty_like(vox)
        self._bp_fwd(_fp(depth), N, NC, H, W, _fp(camdist), _fp(fl), _fp(vox), _fp(cnt), res, res, res)
        return vox, cnt

    def get_surface_mask(self, depth, camdist, fl, cnt):
        depth, camdist, fl, cnt = _c(depth), _c(camdist), _c(fl), _c(cnt)
        N, NC, H, W = depth.shape
        X, Y, Z = cnt.shape[2:]
        mask = np.empty_like(cnt)
        self._mask(_fp(depth), N, NC, H, W, _fp(camdist), _fp(fl), _fp(cnt), _fp(mask), X, Y, Z)
        return mask

    def spherical_back_proj_forward(self, sph, grid, res=128):
        sph = _c(sph)
        N, NC, H, W = sph.shape
        assert grid.shape == (N, NC, H, W, 3)
        vox = np.empty((N, NC, res, res, res), _f)
        cnt = np.empty_like(vox)
        self._sph_fwd(_fp(sph), N, NC, H, W, grid.ctypes.data_as(FP), _grid_strides(grid),
                      _fp(vox), _fp(cnt), res, res, res)
        return vox, cnt

    def spherical_back_proj_backward(self, sph, grid, cnt, grad_in):
        sph, cnt, grad_in = _c(sph), _c(cnt), _c(grad_in)
        N, NC, H, W = sph.shape
        X, Y, Z = cnt.shape[2:]
        gd = np.empty_like(sph)
        self._sph_bwd(_fp(sph), N, NC, H, W, grid.ctypes.data_as(FP), _grid_strides(grid),
                      _fp(cnt), _fp(grad_in), X, Y, Z, _fp(gd))
        return gd

    # ---- calc_prob -----------------------------------------------------------
    def calc_prob_forward(self, prob_in):
        prob_in = _c(prob_in)
        out = np.empty_like(prob_in)
        self._cp_fwd_call(prob_in, out)
        return out

    def calc_prob_backward(self, prob_in, stop_prob_weighted):
        prob_in, spw = _c(prob_in), _c(stop_prob_weighted)
        out = np.empty_like(prob_in)
        self._cp_bwd_call(prob_in, spw, out)
        return out


class Oracle(_Base):
    """Our C restatement (oracle/genre_oracle.c)."""
    kind = "port"

    def __init__(self):
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(
                os.path.join(HERE, "genre_oracle.c")):
            build_oracle()
        L = self.lib = C.CDLL(path)
        i = C.c_int
        L.oracle_back_projection_forward.argtypes = [FP, i, i, i, i, FP, FP, FP, FP, i, i, i]
        L.oracle_back_projection_backward.argtypes = [FP, i, i, i, i, FP, FP, FP, FP, i, i, i,
                                                      FP, FP, FP, DP, DP]
        L.oracle_get_surface_mask.argtypes = [FP, i, i, i, i, FP, FP, FP, FP, i, i, i]
        L.oracle_spherical_back_proj_forward.argtypes = [FP, i, i, i, i, FP, LP, FP, FP, i, i, i]
        L.oracle_spherical_back_proj_backward.argtypes = [FP, i, i, i, i, FP, LP, FP, FP, i, i, i, FP]
        L.oracle_calc_prob_forward.argtypes = [FP, FP, C.c_int64, i]
        L.oracle_calc_prob_backward.argtypes = [FP, FP, FP, C.c_int64, i]
        L.oracle_nnsearch.argtypes = [i, i, i, FP, FP, FP, IP]
        L.oracle_nnd_forward.argtypes = [i, i, i, FP, FP, FP, FP, IP, IP]
        L.oracle_nnd_backward.argtypes = [i, i, i, FP, FP, FP, FP, FP, FP, IP, IP]
        self._bp_fwd = L.oracle_back_projection_forward
        self._mask = L.oracle_get_surface_mask
        self._sph_fwd = L.oracle_spherical_back_proj_forward
        self._sph_bwd = L.oracle_spherical_back_proj_backward

    def back_projection_backward(self, depth, fl, camdist, cnt, grad_in, with_double=False):
        depth, fl, camdist, cnt, grad_in = _c(depth), _c(fl), _c(camdist), _c(cnt), _c(grad_in)
        N, NC, H, W = depth.shape
        X, Y, Z = cnt.shape[2:]
        gd = np.empty_like(depth)
        gc = np.empty((N, NC), _f)
        gf = np.empty((N, NC), _f)
        gcd = np.empty((N, NC), np.float64)
        gfd = np.empty((N, NC), np.float64)
        self.lib.oracle_back_projection_backward(
            _fp(depth), N, NC, H, W, _fp(fl), _fp(camdist), _fp(cnt), _fp(grad_in), X, Y, Z,
            _fp(gd), _fp(gc), _fp(gf), gcd.ctypes.data_as(DP), gfd.ctypes.data_as(DP))
        if with_double:
            return gd, gc, gf, gcd, gfd
        return gd, gc, gf

    def _cp_fwd_call(self, p, out):
        Z = p.shape[-1]
        self.lib.oracle_calc_prob_forward(_fp(p), _fp(out), p.size // Z, Z)

    def _cp_bwd_call(self, p, w, out):
        Z = p.shape[-1]
        self.lib.oracle_calc_prob_backward(_fp(p), _fp(w), _fp(out), p.size // Z, Z)

    def nnd_forward(self, xyz1, xyz2):
        xyz1, xyz2 = _c(xyz1), _c(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        d1, d2 = np.empty((b, n), _f), np.empty((b, m), _f)
        i1, i2 = np.empty((b, n), np.int32), np.empty((b, m), np.int32)
        self.lib.oracle_nnd_forward(b, n, m, _fp(xyz1), _fp(xyz2), _fp(d1), _fp(d2), _ip(i1), _ip(i2))
        return d1, d2, i1, i2

    def nnd_backward(self, xyz1, xyz2, gd1, gd2, idx1, idx2):
        xyz1, xyz2, gd1, gd2 = _c(xyz1), _c(xyz2), _c(gd1), _c(gd2)
        idx1, idx2 = _c(idx1, np.int32), _c(idx2, np.int32)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1, g2 = np.empty_like(xyz1), np.empty_like(xyz2)
        self.lib.oracle_nnd_backward(b, n, m, _fp(xyz1), _fp(xyz2), _fp(g1), _fp(g2),
                                     _fp(gd1), _fp(gd2), _ip(idx1), _ip(idx2))
        return g1, g2


class _StubTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("size", C.c_long * 3)]


def _stub(a):
    t = _StubTensor()
    t.data = a.ctypes.data
    for k in range(min(3, a.ndim)):
        t.size[k] = a.shape[k]
    return t


class Reference(_Base):
    """The reference's own code, host-compiled (oracle/_ref)."""
    kind = "reference"

    def __init__(self):
        if not reference_available():
            raise FileNotFoundError("oracle/_ref is not built (python oracle/build_ref.py needs /root/reference)")
        K = self.k = C.CDLL(os.path.join(HERE, "_ref", "libref_kernels.so"))
        M = self.m = C.CDLL(os.path.join(HERE, "_ref", "libref_mylib.so"))
        i = C.c_int
        K.ref_back_projection_forward.argtypes = [FP, i, i, i, i, FP, FP, FP, FP, i, i, i]
        K.ref_back_projection_backward.argtypes = [FP, i, i, i, i, FP, FP, FP, FP, i, i, i, FP, FP, FP]
        K.ref_get_surface_mask.argtypes = [FP, i, i, i, i, FP, FP, FP, FP, i, i, i]
        K.ref_spherical_back_proj_forward.argtypes = [FP, i, i, i, i, FP, LP, FP, FP, i, i, i]
        K.ref_spherical_back_proj_backward.argtypes = [FP, i, i, i, i, FP, LP, FP, FP, i, i, i, FP]
        K.ref_calc_prob_forward.argtypes = [FP, FP, i, i, i, i, i]
        K.ref_calc_prob_backward.argtypes = [FP, FP, FP, i, i, i, i, i]
        K.ref_nnd_forward_kernels.argtypes = [i, i, FP, i, FP, FP, IP, FP, IP]
        K.ref_nnd_backward_kernels.argtypes = [i, i, FP, i, FP, FP, IP, FP, IP, FP, FP]
        M.nnsearch.argtypes = [i, i, i, FP, FP, FP, IP]
        M.nnd_forward.argtypes = [C.c_void_p] * 6
        M.nnd_backward.argtypes = [C.c_void_p] * 8
        self._bp_fwd = K.ref_back_projection_forward
        self._mask = K.ref_get_surface_mask
        self._sph_fwd = K.ref_spherical_back_proj_forward
        self._sph_bwd = K.ref_spherical_back_proj_backward

    def back_projection_backward(self, depth, fl, camdist, cnt, grad_in):
        """NB reference bug F8 (back_projection_kernel.cu:401): only valid at N == 1."""
        depth, fl, camdist, cnt, grad_in = _c(depth), _c(fl), _c(camdist), _c(cnt), _c(grad_in)
        N, NC, H, W = depth.shape
        assert N == 1 and NC == 1, "reference K4 reads camdist out of bounds for n>0 (SURVEY F8)"
        X, Y, Z = cnt.shape[2:]
        gd = np.empty_like(depth)
        gc = np.empty((N, NC), _f)
        gf = np.empty((N, NC), _f)
        self.k.ref_back_projection_backward(_fp(depth), N, NC, H, W, _fp(fl), _fp(camdist), _fp(cnt),
                                            _fp(grad_in), X, Y, Z, _fp(gd), _fp(gc), _fp(gf))
        return gd, gc, gf

    def _cp_fwd_call(self, p, out):
        N, NC, X, Y, Z = p.shape
        self.k.ref_calc_prob_forward(_fp(p), _fp(out), N, NC, X, Y, Z)

    def _cp_bwd_call(self, p, w, out):
        N, NC, X, Y, Z = p.shape
        self.k.ref_calc_prob_backward(_fp(p), _fp(w), _fp(out), N, NC, X, Y, Z)

    def nnd_forward(self, xyz1, xyz2, path="cpu"):
        """path='cpu': my_lib.c nnd_forward as shipped; 'cuda': NmDistanceKernel bodies."""
        xyz1, xyz2 = _c(xyz1), _c(xyz2)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        d1, d2 = np.zeros((b, n), _f), np.zeros((b, m), _f)
        i1, i2 = np.zeros((b, n), np.int32), np.zeros((b, m), np.int32)
        if path == "cpu":
            ts = [_stub(a) for a in (xyz1, xyz2, d1, d2, i1, i2)]
            self.m.nnd_forward(*[C.addressof(t) for t in ts])
        else:
            self.k.ref_nnd_forward_kernels(b, n, _fp(xyz1), m, _fp(xyz2), _fp(d1), _ip(i1), _fp(d2), _ip(i2))
        return d1, d2, i1, i2

    def nnd_backward(self, xyz1, xyz2, gd1, gd2, idx1, idx2, path="cpu"):
        xyz1, xyz2, gd1, gd2 = _c(xyz1), _c(xyz2), _c(gd1), _c(gd2)
        idx1, idx2 = _c(idx1, np.int32), _c(idx2, np.int32)
        b, n, _ = xyz1.shape
        m = xyz2.shape[1]
        g1, g2 = np.empty_like(xyz1), np.empty_like(xyz2)
        if path == "cpu":
            ts = [_stub(a) for a in (xyz1, xyz2, g1, g2, gd1, gd2, idx1, idx2)]
            self.m.nnd_backward(*[C.addressof(t) for t in ts])
        else:
            self.k.ref_nnd_backward_kernels(b, n, _fp(xyz1), m, _fp(xyz2), _fp(gd1), _ip(idx1),
                                            _fp(gd2), _ip(idx2), _fp(g1), _fp(g2))
        return g1, g2
