"""CPU-torch restatement of the PyTorch-level pieces of the hot path -- TEST INFRASTRUCTURE ONLY.

The reference's render_spherical / sph_pad / Camera_back_projection_layer.shift_tdf are plain
PyTorch-0.4.1 ops around the native kernels (toolbox/spherical_proj.py:21-72,
camera_backprojection_module.py:12-28).  Here they are restated on CPU torch with the native
ops replaced by the C oracle (oracle.Oracle) or the host-compiled reference (oracle.Reference),
wrapped in autograd Functions so a whole forward+backward chain can be checked / timed on
the host.  ``grid_sample`` is called with ``align_corners=True`` -- the 0.4.1 semantics
(environment.yml:14; today's default differs).

Used by tests/ (parity of render_spherical and of the full chain) and by bench.py's
cpu_baseline leg.  Never imported by the product.
"""
import numpy as np
import torch
from torch.autograd import Function


def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy())


def make_functions(backend):
    """autograd Functions over `backend` (an oracle.Oracle or oracle.Reference instance)"""

    class CameraBackProjectionCPU(Function):
        # cam_back_projection.py:12-46
        @staticmethod
        def forward(ctx, depth_t, fl, cam_dist, res=128):
            tdf, cnt = backend.back_projection_forward(_np(depth_t), _np(cam_dist), _np(fl), res)
            ctx.save_for_backward(depth_t, fl, cam_dist)
            ctx.cnt = cnt
            return torch.from_numpy(tdf)

        @staticmethod
        def backward(ctx, grad_output):
            depth_t, fl, cam_dist = ctx.saved_tensors
            N = depth_t.shape[0]
            gds, gcs, gfs = [], [], []
            for i in range(N):        # per sample: the reference's K4 is only valid at N == 1 (F8)
                r = backend.back_projection_backward(_np(depth_t[i:i + 1]), _np(fl[i:i + 1]),
                                                     _np(cam_dist[i:i + 1]), ctx.cnt[i:i + 1],
                                                     _np(grad_output[i:i + 1]))
                gds.append(r[0]); gcs.append(r[1]); gfs.append(r[2])
            return (torch.from_numpy(np.concatenate(gds)), torch.from_numpy(np.concatenate(gfs)),
                    torch.from_numpy(np.concatenate(gcs)), None)

    class CalcStopProbCPU(Function):
        # calc_prob.py:10-29
        @staticmethod
        def forward(ctx, prob_in):
            s = torch.from_numpy(backend.calc_prob_forward(_np(prob_in)))
            ctx.save_for_backward(prob_in, s)
            return s

        @staticmethod
        def backward(ctx, grad_in):
            prob_in, s = ctx.saved_tensors
            w = s * grad_in                                           # :27
            return torch.from_numpy(backend.calc_prob_backward(_np(prob_in), _np(w)))

    class SphericalBackProjectionCPU(Function):
        # sperical_to_tdf.py:13-47
        @staticmethod
        def forward(ctx, spherical, grid, res=128):
            g = grid.detach().cpu().numpy()                            # keeps expand()ed strides
            tdf, cnt = backend.spherical_back_proj_forward(_np(spherical), g, res)
            ctx.save_for_backward(spherical.detach(), grid)
            ctx.cnt = cnt
            cnt_t = torch.from_numpy(cnt)
            ctx.mark_non_differentiable(cnt_t)
            return torch.from_numpy(tdf), cnt_t

        @staticmethod
        def backward(ctx, grad_output, grad_phony):
            spherical, grid = ctx.saved_tensors
            gd = backend.spherical_back_proj_backward(_np(spherical), grid.detach().cpu().numpy(), ctx.cnt,
                                                      _np(grad_output))
            return torch.from_numpy(gd), None, None

    return CameraBackProjectionCPU, CalcStopProbCPU, SphericalBackProjectionCPU


def unit_dirs(res):
    phi = np.linspace(0, 180, res * 2 + 1)[1::2] * np.pi / 180
    theta = np.linspace(0, 360, res + 1)[:-1] * np.pi / 180
    g = np.zeros((res, res, 3))
    g[:, :, 2] = np.cos(phi)[:, None]
    g[:, :, 0] = np.sin(phi)[:, None] * np.cos(theta)[None, :]
    g[:, :, 1] = np.sin(phi)[:, None] * np.sin(theta)[None, :]
    return g


def render_grid(sph_res=128, z_res=256):
    """spherical_proj.py:39-60: sample k of ray (i,j) at 2*dir*(1-alpha_k), alpha = linspace(0,1,z_res)"""
    alpha = np.linspace(0, 1, z_res).reshape(1, 1, z_res, 1)
    grid = (unit_dirs(sph_res) * 2)[:, :, np.newaxis, :] * (1 - alpha)
    return torch.from_numpy(grid).float(), torch.linspace(0, 1, z_res)


class RenderSphericalCPU(torch.nn.Module):
    """spherical_proj.py:31-72 on CPU torch + the oracle's calc_prob"""

    def __init__(self, backend, sph_res=128, z_res=256):
        super().__init__()
        self.grid, self.depth_weight = render_grid(sph_res, z_res)
        self.calc = make_functions(backend)[1].apply

    def forward(self, vox):
        grid = self.grid.expand(vox.shape[0], -1, -1, -1, -1)
        vox = vox.permute(0, 1, 4, 3, 2)
        prob = torch.nn.functional.grid_sample(vox, grid, mode="bilinear", padding_mode="zeros",
                                               align_corners=True)
        prob = torch.clamp(prob, 1e-5, 1 - 1e-5)
        stop = self.calc(prob)
        exp_depth = torch.matmul(stop, self.depth_weight)
        return exp_depth + torch.prod(1.0 - prob, dim=4)


class RenderSphericalF64(torch.nn.Module):
    """spherical_proj.py:31-72 evaluated in float64 on CPU torch (closed forms of calc_prob_kernel.cu:129-141): the
    yardstick for how far an fp32 implementation (the reference's op sequence, the fused kernels) sits from the exact
    value.  samples="fp32" (default): the trilinear SAMPLE VALUES are the fp32 ones every fp32 implementation computes
    (ATen grid_sampler_3d in float32 -- the reference's own arithmetic), promoted to double; everything downstream
    (clamp, transmittance, expectation) and the whole adjoint are float64.  That separates the error of the scans
    and of the gradient accumulation -- what the kernels are responsible for -- from the 2^-24 rounding of the sample
    values, which the clamp bounds and 1/(1-p) amplify identically for every fp32 implementation.  samples="fp64":
    the samples are interpolated in double as well."""

    def __init__(self, sph_res=128, z_res=256, samples="fp32"):
        super().__init__()
        grid, dw = render_grid(sph_res, z_res)
        self.grid32, self.grid, self.depth_weight = grid, grid.double(), dw.double()
        self.lo, self.hi = float(np.float32(1e-5)), float(np.float32(1 - 1e-5))
        self.samples = samples

    def forward(self, vox):
        v64 = vox.double().permute(0, 1, 4, 3, 2)
        kw = dict(mode="bilinear", padding_mode="zeros", align_corners=True)
        prob = torch.nn.functional.grid_sample(v64, self.grid.expand(vox.shape[0], -1, -1, -1, -1), **kw)
        if self.samples == "fp32":
            with torch.no_grad():
                p32 = torch.nn.functional.grid_sample(vox.detach().float().permute(0, 1, 4, 3, 2),
                                                      self.grid32.expand(vox.shape[0], -1, -1, -1, -1), **kw).double()
            prob = prob + (p32 - prob).detach()            # fp32 sample values, float64 adjoint
        prob = torch.clamp(prob, self.lo, self.hi)
        q = 1.0 - prob
        trans = torch.cumprod(q, dim=4)
        before = torch.cat((torch.ones_like(trans[..., :1]), trans[..., :-1]), dim=4)     # prod_{j<k} (1 - p_j)
        stop = prob * before
        return torch.matmul(stop, self.depth_weight) + trans[..., -1]


class _RenderExact(Function):
    """float64 evaluation of the reference's fp32-DEFINED operator: the trilinear sample values are ATen's float32
    ones (torch grid_sample on CPU -- the op the reference calls, spherical_proj.py:65) and the adjoint uses ATen's
    float32 corner weights (its float32 coordinate arithmetic is part of the operator: at |ix| ~ 100 one ulp is
    8e-6, so float64-interpolated weights describe a slightly different operator), but every product and every sum
    -- transmittance, expectation, dL/dp, and the accumulation of up to 2^17 contributions per voxel -- is float64."""

    @staticmethod
    def forward(ctx, vox, grid32, depth_weight):
        n, c, X, Y, Z = vox.shape
        assert c == 1
        kw = dict(mode="bilinear", padding_mode="zeros", align_corners=True)
        v32 = torch.nn.functional.grid_sample(vox.detach().float().permute(0, 1, 4, 3, 2),
                                              grid32.expand(n, -1, -1, -1, -1), **kw)[:, 0].double().numpy()
        lo, hi = float(np.float32(1e-5)), float(np.float32(1 - 1e-5))
        p = np.clip(v32, lo, hi)                                            # [n, R, R, ZR]
        w = depth_weight.double().numpy()
        trans = np.cumprod(1.0 - p, axis=-1)
        before = np.concatenate((np.ones_like(trans[..., :1]), trans[..., :-1]), -1)
        s = p * before
        out = (s * w).sum(-1) + trans[..., -1]
        ctx.state = (v32, p, before, s, w, trans[..., -1], grid32.numpy(), (X, Y, Z))
        return torch.from_numpy(out)[:, None]

    @staticmethod
    def backward(ctx, g):
        v32, p, before, s, w, tail, grid, (X, Y, Z) = ctx.state
        lo, hi = float(np.float32(1e-5)), float(np.float32(1 - 1e-5))
        gd = g[:, 0].double().numpy()[..., None]
        sw = s * w
        after = np.flip(np.cumsum(np.flip(sw, -1), -1), -1) - sw + tail[..., None]       # sum_{j>k} s_j w_j + prod(1-p)
        dp = gd * (before * w - after / (1.0 - p)) * ((v32 >= lo) & (v32 <= hi))
        # ATen grid_sampler_3d corner indices and float32 weights (align_corners=True); grid x -> X axis
        one, two = np.float32(1), np.float32(2)
        idx, wts = [], []
        for ax, size in enumerate((X, Y, Z)):
            ix = ((grid[..., ax] + one) / two) * np.float32(size - 1)
            f = np.floor(ix)
            idx.append(f.astype(np.int64))
            wts.append(((f + one) - ix, ix - f))
        grad = np.zeros((gd.shape[0], X * Y * Z))
        for cnr in range(8):
            b = (cnr & 1, (cnr >> 1) & 1, (cnr >> 2) & 1)
            ci = [idx[a] + b[a] for a in range(3)]
            ok = np.ones(ci[0].shape, bool)
            for a, size in enumerate((X, Y, Z)):
                ok &= (ci[a] >= 0) & (ci[a] < size)
            wc = ((wts[0][b[0]] * wts[1][b[1]]) * wts[2][b[2]]).astype(np.float64)      # ATen: (wx * wy) * wz in float32
            lin = ((ci[0] * Y + ci[1]) * Z + ci[2])[ok]
            for i in range(gd.shape[0]):
                grad[i] += np.bincount(lin, weights=(wc * dp[i])[ok], minlength=X * Y * Z)
        return torch.from_numpy(grad.reshape(gd.shape[0], 1, X, Y, Z)), None, None


class RenderSphericalExact(torch.nn.Module):
    """the yardstick of the gradient parity tests (see _RenderExact); vox [N,1,X,Y,Z] -> float64 [N,1,R,R]"""

    def __init__(self, sph_res=128, z_res=256):
        super().__init__()
        self.grid, self.depth_weight = render_grid(sph_res, z_res)

    def forward(self, vox):
        return _RenderExact.apply(vox.double(), self.grid, self.depth_weight)


def sph_pad(sph, pm=16):
    """spherical_proj.py:21-28"""
    out = torch.nn.functional.pad(sph, (pm, pm, pm, pm), mode="replicate")
    _, _, h, w = out.shape
    out[:, :, :, 0:pm] = out[:, :, :, w - 2 * pm:w - pm]
    out[:, :, :, h - pm:] = out[:, :, :, pm:2 * pm]
    return out


class HotPathCPU:
    """configs[1] on the host: depth -> cam_bp -> shift/x50/clamp -> render_spherical -> pad,
    forward and backward (depth_pred_with_sph_inpaint.py:120-126)."""

    def __init__(self, backend, fl=418.3, cam_dist=2.2):
        self.cam = make_functions(backend)[0].apply
        self.render = RenderSphericalCPU(backend)
        self.fl, self.cam_dist = fl, cam_dist

    def forward(self, depth):
        n = depth.shape[0]
        fl = torch.full((n, 1), self.fl)
        cd = torch.full((n, 1), self.cam_dist)
        tdf = self.cam(depth, fl, cd, 128)
        proj = 1 - 128 * tdf                                           # shift_tdf
        sph = self.render(torch.clamp(proj * 50, 1e-5, 1 - 1e-5))
        return sph_pad(sph, 16)

    def forward_backward(self, depth, grad_out):
        depth = depth.clone().requires_grad_(True)
        out = self.forward(depth)
        out.backward(grad_out)
        return out.detach(), depth.grad


class GenReGlueCPU:
    """the two glue sections of the reference's GenRe models on CPU torch + the oracle
    (depth_pred_with_sph_inpaint.py:120-129, genre_full_model.py:122-127,134-143)"""

    def __init__(self, backend, margin=16):
        self.hot = HotPathCPU(backend)
        self.sph_bp = make_functions(backend)[2].apply
        self.margin = margin
        self.grid = torch.from_numpy(unit_dirs(128).reshape(1, 1, 128, 128, 3)).float()

    @staticmethod
    def get_abs_depth(pred_depth, depth_minmax, silhou, scale_25d=100):
        """depth_pred_with_sph_inpaint.py:131-142 with marrnetbase.py:138-151 inlined (postprocess = x / scale_25d,
        to_abs_depth = rel * (max - min + 1e-4) + min), on CPU torch"""
        pred = pred_depth / scale_25d                                  # postprocess (:133)
        mm = depth_minmax.detach()                                     # :134
        depth_min = mm[:, 0].view(-1, 1, 1, 1)
        depth_max = mm[:, 1].view(-1, 1, 1, 1)
        abs_depth = (1 - pred) * (depth_max - depth_min + 1e-4) + depth_min      # :135, marrnetbase.py:150
        sil = (silhou / scale_25d).detach()                            # :136
        abs_depth[sil < 0.5] = 0                                       # :137
        abs_depth = abs_depth.permute(0, 1, 3, 2)                      # :138
        return torch.flip(abs_depth, [2])                              # :139

    def depth_to_spherical(self, depth):
        n = depth.shape[0]
        tdf = self.hot.cam(depth, torch.full((n, 1), self.hot.fl), torch.full((n, 1), self.hot.cam_dist), 128)
        proj = 1 - 128 * tdf
        sph_in = self.hot.render(torch.clamp(proj * 50, 1e-5, 1 - 1e-5))
        return proj * 50, sph_pad(sph_in, self.margin)

    def refiner_input(self, sph, proj_depth):
        b, _, h, w = sph.shape
        m = self.margin
        grid = self.grid.expand(b, -1, -1, -1, -1)
        crop = sph[:, :, m:h - m, m:w - m]
        proj_df, cnt = self.sph_bp(1 - crop, grid, 128)
        mask = torch.clamp(cnt.detach(), 0, 1)
        proj_df = (-proj_df + 1 / 128) * 128
        proj_df = proj_df * mask
        pd = torch.clamp(proj_depth / 50, 1e-5, 1 - 1e-5)
        return torch.cat((proj_df, pd), dim=1), cnt


def make_nnd(backend):
    """toolbox/nndistance/functions/nnd.py:8-63 over the oracle's Chamfer (my_lib.c:30-118)"""

    class NNDCPU(Function):
        @staticmethod
        def forward(ctx, xyz1, xyz2):
            d1, d2, i1, i2 = backend.nnd_forward(_np(xyz1), _np(xyz2))
            ctx.save_for_backward(xyz1, xyz2)
            ctx.idx = (i1, i2)
            return torch.from_numpy(d1), torch.from_numpy(d2)

        @staticmethod
        def backward(ctx, gd1, gd2):
            xyz1, xyz2 = ctx.saved_tensors
            g1, g2 = backend.nnd_backward(_np(xyz1), _np(xyz2), _np(gd1), _np(gd2), *ctx.idx)
            return torch.from_numpy(g1), torch.from_numpy(g2)

    return NNDCPU


class GenReCPU:
    """the forward of the reference's GenRe full model (genre_full_model.py:116-143 around
    depth_pred_with_sph_inpaint.py:113-129) on CPU torch: the three networks are the ones handed in (CPU copies of the
    model under test or the reference's own classes), every geometric op between them is the oracle's, wrapped in the
    autograd Functions above so the chain is differentiable end to end.  Returns the reference's output dict."""

    def __init__(self, backend, net1, net2, refine_net, margin=16):
        self.glue = GenReGlueCPU(backend, margin)
        self.net1, self.net2, self.refine_net = net1, net2, refine_net
        self.nnd = make_nnd(backend).apply

    def forward(self, input_struct, joint_train=True):
        with torch.set_grad_enabled(joint_train and torch.is_grad_enabled()):
            out = self.net1(input_struct)
            depth = self.glue.get_abs_depth(out["depth"], out["depth_minmax"], input_struct.silhou)
            proj50, sph_in = self.glue.depth_to_spherical(depth)
            out["abs_depth"] = depth
            out["proj_depth"] = proj50
            out["pred_sph_partial"] = sph_in
            out["pred_sph_full"] = self.net2(sph_in)["spherical"]
        refine_input, _ = self.glue.refiner_input(out["pred_sph_full"], out["proj_depth"])
        out["pred_proj_sph_full"] = refine_input[:, 0:1]
        out["pred_proj_depth"] = refine_input[:, 1:2]
        out["pred_voxel"] = self.refine_net(refine_input)
        return out
