"""Camera_back_projection_layer -- drop-in for
toolbox/cam_bp/cam_bp/modules/camera_backprojection_module.py:6-28 (the only live class of
that file; its `camera_backprojection` and Spherical_backproj.py classes reference undefined
names in the reference and are not reproduced)."""
import torch
from torch import nn

from ..functions import CameraBackProjection
from ..functions.cam_back_projection import ShiftedCameraBackProjection


class Camera_back_projection_layer(nn.Module):
    def __init__(self, res=128, batch_minor=False):
        """batch_minor (extension): batches of >= 16 single-channel maps get their volume laid out with the image
        index fastest in memory (same logical shape and values; `.is_contiguous()` is False) -- the layout in which
        render_spherical's fused kernels are fastest, forward and backward (smaller batches keep NCXYZ whatever the flag says)."""
        super().__init__()
        assert res == 128
        self.res = 128
        self.batch_minor = batch_minor
        self._consts = {}

    _MAX_CONSTS = 32

    def _const(self, value, n, device):
        # the reference allocates + fills a fresh [n,1] tensor every call (:16-21); cache it so the
        # layer issues no extra kernels and can be captured in a HIP graph.
        # One tensor per (constant, batch size, device), never replaced while it is in the cache: a captured HIP graph
        # holds the raw pointer of the tensor that existed at capture, so a tensor handed out once must stay alive
        # (GenReInference additionally pins the tensors its graphs saw).  The cache is a small LRU so that it cannot grow
        # with the batch sizes seen; the tensors are read-only inputs of the op (autograd saves them) -- do not modify.
        key = (float(value), int(n), str(device))
        t = self._consts.pop(key, None)
        if t is None:
            t = torch.full((n, 1), float(value), dtype=torch.float32, device=device)
            while len(self._consts) >= self._MAX_CONSTS:
                self._consts.pop(next(iter(self._consts)))
        self._consts[key] = t                                            # most recently used last
        return t

    def forward(self, depth_t, fl=418.3, cam_dist=2.2, shift=True):
        n = depth_t.size(0)
        const = (fl, cam_dist) if type(fl) == float and type(cam_dist) == float and depth_t.size(1) == 1 else None
        if type(fl) == float:
            fl = self._const(fl, n, depth_t.device)
        if type(cam_dist) == float:
            cam_dist = self._const(cam_dist, n, depth_t.device)
        if shift:       # 1 - res*tdf evaluated inside the native op (same values as shift_tdf(df))
            bm = self.batch_minor and n >= 16 and depth_t.size(1) == 1
            if const is None:
                return ShiftedCameraBackProjection.apply(depth_t, fl, cam_dist, self.res, bm, None)
            # camera by value: the op also says which parts of the volume hold anything but the fill value -- per group of 32
            # images and renderer brick (leader pass, image-minor volumes) or per image and 8x8x32-voxel cell (brick kernel, dense
            # volumes); render_spherical then does not read what it knows to be empty (toolbox/_fused_render.py: occupancy_hint,
            # occupancy_hint_std; csrc/sph_render_bm.hip, csrc/sph_render_seg.hip)
            from ...._fused_render import attach_hint
            hint = {}
            out = ShiftedCameraBackProjection.apply(depth_t, fl, cam_dist, self.res, bm, const, hint)
            if hint.get("words") is None:
                return out
            return attach_hint(out, hint["words"], self.res, hint["cell"])
        return CameraBackProjection.apply(depth_t, fl, cam_dist, self.res)

    @staticmethod
    def shift_tdf(input_tdf, res=128):
        return 1 - res * input_tdf
