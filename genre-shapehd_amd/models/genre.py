"""GenRe on MI355X: the three networks composed around the native geometric ops.

    rgb [N,3,256,256] --net1 (U-ResNet-18)--> depth / normal / silhouette / depth min-max
      --get_abs_depth--> ray depth --cam_bp--> 128^3 projection --render_spherical--> partial spherical map
      --sph_pad--> net2 (inpainting U-ResNet-18) --> full spherical map --spherical back-projection--> 128^3
      cat(projected sphere, projected depth) --Unet_3D--> voxel logits

Module and buffer names equal the reference's (models/depth_pred_with_sph_inpaint.py:104-131, models/genre_full_model.py:
104-143, models/marrnet1.py:137-161), so `{'nets': [state_dict]}` checkpoints of the reference load key for key
(models/checkpoint.py).  Between the networks the glue is the fused native path of callers.py: get_abs_depth as one
pass, shift / x50 / clamp / sph_pad folded into cam_bp and the renderer, the refiner input written once."""
from dataclasses import dataclass
from types import SimpleNamespace

import torch
from torch import nn
import torch.nn.functional as F

from ..callers import AbsDepth, RefinerInput
from ..networks import Net as Uresnet, Net_inpaint, Unet_3D, ViewAsLinear
from ..toolbox.cam_bp.cam_bp.modules.camera_backprojection_module import Camera_back_projection_layer
from ..toolbox.spherical_proj import gen_sph_grid, render_spherical

SCALE_25D = 100.0            # models/marrnetbase.py:17


@dataclass
class GenReOptions:
    """the option fields the reference's Net classes read (depth_pred_with_sph_inpaint.py:14-26, genre_full_model.py:22-28)"""
    joint_train: bool = False
    load_offline: bool = False
    padding_margin: int = 16
    net1_path: str = None
    inpaint_path: str = None
    surface_weight: float = 1.0


def Inputs(rgb, silhou, **extra):
    """the `input_struct` of the reference's forward() methods: attributes rgb [N,3,H,W] and silhou [N,1,H,W] (x100,
    as its data pipeline scales 2.5-D maps by scale_25d)"""
    return SimpleNamespace(rgb=rgb, silhou=silhou, **extra)


class MarrNet1Net(Uresnet):
    """MarrNet-1: U-ResNet-18 with three decoders + the depth min/max head (models/marrnet1.py:137-161)"""

    def __init__(self, *args, pred_depth_minmax=True):
        super().__init__(*args)
        self.pred_depth_minmax = pred_depth_minmax
        if pred_depth_minmax:
            self.decoder_minmax = nn.Sequential(
                nn.Conv2d(512, 512, 2, stride=2), nn.Conv2d(512, 512, 4, stride=1), ViewAsLinear(),
                nn.Linear(512, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True),
                nn.Linear(256, 128), nn.BatchNorm1d(128), nn.ReLU(inplace=True), nn.Linear(128, 2))

    def forward(self, input_struct):
        out = super().forward(input_struct.rgb)
        if self.pred_depth_minmax:
            out["depth_minmax"] = self.decoder_minmax(self.encoder_out)
        return out


class DepthInpaintNet(nn.Module):
    """stage 1 + 2 (depth_pred_with_sph_inpaint.py:104-142)"""

    def __init__(self, opt=None):
        super().__init__()
        opt = opt or GenReOptions()
        self.net1 = MarrNet1Net([3, 1, 1], ["normal", "depth", "silhou"], pred_depth_minmax=True)
        self.net2 = Net_inpaint([1], ["spherical"], input_planes=1)
        # the volume's memory layout follows the batch size (round 5): batches of >= 16 images are laid out image-minor, where
        # the renderer's tile kernels (csrc/sph_render_bm.hip) run; the reference's batch sizes (1, 4, 8 per GPU) keep NCXYZ
        self.proj_depth = Camera_back_projection_layer(batch_minor=True)
        self.render_spherical = render_spherical()
        self.joint_train, self.load_offline, self.padding_margin = opt.joint_train, opt.load_offline, opt.padding_margin
        if opt.net1_path:
            self.net1.load_state_dict(torch.load(opt.net1_path, map_location="cpu")["nets"][0])

    def get_abs_depth(self, pred, input_struct):
        """:131-142 -- divide by scale_25d, 1 - x, min/max range, silhouette mask, permute + flip: one native pass"""
        return AbsDepth.apply(pred["depth"], pred["depth_minmax"], input_struct.silhou, SCALE_25D)

    def forward(self, input_struct):
        with torch.set_grad_enabled(self.joint_train and torch.is_grad_enabled()):
            out = self.net1(input_struct)
        proj = self.proj_depth(self.get_abs_depth(out, input_struct))              # 1 - 128 * tdf  (:120)
        if self.load_offline:
            from ..toolbox.spherical_proj import sph_pad
            sph_in = sph_pad(input_struct.spherical_depth, self.padding_margin)
        else:   # sph_pad(render(clamp(proj * 50, 1e-5, 1 - 1e-5)), margin)  (:124-126), clamp and padding folded in
            sph_in = self.render_spherical(proj, pre_scale=50.0, pad=self.padding_margin)
        out["proj_depth"] = proj * 50
        out["pred_sph_partial"] = sph_in
        out["pred_sph_full"] = self.net2(sph_in)["spherical"]
        return out


class GenReNet(nn.Module):
    """the full model (genre_full_model.py:104-143)"""

    def __init__(self, opt=None):
        super().__init__()
        opt = opt or GenReOptions()
        self.depth_and_inpaint = DepthInpaintNet(opt)
        self.refine_net = Unet_3D()
        self.proj_depth = Camera_back_projection_layer()
        self.joint_train = opt.joint_train
        self.register_buffer("grid", gen_sph_grid().expand(1, -1, -1, -1, -1).clone())
        self.margin = opt.padding_margin
        if opt.inpaint_path is not None:
            self.depth_and_inpaint.load_state_dict(torch.load(opt.inpaint_path, map_location="cpu")["nets"][0])

    def forward(self, input_struct):
        with torch.set_grad_enabled(self.joint_train and torch.is_grad_enabled()):
            out = self.depth_and_inpaint(input_struct)
        # :122-127,134-143: crop, 1 - x, spherical back-projection, (-tdf + 1/128) * 128 * mask into channel 0,
        # clamp(proj_depth / 50) into channel 1 -- one native op + one elementwise kernel, no cat
        grid = self.grid.expand(out["pred_sph_full"].shape[0], -1, -1, -1, -1)
        refine_input, _ = RefinerInput.apply(out["pred_sph_full"], grid, out["proj_depth"], self.margin)
        out["pred_proj_sph_full"] = refine_input[:, 0:1]
        out["pred_proj_depth"] = refine_input[:, 1:2]
        out["pred_voxel"] = self.refine_net(refine_input)
        return out


def genre_loss(pred, gt, opt=None, joint=False):
    """compute_loss of the three nested reference models: voxel BCE + surface term (genre_full_model.py:60-74); with
    joint training also the spherical MSE (depth_pred_with_sph_inpaint.py:60-69) and MarrNet-1's masked MSEs +
    min/max term (marrnet1.py:104-134).  gt: namespace with voxel [N,1,128^3] and, for joint training,
    spherical_object [N,1,160,160], depth, normal, silhou, depth_minmax."""
    opt = opt or GenReOptions()
    loss = F.binary_cross_entropy_with_logits(pred["pred_voxel"], gt.voxel)
    surface = F.binary_cross_entropy(torch.sigmoid(pred["pred_voxel"]) * gt.voxel, gt.voxel)
    loss = loss + surface * opt.surface_weight
    if joint:
        loss = loss + F.mse_loss(pred["pred_sph_full"], gt.spherical_object)
        fg = gt.silhou != 0
        loss = loss + F.mse_loss(pred["normal"][fg.expand_as(pred["normal"])], gt.normal[fg.expand_as(gt.normal)])
        loss = loss + F.mse_loss(pred["depth"][fg], gt.depth[fg]) + F.mse_loss(pred["silhou"], gt.silhou)
        loss = loss + (256 ** 2) / 2 * F.mse_loss(pred["depth_minmax"], gt.depth_minmax)
    return loss


class GenReInference:
    """the trimesh-free test entry (genre_full_model.py:175-186 with use_trimesh=False -> NetInterface.predict,
    netinterface.py:340-350): rgb + silhouette in, {'pred_voxel': ...} out, under no_grad.  With graph=True the whole
    forward of a fixed batch shape is captured once in a HIP graph and replayed (batch-1 latency is launch-bound:
    ~150 kernels of a few microseconds each)."""

    def __init__(self, net=None, device="cuda", graph=False, channels_last=True):
        """channels_last: the two 2-D U-ResNets (MarrNet-1, the inpainting network) keep their weights and activations in
        torch.channels_last.  MIOpen's fastest 2-D solvers on gfx950 are its igemm_*_nhwc kernels, which it wraps in
        batched_transpose launches when the tensors arrive NCHW (profiles/r05a_m1_b1_kernel_stats.txt: ~70 transposes per
        forward); NHWC tensors need none: 5.57 -> 5.45 ms per forward at batch 1, 22.3 -> 21.9 ms at batch 8
        (profiles/r05d_m1_rewrites_experiment.txt), logits equal to 3e-6.  Values, state_dict keys and shapes are unchanged (a
        memory format is a stride permutation); the 3-D refiner stays NCDHW (channels_last_3d measured 2x slower)"""
        self.net = (net or GenReNet()).to(device).eval()
        if channels_last and torch.device(device).type == "cuda":
            self.net.depth_and_inpaint.net1.to(memory_format=torch.channels_last)
            self.net.depth_and_inpaint.net2.to(memory_format=torch.channels_last)
        self.device = torch.device(device)
        self.graph = graph
        self._captured = {}

    def load(self, path):
        from .checkpoint import load_state_dict
        return load_state_dict(path, [self.net], None)

    @torch.no_grad()
    def predict(self, rgb, silhou):
        rgb, silhou = rgb.to(self.device), silhou.to(self.device)
        net = self.net
        if not self.graph:
            return {"pred_voxel": net(Inputs(rgb, silhou))["pred_voxel"]}
        key = tuple(rgb.shape)
        cap = self._captured.get(key)
        if cap is None:
            s_rgb, s_sil = rgb.clone(), silhou.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):                                   # warm-up: table builds, MIOpen find, allocations
                    net(Inputs(s_rgb, s_sil))
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = net(Inputs(s_rgb, s_sil))["pred_voxel"]
            # the graph holds raw pointers of every tensor the forward read, among them the cached fl / cam_dist
            # constants of the back-projection layers: pin them for the life of the graph
            from ..toolbox import _fused_render
            pinned = [t for m in net.modules() for t in getattr(m, "_consts", {}).values()]
            pinned.append(list(_fused_render._TABLES.values()))          # the renderer's geometry tables likewise
            cap = self._captured[key] = (g, s_rgb, s_sil, out, pinned)
        g, s_rgb, s_sil, out = cap[:4]
        s_rgb.copy_(rgb)
        s_sil.copy_(silhou)
        g.replay()
        return {"pred_voxel": out}
