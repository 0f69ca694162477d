import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, inputs
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render
from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
dev = torch.device("cuda:0")
lib = _fused_render._loader().render_lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
fl = torch.full((B, 1), 418.3, device=dev); cd = torch.full((B, 1), 2.2, device=dev)
tdf = torch.empty((B, 1, 128, 128, 128), device=dev); cnt = torch.empty_like(tdf)
cam_bp_lib.back_projection_forward(d, cd, fl, tdf, cnt)
mod = G.render_spherical(fused=True).to(dev)
dirs = mod._dirs64.view(torch.float32)
out = torch.empty((B, 1, 128, 128), device=dev); gout = torch.randn_like(out)
vox = torch.clamp((1 - 128 * tdf) * 50, 1e-5, 1 - 1e-5)
T = _fused_render.tables_for(vox.shape, dev, mod._dirs64, 256)
vbuf = torch.empty((B * 128 * 128 * 256,), device=dev)
scratch = torch.empty((B * 128 * 128 * 256 + 4,), device=dev); gvox = torch.empty_like(vox)
for _ in range(reps):
    lib.render_spherical_forward(vox, dirs, mod.depth_weight, out, vbuf, T["fwd_table"], T["fwd_chunks"], T["kin"])
    lib.render_spherical_backward(vox, dirs, mod.depth_weight, gout, gvox, scratch, T["bwd_table"], T["bwd_chunks"], vbuf, T["kin"])
torch.cuda.synchronize()
