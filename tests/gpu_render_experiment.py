import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import inputs
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render
from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
dev = torch.device("cuda:0")
lib = _fused_render._loader().render_lib
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
B = 8
d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
fl = torch.full((B, 1), 418.3, device=dev); cd = torch.full((B, 1), 2.2, device=dev)
tdf = torch.empty((B, 1, 128, 128, 128), device=dev); cnt = torch.empty_like(tdf)
cam_bp_lib.back_projection_forward(d, cd, fl, tdf, cnt)
mod = G.render_spherical(fused=True).to(dev)
dirs = mod._dirs64.view(torch.float32)
out = torch.empty((B, 1, 128, 128), device=dev); gout = torch.randn_like(out)
fields = {"genre": torch.clamp((1 - 128 * tdf) * 50, 1e-5, 1 - 1e-5),
          "random": torch.rand_like(tdf) * 0.05 + 0.001,
          "zeros": torch.zeros_like(tdf)}
res = {}
for name, vox in fields.items():
    gvox = torch.empty_like(vox)
    res[name + "_fwd_us_per_img"] = timeit(lambda: lib.render_spherical_forward(vox, dirs, mod.depth_weight, out)) / B
    for dbg in (0, 1, 2, 3):
        os.environ["GENRE_DBG"] = str(dbg)
        res[f"{name}_bwd_dbg{dbg}_us_per_img"] = timeit(lambda: lib.render_spherical_backward(vox, dirs, mod.depth_weight, gout, gvox)) / B
    os.environ["GENRE_DBG"] = "0"
print(json.dumps(res, indent=1))
