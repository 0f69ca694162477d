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
res = {}
mod = G.render_spherical(fused=True).to(dev)
dirs = mod._dirs64.view(torch.float32)
T = _fused_render.tables_for((1, 1, 128, 128, 128), dev, mod._dirs64, 256)
for B in (1, 8, 32):
    d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
    fl = torch.full((B, 1), 418.3, device=dev); cd = torch.full((B, 1), 2.2, device=dev)
    tdf = torch.empty((B, 1, 128, 128, 128), device=dev); cnt = torch.empty_like(tdf)
    cam_bp_lib.back_projection_forward(d, cd, fl, tdf, cnt)
    out = torch.empty((B, 1, 128, 128), device=dev); out2 = torch.empty_like(out); gout = torch.randn_like(out)
    vox = torch.clamp((1 - 128 * tdf) * 50, 1e-5, 1 - 1e-5)
    vbuf = torch.empty((B * 128 * 128 * 256,), device=dev)
    scratch = torch.empty((B * 128 * 128 * 256 + 4,), device=dev); gvox = torch.empty_like(vox); gvox2 = torch.empty_like(vox)
    res[f"B{B}_fwd_gather_us_per_img"] = round(timeit(lambda: lib.render_spherical_forward(vox, dirs, mod.depth_weight, out2)) / B, 1)
    res[f"B{B}_fwd_brick_us_per_img"] = round(timeit(lambda: lib.render_spherical_forward(vox, dirs, mod.depth_weight, out, vbuf, T["fwd_table"], T["fwd_chunks"], T["kin"])) / B, 1)
    res[f"B{B}_fwd_brick_vs_gather_maxabs"] = (out - out2).abs().max().item()
    res[f"B{B}_bwd_recompute_us_per_img"] = round(timeit(lambda: lib.render_spherical_backward(vox, dirs, mod.depth_weight, gout, gvox2, scratch, T["bwd_table"], T["bwd_chunks"])) / B, 1)
    res[f"B{B}_bwd_scan_us_per_img"] = round(timeit(lambda: lib.render_spherical_backward(vox, dirs, mod.depth_weight, gout, gvox, scratch, T["bwd_table"], T["bwd_chunks"], vbuf, T["kin"])) / B, 1)
    res[f"B{B}_bwd_scan_vs_recompute_maxrel"] = ((gvox - gvox2).abs() / (1 + gvox2.abs())).max().item()
print(json.dumps(res, indent=1))
