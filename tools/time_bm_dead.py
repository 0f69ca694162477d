"""the image-minor sampler's launch with EVERY tile dead (occupancy words all zero): what 8 343 workgroups cost that only copy
constants -- dispatch + one dependent load chain each -- against the shipped mix (61 % dead) and the dense launch"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, inputs
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render as F
B = 32
dev = torch.device("cuda:0")
lib = F._loader().render_lib
mod = G.render_spherical().to(dev)
layer = G.Camera_back_projection_layer(batch_minor=True).to(dev)
d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
with torch.no_grad():
    proj = layer(d)
TB = F.bm_tables_for(proj.shape, dev, mod._dirs64, mod.depth_weight)
ps = torch.empty((TB["segs"].shape[0] * 64,), device=dev)
out = torch.empty((B, 1, 160, 160), device=dev)
words, pe = F.occupancy_hint(proj, TB, 50.0, lib, with_grad=False)
zero = torch.zeros_like(words)
def ev(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
def run(w):
    lib.render_bm_forward(proj, out, TB["segs"], TB["rec_f"], TB["fwd_rows"], TB["ray_ptr"], TB["ray_seg"], TB["ray_pre"], ps, None, None, 50.0,
                          w, pe if w is not None else None)
print("rows", TB["fwd_rows"].shape[0], "live fraction %.3f" % (words != 0).float().mean().item())
print("all dead %.1f us | shipped mix %.1f us | dense %.1f us   (sampler + per-ray pass ~22 us)" % (ev(lambda: run(zero)), ev(lambda: run(words)), ev(lambda: run(None))))
