"""time the fused render forward / backward entry points at batch B (HIP events, back-to-back launches).
usage: python tools/time_render.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import inputs  # noqa: E402
import genre_shapehd_amd as G  # noqa: E402
from genre_shapehd_amd.toolbox import _fused_render  # noqa: E402
from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
fl = torch.full((B, 1), 418.3, device=dev)
cd = torch.full((B, 1), 2.2, device=dev)
tdf = torch.empty((B, 1, 128, 128, 128), device=dev)
cnt = torch.empty_like(tdf)
cam_bp_lib.back_projection_forward_shifted(d, cd, fl, tdf, cnt)
mod = G.render_spherical(fused=True).to(dev)
dirs = mod._dirs64.view(torch.float32)
T = _fused_render.tables_for(tdf.shape, dev, mod._dirs64, 256)
out = torch.empty((B, 1, 128, 128), device=dev)
gout = torch.randn_like(out)
vbuf = torch.empty((B * 128 * 128 * 256,), device=dev)
scratch = torch.empty((vbuf.numel() + max(4, B),), device=dev)
gvox = torch.empty_like(tdf)
lib = _fused_render._loader().render_lib


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


f = timeit(lambda: lib.render_spherical_forward(tdf, dirs, mod.depth_weight, out, vbuf, T["fwd_table"], T["fwd_chunks"],
                                                T["kin"], 50.0))
b = timeit(lambda: lib.render_spherical_backward(tdf, dirs, mod.depth_weight, gout, gvox, scratch, T["bwd_table"],
                                                 T["bwd_chunks"], vbuf, T["kin"], 50.0))
print("B=%d  forward %.1f us  backward %.1f us  checksum %.6e %.6e" % (
    B, f, b, out.double().sum().item(), gvox.double().abs().sum().item()))
dpv = scratch[: B * 128 * 128 * 256].view(B, 128 * 128, 256)
kin = T["kin"].view(1, -1, 1)
ks = torch.arange(256, device=dev).view(1, 1, -1)
involume = (ks >= kin).expand(B, -1, -1)
nz = (dpv != 0) & involume
print("in-volume samples per image %.0f, with dL/dp != 0: %.1f %%" % (involume.sum().item() / B, 100.0 * nz.sum().item() / involume.sum().item()))
