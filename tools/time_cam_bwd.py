"""camera backward at batch 32, both layouts (A/B of the launch geometry: GENRE_CAM_BWD_NARROW=1 keeps 256-thread workgroups)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, inputs
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render as F
from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
dev = torch.device("cuda:0")
B = 32
d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
fl = torch.full((B, 1), 418.3, device=dev); cd = torch.full((B, 1), 2.2, device=dev)
gd = torch.empty_like(d); gfl = torch.empty((B, 1), device=dev); gcd = torch.empty((B, 1), device=dev)
def ev(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters
for name, bm in (("std", False), ("bm", True)):
    shape = (B, 1, 128, 128, 128)
    mk = (lambda: F.empty_batch_minor(shape, torch.float32, dev)) if bm else (lambda: torch.empty(shape, device=dev))
    proj, cnt, g = mk(), mk(), mk()
    cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, proj, cnt, shifted=True)
    g.copy_(torch.randn(shape, device=dev))
    zero = mk(); zero.zero_()
    print(name, "random grad %.1f us" % ev(lambda: cam_bp_lib.back_projection_backward_shifted(d, fl, cd, cnt, g, gd, gcd, gfl)),
          "zero grad %.1f us" % ev(lambda: cam_bp_lib.back_projection_backward_shifted(d, fl, cd, cnt, zero, gd, gcd, gfl)))
