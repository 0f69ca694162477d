"""Ad-hoc first timing of the kernels (HIP events on torch's current stream)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import inputs
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
from genre_shapehd_amd.toolbox.calc_prob.calc_prob._ext import calc_prob_lib

dev = torch.device("cuda:0")
def timeit(fn, iters=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters   # us

res = {}
for B in (1, 8, 32):
    d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
    fl = torch.full((B, 1), 418.3, device=dev); cd = torch.full((B, 1), 2.2, device=dev)
    tdf = torch.empty((B, 1, 128, 128, 128), device=dev); cnt = torch.empty_like(tdf)
    us = timeit(lambda: cam_bp_lib.back_projection_forward(d, cd, fl, tdf, cnt))
    res[f"cam_bp_fwd_B{B}_us"] = us; res[f"cam_bp_fwd_B{B}_GBs"] = B * 17039360 / us / 1e3
    p = torch.rand((B, 1, 128, 128, 256), device=dev).clamp_(1e-5, 1 - 1e-5)
    s = torch.empty_like(p)
    us = timeit(lambda: calc_prob_lib.calc_prob_forward(p, s))
    res[f"calc_prob_fwd_B{B}_us"] = us; res[f"calc_prob_fwd_B{B}_GBs"] = B * 33554432 / us / 1e3
    g = torch.randn_like(p); o = torch.empty_like(p)
    us = timeit(lambda: calc_prob_lib.calc_prob_backward_fused(p, s, g, o))
    res[f"calc_prob_bwd_fused_B{B}_us"] = us; res[f"calc_prob_bwd_fused_B{B}_GBs"] = B * 67108864 / us / 1e3
    gi = torch.randn_like(tdf); gd = torch.empty_like(d); gf = torch.empty_like(fl); gc = torch.empty_like(fl)
    us = timeit(lambda: cam_bp_lib.back_projection_backward(d, fl, cd, cnt, gi, gd, gc, gf))
    res[f"cam_bp_bwd_B{B}_us"] = us
    x1 = torch.rand((B, 2048, 3), device=dev); x2 = torch.rand((B, 2048, 3), device=dev)
    us = timeit(lambda: G.nndistance_w_idx(x1, x2))
    res[f"nnd_fwd_B{B}_us"] = us; res[f"nnd_fwd_B{B}_TFLOPs"] = B * 67.1e6 / us / 1e6
    del p, s, g, o, tdf, cnt, gi
# raw copy ceiling
x = torch.empty(256 * 1024 * 1024 // 4, device=dev); y = torch.empty_like(x)
us = timeit(lambda: y.copy_(x), 50, 5)
res["copy_256MiB_GBs"] = 2 * x.numel() * 4 / us / 1e3
print(json.dumps(res, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "quick_timing.json"), "w"), indent=1)
