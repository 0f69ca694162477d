"""A/B of the two batch-minor backward forms at batch B (HIP events, back-to-back launches): scatter (LDS fp64 atomics)
vs gather (register sums over contribution lists), and their difference on the bench's volume.
usage: python tools/time_bm_bwd.py [B]"""
import os
import sys

os.environ.setdefault("GENRE_BM_BWD", "gather")          # the gather tables are only built when it is selected

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import inputs  # noqa: E402
import genre_shapehd_amd as G  # noqa: E402
from genre_shapehd_amd.toolbox import _fused_render  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
layer = G.Camera_back_projection_layer(batch_minor=True).to(dev)
mod = G.render_spherical(fused=True).to(dev)
with torch.no_grad():
    proj = layer(d)
T = _fused_render.bm_tables_for(proj.shape, dev, mod._dirs64, mod.depth_weight)
lib = _fused_render._loader().render_lib
groups = -(-B // 32)
f32 = dict(dtype=torch.float32, device=dev)
ps = torch.empty((groups * T["segs"].shape[0] * 64,), **f32)
tr = torch.empty((ps.numel() + 64,), **f32)
stash = torch.empty((groups * T["rec_f"].shape[0] * 32,), **f32)
mask = torch.empty((groups * 128 ** 3 + groups,), dtype=torch.int32, device=dev)
out = torch.empty((B, 1, 160, 160), **f32)
gout = torch.randn_like(out)
ga = _fused_render.empty_batch_minor(proj.shape, torch.float32, dev)
gb = _fused_render.empty_batch_minor(proj.shape, torch.float32, dev)


def timeit(fn, iters=30, warm=3):
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


lib.render_bm_forward(proj, out, T["segs"], T["rec_f"], T["fwd_rows"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], ps, stash, mask, 50.0)


def scatter():
    lib.render_bm_backward(gout, ga, T["segs"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], T["ent"], T["rec_b"],
                           T["bwd_rows"], mod.depth_weight, ps, tr, stash, mask, 50.0, T["pull_code"])


def gather():
    lib.render_bm_backward_gather(gout, gb, T["segs"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], T["g_ent"], T["g_chunks"],
                                  T["g_blob"], T["g_rows"], mod.depth_weight, ps, tr, stash, mask, 50.0)


ts, tg = timeit(scatter), timeit(gather)
diff = (ga - gb).abs().max().item()
print("AB bm_bwd B=%d scatter %.1f us gather %.1f us  max|diff| %.3e  max|grad| %.3e" % (B, ts, tg, diff, ga.abs().max().item()))
# a volume whose gradient is not identically zero (off the clamp bounds): uniform random occupancy
x = _fused_render.empty_batch_minor(proj.shape, torch.float32, dev)
x.copy_(torch.rand((B, 1, 128, 128, 128), device=dev) * 0.02)
lib.render_bm_forward(x, out, T["segs"], T["rec_f"], T["fwd_rows"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], ps, stash, None, 0.0)


def scatter0():
    lib.render_bm_backward(gout, ga, T["segs"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], T["ent"], T["rec_b"],
                           T["bwd_rows"], mod.depth_weight, ps, tr, stash, None, 0.0, T["pull_code"])


def gather0():
    lib.render_bm_backward_gather(gout, gb, T["segs"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], T["g_ent"], T["g_chunks"],
                                  T["g_blob"], T["g_rows"], mod.depth_weight, ps, tr, stash, None, 0.0)


ts, tg = timeit(scatter0), timeit(gather0)
rel = ((ga - gb).abs() / ga.abs().clamp(min=1e-3 * ga.abs().max().item())).max().item()
print("AB bm_bwd(no pre_scale, soft volume) scatter %.1f us gather %.1f us  max rel diff %.3e  max|grad| %.3e" % (ts, tg, rel, ga.abs().max().item()))
# the same soft volume through the pre_scale code path (pre_scale = 1: clamp mask written and applied): code path vs data
lib.render_bm_forward(x, out, T["segs"], T["rec_f"], T["fwd_rows"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], ps, stash, mask, 1.0)


def scatter1():
    lib.render_bm_backward(gout, ga, T["segs"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], T["ent"], T["rec_b"],
                           T["bwd_rows"], mod.depth_weight, ps, tr, stash, mask, 1.0, T["pull_code"])


def gather1():
    lib.render_bm_backward_gather(gout, gb, T["segs"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], T["g_ent"], T["g_chunks"],
                                  T["g_blob"], T["g_rows"], mod.depth_weight, ps, tr, stash, mask, 1.0)


print("AB bm_bwd(pre_scale 1, soft volume) scatter %.1f us gather %.1f us" % (timeit(scatter1), timeit(gather1)))
