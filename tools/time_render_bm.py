"""time the batch-minor tile renderer's entry points at batch B (HIP events, back-to-back launches), and the whole
autograd chain cam_bp -> render(pre_scale, pad) forward + backward in both layouts.
usage: python tools/time_render_bm.py [B]"""
import os
import sys

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
assert proj.stride(0) == 1
T = _fused_render.bm_tables_for(proj.shape, dev, mod._dirs64, mod.depth_weight)
lib = _fused_render._loader().render_lib
groups = -(-B // 32)
f32 = dict(dtype=torch.float32, device=dev)
ps = torch.empty((groups * T["segs"].shape[0] * 64,), **f32)
tr = torch.empty_like(ps)
stash = torch.empty((groups * T["rec_f"].shape[0] * 32,), **f32)
mask = torch.empty((groups * 128 ** 3 + groups,), dtype=torch.int32, device=dev)
out = torch.empty((B, 1, 160, 160), **f32)
gout = torch.randn_like(out)
gvox = _fused_render.empty_batch_minor(proj.shape, torch.float32, dev)


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


def fwd(save):
    lib.render_bm_forward(proj, out, T["segs"], T["rec_f"], T["fwd_rows"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], ps,
                          stash if save else None, mask if save else None, 50.0)


def bwd():
    lib.render_bm_backward(gout, gvox, T["segs"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], T["ent"], T["rec_b"],
                           T["bwd_rows"], mod.depth_weight, ps, tr, stash, mask, 50.0, T["pull_code"])


print("B=%d  forward(no save) %.1f us  forward(save) %.1f us  backward %.1f us" % (
    B, timeit(lambda: fwd(False)), timeit(lambda: fwd(True)), timeit(bwd)))
print("checksums %.6e %.6e" % (out.double().sum().item(), gvox.double().abs().sum().item()))

for bm in (True, False):
    lay = G.Camera_back_projection_layer(batch_minor=bm).to(dev)
    dd = d.clone().requires_grad_(True)
    g = torch.randn((B, 1, 160, 160), device=dev)

    def step():
        dd.grad = None
        o = mod(lay(dd), pre_scale=50.0, pad=16)
        o.backward(g)
    print("chain fwd+bwd, batch_minor=%s: %.1f us" % (bm, timeit(step, iters=10)))
