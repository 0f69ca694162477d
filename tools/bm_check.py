"""batch-minor path vs the standard path on the same logical tensor (development check)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import genre_shapehd_amd as G

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(0)
vox = torch.from_numpy(rng.uniform(0, 0.05, (B, 1, 128, 128, 128)).astype(np.float32)).to(dev)


def batch_minor(t):
    n, c, x, y, z = t.shape
    out = torch.empty_strided((n, c, x, y, z), (1, n * x * y * z, y * z * n, z * n, n), dtype=t.dtype, device=t.device)
    out.copy_(t)
    return out


def timeit(fn, iters=10, warm=2):
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


vbm = batch_minor(vox)
assert vbm.stride(0) == 1 and torch.equal(vbm, vox)
mod = G.render_spherical().to(dev)
with torch.no_grad():
    for ps, pad in ((None, 0), (20.0, 16)):
        ref = mod(vox, pre_scale=ps, pad=pad)
        got = mod(vbm, pre_scale=ps, pad=pad)
        print("pre_scale", ps, "pad", pad, "max |diff|", (ref - got).abs().max().item(), "bit-equal", torch.equal(ref, got))
    print("forward standard %.1f us   batch-minor %.1f us" % (timeit(lambda: mod(vox, pre_scale=20.0, pad=16)),
                                                              timeit(lambda: mod(vbm, pre_scale=20.0, pad=16))))
if len(sys.argv) > 2:
    g = torch.from_numpy(rng.standard_normal((B, 1, 160, 160)).astype(np.float32)).to(dev)
    a = vox.clone().requires_grad_(True)
    b = batch_minor(vox).requires_grad_(True)
    mod(a, pre_scale=20.0, pad=16).backward(g)
    mod(b, pre_scale=20.0, pad=16).backward(g)
    sc = a.grad.abs().max().item()
    print("grad max |diff| / max", (a.grad - b.grad).abs().max().item() / sc, "grad layout", b.grad.stride())

    def fb(v):
        v.grad = None
        mod(v, pre_scale=20.0, pad=16).backward(g)
    print("fwd+bwd standard %.1f us   batch-minor %.1f us" % (timeit(lambda: fb(a)), timeit(lambda: fb(b))))
