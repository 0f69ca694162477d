"""debug: a slab through the centre of the cube -- fused (standard kernels) vs the reference op sequence on the GPU vs the
batch-minor renderer; run-to-run determinism"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import genre_shapehd_amd as G
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
# depth sheet like the failing test: abs depth 2.1..2.3 over a disc
ax = np.linspace(-1, 1, 256)
sil = ((ax[:, None] ** 2 + ax[None, :] ** 2) < 0.5)
pd = np.where(rng.random((256, 256)) < 0.5, 0.4, 0.6)
d = np.where(sil, (1 - pd) * (2.5 - 1.9 + 1e-4) + 1.9, 0).astype(np.float32)[None, None]      # two interleaved planes
dt = torch.from_numpy(d).to(dev)
layer = G.Camera_back_projection_layer().to(dev)
p1, p2 = layer(dt), layer(dt)
print("cam_bp run-to-run max diff", (p1 - p2).abs().max().item(), "nonzero voxels", (p1 != 0).sum().item())
fused, ref = G.render_spherical(fused=True).to(dev), G.render_spherical(fused=False).to(dev)
for ps, pad in ((50.0, 16), (None, 0)):
    x = p1 if ps else torch.clamp(p1 * 50, 1e-5, 1 - 1e-5)
    a, a2 = fused(x, pre_scale=ps, pad=pad), fused(x, pre_scale=ps, pad=pad)
    r = ref(x, pre_scale=ps, pad=pad)
    print("pre_scale", ps, "pad", pad, ": fused run-to-run", (a - a2).abs().max().item(), " fused vs op sequence", (a - r).abs().max().item())
    x16 = x.expand(16, -1, -1, -1, -1)
    bm = torch.empty_strided(tuple(x16.shape), (1, 16 * 128 ** 3, 128 * 128 * 16, 128 * 16, 16), dtype=x.dtype, device=dev)
    bm.copy_(x16)
    b = fused(bm, pre_scale=ps, pad=pad)
    print("    batch-minor vs op sequence", (b[0] - r[0]).abs().max().item(), (b[15] - r[0]).abs().max().item())
    s8 = fused(x.expand(8, -1, -1, -1, -1).contiguous(), pre_scale=ps, pad=pad)
    print("    std batch 8 vs op sequence", (s8[3] - r[0]).abs().max().item())
