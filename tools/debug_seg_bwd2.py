import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import genre_shapehd_amd as G
import inputs
from oracle.oracle import Oracle
from oracle.torch_oracle import RenderSphericalExact, sph_pad
dev = torch.device("cuda:0")
vols_all = inputs.genre_offclamp_volumes(Oracle(), 8)
mod = G.render_spherical(fused=True).to(dev)
for n, pad, scale in ((8, 16, 50.0), (8, 0, 50.0), (2, 16, 50.0), (8, 16, None)):
    vols = vols_all[:n].copy()
    if scale: vols = (vols / np.float32(scale)).astype(np.float32)
    rng = np.random.default_rng(7 + 8)
    side = 128 + 2 * pad
    g = rng.standard_normal((n, 1, side, side)).astype(np.float32)
    x = torch.from_numpy(vols).to(dev).requires_grad_(True)
    out = mod(x, pre_scale=scale, pad=pad)
    out.backward(torch.from_numpy(g).to(dev))
    xe = torch.from_numpy(vols[:1]).requires_grad_(True)
    v = xe if scale is None else torch.clamp(xe * scale, 1e-5, 1 - 1e-5)
    o = RenderSphericalExact()(v)
    if pad: o = sph_pad(o, pad)
    o.backward(torch.from_numpy(g[:1]).to(o.dtype))
    ex = xe.grad[0, 0]
    got = x.grad[0, 0].cpu().double()
    s = scale or 1.0
    err = (got - ex).abs() / ex.abs().clamp(min=s)
    top = torch.topk(err.flatten(), 6)
    print("n", n, "pad", pad, "scale", scale, "map err %.3g" % (out[0].cpu().double() - o[0]).abs().max().item(), "max rel err %.3g" % err.max().item(),
          [("%.2e" % v) for v in top.values.tolist()], [(j // 16384, (j // 128) % 128, j % 128) for j in top.indices.tolist()])
    j = top.indices[0].item(); ix, iy, iz = j // 16384, (j // 128) % 128, j % 128
    print("    exact %.6f got %.6f value %.8f" % (ex[ix, iy, iz].item(), got[ix, iy, iz].item(), vols[0, 0, ix, iy, iz] * s))
