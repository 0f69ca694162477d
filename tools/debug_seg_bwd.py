import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render as F
dev = torch.device("cuda:0")
rng = np.random.default_rng(8)
n = 2
vox = torch.from_numpy(rng.uniform(0.001, 0.019, (n, 1, 128, 128, 128)).astype(np.float32)).to(dev)
if len(sys.argv) > 1 and sys.argv[1] == "sharp":
    import inputs
    from oracle.oracle import Oracle
    vox = torch.from_numpy((inputs.genre_offclamp_volumes(Oracle(), n) / np.float32(50.0)).astype(np.float32)).to(dev)
mod = G.render_spherical().to(dev)
lib = F._loader().render_lib
T = F.tables_for(vox.shape, dev, mod._dirs64, mod.z_res)
S = F.seg_tables_for(vox.shape, dev, mod._dirs64, mod.depth_weight)
dirs = mod._dirs64.view(torch.float32)
PAD = 16 if "pad" in sys.argv else 0
SIDE = 128 + 2 * PAD
g = torch.from_numpy(np.random.default_rng(15).standard_normal((8, 1, SIDE, SIDE)).astype(np.float32)[:n]).to(dev)
rays = n * 128 * 128
for scale in (0.0, 50.0):
    live = torch.empty((n * 513,), dtype=torch.int32, device=dev) if scale else None
    out = torch.empty((n, 1, SIDE, SIDE), device=dev)
    v_old = torch.zeros((rays * 256,), device=dev)
    lib.render_spherical_forward(vox, dirs, mod.depth_weight, out, v_old, T["fwd_table"], T["fwd_chunks"], T["kin"], scale, live)
    gv_old = torch.empty_like(vox); sc_old = torch.zeros((rays * 256 + n,), device=dev)
    lib.render_spherical_backward(vox, dirs, mod.depth_weight, g, gv_old, sc_old, T["bwd_table"], T["bwd_chunks"], v_old, T["kin"], scale, live)
    ps = torch.zeros((n * S["smax"] * 128 * 128 * 2,), device=dev)
    v_new = F.seg_v_scratch(S, n, dev)
    out2 = torch.empty_like(out)
    lib.render_seg_forward(vox, dirs, mod.depth_weight, out2, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, scale, live, None, None, 0, v_new)
    kin = T["kin"].long()
    k = torch.arange(256, device=dev)
    inside = (k[None, :] >= kin[:, None]).repeat(n, 1)
    print('scale', scale, 'map diff', (out - out2).abs().max().item())
    gv_new = torch.empty_like(vox); tr = F.seg_tr_scratch(ps, vox, mod._dirs64)
    lib.render_seg_backward(vox, dirs, mod.depth_weight, g, gv_new, S["bwd_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, tr, v_new, F.seg_halo_scratch(S, vox), scale, live)
    print("   grad diff max %.4g of %.4g" % ((gv_old - gv_new).abs().max().item(), gv_old.abs().max().item()))
    if len(sys.argv) > 1 and sys.argv[1] == "sharp" and scale:
        from oracle.torch_oracle import RenderSphericalExact
        x = vox[:1].cpu().clone().requires_grad_(True)
        o = RenderSphericalExact()(torch.clamp(x * scale, 1e-5, 1 - 1e-5))
        if PAD:
            from oracle.torch_oracle import sph_pad
            o = sph_pad(o, PAD)
        o.backward(g[:1].cpu().to(o.dtype))
        ex = x.grad[0, 0]
        for name, gv in (("old", gv_old), ("new", gv_new)):
            err = (gv[0, 0].cpu().double() - ex).abs() / ex.abs().clamp(min=scale)
            i = err.argmax().item()
            ix, iy, iz = i // 16384, (i // 128) % 128, i % 128
            print("   ", name, "vs exact: max rel err %.3g at voxel" % err.max().item(), (ix, iy, iz), "exact %.6f got %.6f old %.6f" %
                  (ex[ix, iy, iz].item(), gv[0, 0, ix, iy, iz].item(), gv_old[0, 0, ix, iy, iz].item()), "value", vox[0, 0, ix, iy, iz].item() * scale)
            top = torch.topk(err.flatten(), 8)
            print("      top errors", [("%.2e" % v) for v in top.values.tolist()], [(j // 16384, (j // 128) % 128, j % 128) for j in top.indices.tolist()])
