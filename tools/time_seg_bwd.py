"""times genre_render_seg_backward on a soft volume (every sample passes the clamps) at the given batch sizes; with
GENRE_HIP_LIB pointing at a variant build (tools/build_variants.sh) an A/B of seg_scatter_kernel"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render as F
dev = torch.device("cuda:0")
lib = F._loader().render_lib
mod = G.render_spherical().to(dev)
dirs = mod._dirs64.view(torch.float32)
for B in [int(a) for a in (sys.argv[1:] or ["32"])]:
    g = torch.Generator(device="cpu").manual_seed(1)
    vox = ((torch.rand((B, 1, 128, 128, 128), generator=g) * 0.9 + 0.05) * 0.02).to(dev)
    S = F.seg_tables_for(vox.shape, dev, mod._dirs64, mod.depth_weight)
    out = torch.empty((B, 1, 160, 160), device=dev); gout = torch.randn_like(out)
    ps = torch.empty((B * S["smax"] * 128 * 128 * 2,), device=dev); tr = F.seg_tr_scratch(ps, vox, mod._dirs64)
    v = F.seg_v_scratch(S, B, dev)
    live = torch.empty((B * 513,), dtype=torch.int32, device=dev)
    gv = torch.empty_like(vox)
    halo = F.seg_halo_scratch(S, vox)
    lib.render_seg_forward(vox, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, 50.0, live, None, None, 0, v)
    def bwd():
        lib.render_seg_backward(vox, dirs, mod.depth_weight, gout, gv, S["bwd_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, tr, v, halo, 50.0, live)
    for _ in range(3): bwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): bwd()
    e1.record(); torch.cuda.synchronize()
    print("batch", B, "seg backward %.1f us" % (e0.elapsed_time(e1) * 100), os.environ.get("GENRE_HIP_LIB", "(shipped)"), flush=True)
