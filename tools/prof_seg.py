"""rocprofv3 target: the standard-layout forward chain at batch 1 (graph replay) and batch 8 / 32 (eager), camera forward + segment
renderer with and without the occupancy words.  Usage: rocprofv3 --kernel-trace --stats -d DIR -- python tools/prof_seg.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import inputs
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render as F

dev = torch.device("cuda:0")
lib = F._loader().render_lib
mod = G.render_spherical().to(dev)
dirs = mod._dirs64.view(torch.float32)
layer = G.Camera_back_projection_layer().to(dev)
which = sys.argv[1:] or ["1", "8", "32"]
for B in [int(w) for w in which]:
    d = torch.from_numpy(inputs.batch_depth(B) if B > 1 else inputs.sphere_depth(noise_seed=2)).to(dev)
    with torch.no_grad():
        proj = layer(d)
        S = F.seg_tables_for(proj.shape, dev, mod._dirs64, mod.depth_weight)
        out = torch.empty((B, 1, 160, 160), device=dev)
        ps = torch.empty((B * S["smax"] * 128 * 128 * 2,), device=dev)
        occ, pe, cell = F.occupancy_hint_std(proj, S, mod._dirs64, mod.depth_weight, 50.0, lib)
        soft = (torch.rand(proj.shape, device=dev) * 0.9 + 0.05) * 0.02

        def hint():
            lib.render_seg_forward(proj, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, 50.0, None, occ, pe, cell)

        def dense():
            lib.render_seg_forward(soft, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, 50.0)

        def chain():
            mod(layer(d), pre_scale=50.0, pad=16)

        for fn in (hint, dense, chain):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            if B == 1:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    fn()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(10):
                        fn()
                for _ in range(5):
                    g.replay()
            else:
                for _ in range(10):
                    fn()
            torch.cuda.synchronize()
print("done")
