"""Timing of the standard-layout forward, old (per-sample scratch + scan) against new (per-segment pairs), with and without the
camera forward's occupancy words; batch 1 from a HIP graph, batch 8 / 32 with HIP events.  Usage: python tools/time_render_seg.py"""
import os
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import inputs
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render as F

dev = torch.device("cuda:0")
lib = F._loader().render_lib
mod = G.render_spherical().to(dev)
dirs = mod._dirs64.view(torch.float32)


def ev(fn, iters=50, warm=5):
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


def graphed(fn, reps=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    return ev(g.replay, 20, 3) / reps


res = {}
layer = G.Camera_back_projection_layer().to(dev)
for B in (1, 8, 32):
    d = torch.from_numpy(inputs.batch_depth(B) if B > 1 else inputs.sphere_depth(noise_seed=2)).to(dev)
    with torch.no_grad():
        proj = layer(d)
    soft = (torch.rand(proj.shape, device=dev) * 0.9 + 0.05) * 0.02
    T = F.tables_for(proj.shape, dev, mod._dirs64, mod.z_res)
    S = F.seg_tables_for(proj.shape, dev, mod._dirs64, mod.depth_weight)
    out = torch.empty((B, 1, 160, 160), device=dev)
    v = torch.empty((B * 128 * 128 * 256,), device=dev)
    ps = torch.empty((B * S["smax"] * 128 * 128 * 2,), device=dev)
    live = torch.empty((B * 513,), dtype=torch.int32, device=dev)
    occ, pe, cell = F.occupancy_hint_std(proj, S, mod._dirs64, mod.depth_weight, 50.0, lib)
    tm = graphed if B == 1 else ev
    r = {}
    r["old_genre"] = tm(lambda: lib.render_spherical_forward(proj, dirs, mod.depth_weight, out, v, T["fwd_table"], T["fwd_chunks"], T["kin"], 50.0, live))
    r["old_soft"] = tm(lambda: lib.render_spherical_forward(soft, dirs, mod.depth_weight, out, v, T["fwd_table"], T["fwd_chunks"], T["kin"], 50.0, live))
    for cfg in (("default", None),):
        r["seg_genre_hint"] = tm(lambda: lib.render_seg_forward(proj, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, 50.0, live, occ, pe, cell))
        r["seg_genre_hint_nolive"] = tm(lambda: lib.render_seg_forward(proj, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, 50.0, None, occ, pe, cell))
        r["seg_genre_dense"] = tm(lambda: lib.render_seg_forward(proj, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, 50.0, live))
        r["seg_soft"] = tm(lambda: lib.render_seg_forward(soft, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, 50.0, live))
    r["tiles_occupied_cells_frac"] = float((occ != 0).float().mean()) if occ is not None else None
    r["cfg"] = os.environ.get("GENRE_SEG_CFG", "default")
    res["batch%d" % B] = r
    print("batch", B, json.dumps(r), flush=True)
print(json.dumps(res))
