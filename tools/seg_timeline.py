"""Per-workgroup timeline of seg_sample_kernel (variant library built with -DGENRE_SEG_TIMELINE: tools/build_variants.sh
sph_render_seg.hip tl:"-DGENRE_SEG_TIMELINE"; GENRE_HIP_LIB points the loader at it).  Marks (s_memrealtime, 10 ns):
0 row loaded | 1 occupancy decided | 2 all loads issued, segment entry arrived | 3 LDS tile stored | 4 barrier passed |
5 march of the last chunk done | 6 end.   Usage: GENRE_HIP_LIB=tools/variants/libgenre_hip_tl.so python tools/seg_timeline.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import inputs
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
lib = F._loader().render_lib
mod = G.render_spherical().to(dev)
dirs = mod._dirs64.view(torch.float32)
layer = G.Camera_back_projection_layer().to(dev)
d = torch.from_numpy(inputs.batch_depth(B) if B > 1 else inputs.sphere_depth(noise_seed=2)).to(dev)
with torch.no_grad():
    proj = layer(d)
S = F.seg_tables_for(proj.shape, dev, mod._dirs64, mod.depth_weight)
out = torch.empty((B, 1, 160, 160), device=dev)
ps = torch.empty((B * S["smax"] * 128 * 128 * 2,), device=dev)
occ, pe, cell = F.occupancy_hint_std(proj, S, mod._dirs64, mod.depth_weight, 50.0, lib)
path = "/tmp/seg_tl.bin"
for name, args in (("hint", (occ, pe, cell)), ("dense", (None, None, 0))):
    os.environ.pop("GENRE_SEG_TIMELINE", None)
    for _ in range(5):
        lib.render_seg_forward(proj, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, 50.0, None, *args)
    torch.cuda.synchronize()
    os.environ["GENRE_SEG_TIMELINE"] = path
    lib.render_seg_forward(proj, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps, 50.0, None, *args)
    torch.cuda.synchronize()
    raw = np.fromfile(path, dtype=np.uint8)
    gx, gy, g_, nt = np.frombuffer(raw[:16], np.int32)
    t = np.frombuffer(raw[16:], np.uint64).reshape(gy, gx, 8).astype(np.float64)
    t0 = t[t > 0].min()
    us = np.where(t > 0, (t - t0) / 100.0, np.nan)                       # 100 MHz -> us
    full = ~np.isnan(us[..., 6]) & ~np.isnan(us[..., 4])                 # workgroups that marched
    early = ~np.isnan(us[..., 0]) & np.isnan(us[..., 4])                  # dead / empty: returned before the barrier
    print("== %s: grid %d x %d, G %d, NT %d; %d marched, %d returned early" % (name, gx, gy, g_, nt, full.sum(), early.sum()))
    print("   kernel span (first mark -> last mark): %.2f us" % np.nanmax(us))
    print("   mark 0 (row loaded): first %.2f  median %.2f  last %.2f us" % (np.nanmin(us[..., 0]), np.nanmedian(us[..., 0]), np.nanmax(us[..., 0])))
    if full.any():
        f = us[full]
        names = ["row->occ", "occ->entry arrived", "entry->LDS stored", "LDS->barrier", "barrier->march done", "march->end"]
        for i, nm in enumerate(names):
            dlt = f[:, i + 1] - f[:, i]
            print("   %-22s median %.2f  p90 %.2f  max %.2f us" % (nm, np.median(dlt), np.percentile(dlt, 90), dlt.max()))
        tot = f[:, 6] - f[:, 0]
        print("   workgroup total        median %.2f  p90 %.2f  max %.2f us;  end of last workgroup %.2f us" % (np.median(tot), np.percentile(tot, 90), tot.max(), f[:, 6].max()))
