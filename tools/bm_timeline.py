"""Per-workgroup timeline of bm_sample_kernel (variant library: tools/build_variants.sh sph_render_bm.hip tlb:"-DGENRE_BM_TIMELINE";
GENRE_HIP_LIB points the loader at it).  Marks (s_memrealtime, 10 ns): 0 row loaded | 1 occupancy word + headers decided |
2 tile loads issued, records arrived | 3 LDS tile stored | 4 barrier passed | 6 march done | 7 (dead tile) constants copied.
Usage: GENRE_HIP_LIB=tools/variants/libgenre_hip_tlb.so python tools/bm_timeline.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import inputs
import genre_shapehd_amd as G
from genre_shapehd_amd.toolbox import _fused_render as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
lib = F._loader().render_lib
mod = G.render_spherical().to(dev)
layer = G.Camera_back_projection_layer(batch_minor=True).to(dev)
d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
with torch.no_grad():
    proj = layer(d)
TB = F.bm_tables_for(proj.shape, dev, mod._dirs64, mod.depth_weight)
groups = -(-B // 32)
ps = torch.empty((groups * TB["segs"].shape[0] * 64,), device=dev)
stash = torch.empty((groups * TB["rec_f"].shape[0] * 32,), device=dev)
mask = torch.empty((groups * 128 ** 3 + groups,), dtype=torch.int32, device=dev)
out = torch.empty((B, 1, 160, 160), device=dev)
words, pe = F.occupancy_hint(proj, TB, 50.0, lib, with_grad=True)
soft = F.empty_batch_minor(proj.shape, torch.float32, dev)
soft.copy_((torch.rand(proj.shape, device=dev) * 0.9 + 0.05) * 0.02)
path = "/tmp/bm_tl.bin"
for name, vol, hint, save in (("genre+hint, saved state", proj, True, True), ("genre+hint, inference", proj, True, False),
                              ("genre dense", proj, False, True), ("soft", soft, False, True)):
    def run():
        lib.render_bm_forward(vol, out, TB["segs"], TB["rec_f"], TB["fwd_rows"], TB["ray_ptr"], TB["ray_seg"], TB["ray_pre"], ps,
                              stash if save else None, mask if save else None, 50.0, words if hint else None, pe if hint else None)
    os.environ.pop("GENRE_BM_TIMELINE", None)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    os.environ["GENRE_BM_TIMELINE"] = path
    run()
    torch.cuda.synchronize()
    raw = np.fromfile(path, dtype=np.uint8)
    gx, gy, _, nt = np.frombuffer(raw[:16], np.int32)
    t = np.frombuffer(raw[16:], np.uint64).reshape(gy, gx, 8).astype(np.float64)
    t0 = t[t > 0].min()
    us = np.where(t > 0, (t - t0) / 100.0, np.nan)
    live = ~np.isnan(us[..., 6])
    dead = ~np.isnan(us[..., 7])
    print("== %s: grid %d x %d, NT %d; %d marched, %d dead tiles; kernel span %.1f us" % (name, gx, gy, nt, live.sum(), dead.sum(), np.nanmax(us)))
    if live.any():
        f = us[live]
        for a, b, nm in ((0, 1, "row -> word+headers"), (1, 2, "-> tile issued, records arrived"), (2, 3, "-> LDS stored"),
                         (3, 4, "-> barrier passed"), (4, 6, "-> march done (all segments)")):
            dlt = f[:, b] - f[:, a]
            dlt = dlt[~np.isnan(dlt)]
            if len(dlt):
                print("   %-34s median %.2f  p90 %.2f  max %.2f us" % (nm, np.median(dlt), np.percentile(dlt, 90), dlt.max()))
        tot = f[:, 6] - f[:, 0]
        print("   live workgroup total               median %.2f  p90 %.2f  max %.2f us" % (np.median(tot), np.percentile(tot, 90), tot.max()))
    if dead.any():
        f = us[dead]
        print("   dead: row -> word %.2f, -> constants copied %.2f (median)  total p90 %.2f us" % (
            np.median(f[:, 1] - f[:, 0]), np.median(f[:, 7] - f[:, 1]), np.percentile(f[:, 7] - f[:, 0], 90)))
    # residency: workgroups in flight over time
    start = us[..., 0][~np.isnan(us[..., 0])]
    end = np.where(np.isnan(us[..., 6]), us[..., 7], us[..., 6])
    end = end[~np.isnan(end)]
    grid_t = np.linspace(0, np.nanmax(us), 200)
    infl = [(start <= x).sum() - (end <= x).sum() for x in grid_t]
    print("   workgroups in flight: median %d  max %d  (256 CUs)" % (np.median(infl), max(infl)))
