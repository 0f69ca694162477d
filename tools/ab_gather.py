"""A/B of library variants of the gather backward in ONE GPU call (run from the repo root on the MI355X box):
   python tools/ab_gather.py            every tools/variants/libgenre_hip_g*.so in turn (copied over csrc/libgenre_hip.so,
                                        one subprocess each), then the regular library again
   python tools/ab_gather.py --worker TAG
Variants: tools/build_variants.sh sph_render_bm.hip gXX:"-DGENRE_G_ABL=..." ..."""
import glob
import os
import shutil
import subprocess
import sys

os.environ.setdefault("GENRE_BM_BWD", "gather")          # the gather tables are only built when it is selected
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "genre-shapehd_amd", "csrc", "libgenre_hip.so")


def worker(tag):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import inputs
    import genre_shapehd_amd as G
    from genre_shapehd_amd.toolbox import _fused_render
    dev = torch.device("cuda:0")
    B = 32
    d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
    layer = G.Camera_back_projection_layer(batch_minor=True).to(dev)
    mod = G.render_spherical(fused=True).to(dev)
    with torch.no_grad():
        proj = layer(d)
    T = _fused_render.bm_tables_for(proj.shape, dev, mod._dirs64, mod.depth_weight)
    lib = _fused_render._loader().render_lib
    f32 = dict(dtype=torch.float32, device=dev)
    ps = torch.empty((T["segs"].shape[0] * 64,), **f32)
    tr = torch.empty((ps.numel() + 64,), **f32)
    stash = torch.empty((T["rec_f"].shape[0] * 32,), **f32)
    mask = torch.empty((128 ** 3 + 1,), dtype=torch.int32, device=dev)
    out = torch.empty((B, 1, 160, 160), **f32)
    gout = torch.randn_like(out)
    gb = _fused_render.empty_batch_minor(proj.shape, torch.float32, dev)
    x = _fused_render.empty_batch_minor(proj.shape, torch.float32, dev)
    x.copy_(torch.rand((B, 1, 128, 128, 128), device=dev) * 0.02)

    def timeit(fn, iters=40, warm=5):
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

    res = []
    for name, vol, pre, m in (("bench", proj, 50.0, mask), ("soft", x, 0.0, None)):
        lib.render_bm_forward(vol, out, T["segs"], T["rec_f"], T["fwd_rows"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], ps, stash, m, pre)
        res.append("%s %.1f" % (name, timeit(lambda: lib.render_bm_backward_gather(
            gout, gb, T["segs"], T["ray_ptr"], T["ray_seg"], T["ray_pre"], T["g_ent"], T["g_chunks"], T["g_blob"], T["g_rows"],
            mod.depth_weight, ps, tr, stash, m, pre))))
    print("AB gather %-12s %s us (whole backward: combine + zero + gather)   checksum %.6e" % (
        tag, "  ".join(res), gb.double().abs().sum().item()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(sys.argv[2])
        sys.exit(0)
    keep = LIB + ".orig"
    shutil.copy(LIB, keep)
    try:
        subprocess.call([sys.executable, __file__, "--worker", "regular"])
        for path in sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "libgenre_hip_g*.so"))):
            shutil.copy(path, LIB)
            tag = os.path.basename(path)[len("libgenre_hip_"):-3]
            subprocess.call([sys.executable, __file__, "--worker", tag])
    finally:
        shutil.copy(keep, LIB)
        os.remove(keep)
