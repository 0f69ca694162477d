"""A/B harness for one GPU call (run on the MI355X box from the repo root):

  python tools/ab_round2.py                 drive everything below, one subprocess per library variant / mode
  python tools/ab_round2.py --worker bm TAG time render_bm forward / backward at batch 32 with the library installed in
                                            csrc/, and compare outputs with the first variant's (kept in /tmp)
  python tools/ab_round2.py --worker cam    time the camera forward at batch 1, 2, 4, 8 (HIP-graph replay) under the
                                            GENRE_CAMBP_MODE of the environment

  python tools/ab_round2.py --cam-variants  the cam worker (brick mode) once per variant library
  python tools/ab_round2.py --no-cam        only the render_bm workers

Variant libraries come from tools/build_variants.sh (tools/variants/libgenre_hip_<name>.so, not tracked): one source
recompiled with experiment macros.  This harness produced the A/B numbers quoted in csrc/sph_render_bm.hip,
csrc/cam_bp.hip and DESIGN.md (round 2); a whole call is ~30 s of GPU time."""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "genre-shapehd_amd", "csrc", "libgenre_hip.so")
VAR = os.path.join(ROOT, "tools", "variants")


def event_us(fn, iters=80, warm=5):
    import torch
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


def worker_bm(tag):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import inputs
    import genre_shapehd_amd as G
    from genre_shapehd_amd.toolbox import _fused_render
    dev = torch.device("cuda:0")
    B = 32
    lib = _fused_render._loader().render_lib
    mod = G.render_spherical(fused=True).to(dev)
    d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
    layer = G.Camera_back_projection_layer(batch_minor=True).to(dev)
    with torch.no_grad():
        proj = layer(d)
    TB = _fused_render.bm_tables_for(proj.shape, dev, mod._dirs64, mod.depth_weight)
    groups = -(-B // 32)
    ps = torch.empty((groups * TB["segs"].shape[0] * 64,), device=dev)
    tr = torch.empty_like(ps)
    stash = torch.empty((groups * TB["rec_f"].shape[0] * 32,), device=dev)
    mask = torch.empty((groups * 128 ** 3 + groups,), dtype=torch.int32, device=dev)
    out = torch.empty((B, 1, 160, 160), device=dev)
    torch.manual_seed(0)
    gout = torch.randn_like(out)
    gvox = _fused_render.empty_batch_minor(proj.shape, torch.float32, dev)

    def fwd(vol, save, scale):
        lib.render_bm_forward(vol, out, TB["segs"], TB["rec_f"], TB["fwd_rows"], TB["ray_ptr"], TB["ray_seg"],
                              TB["ray_pre"], ps, stash if save else None, mask if (save and scale) else None, scale)

    def bwd(scale):
        lib.render_bm_backward(gout, gvox, TB["segs"], TB["ray_ptr"], TB["ray_seg"], TB["ray_pre"], TB["ent"], TB["rec_b"],
                               TB["bwd_rows"], mod.depth_weight, ps, tr, stash, mask if scale else None, scale,
                               TB["pull_code"])
    res = {}
    # correctness material: a volume whose samples pass the clamp (gradient flows everywhere), with and without pre_scale
    g = torch.Generator(device="cpu").manual_seed(1)
    soft = _fused_render.empty_batch_minor(proj.shape, torch.float32, dev)
    soft.copy_((torch.rand(proj.shape, generator=g) * 0.9 + 0.05).to(dev))
    outs = {}
    for name, vol, scale in (("soft", soft, 0.0), ("soft50", soft * 0.02, 50.0), ("proj50", proj, 50.0)):
        fwd(vol, True, scale)
        bwd(scale)
        torch.cuda.synchronize()
        outs[name + "_out"] = out.clone().cpu()
        outs[name + "_gv"] = gvox.clone().cpu()
    ref_path = "/tmp/ab_bm_ref.pt"
    if not os.path.exists(ref_path):
        torch.save(outs, ref_path)
        res["ref"] = "saved"
    else:
        ref = torch.load(ref_path)
        for k, v in outs.items():
            r = ref[k]
            den = max(1e-30, r.abs().max().item())
            res["d_" + k] = "%.2e (max %.2e, nan %d)" % ((v - r).abs().max().item() / den, den, int(torch.isnan(v).sum()))
    fwd(proj, True, 50.0)
    res["fwd_save_us"] = round(event_us(lambda: fwd(proj, True, 50.0)), 1)
    res["fwd_nosave_us"] = round(event_us(lambda: fwd(proj, False, 50.0)), 1)
    fwd(proj, True, 50.0)
    res["bwd_us"] = round(event_us(lambda: bwd(50.0)), 1)
    print("AB bm %s %s" % (tag, res), flush=True)


def worker_cam():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import inputs
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    dev = torch.device("cuda:0")
    res = {}
    for B in (1, 2, 4, 8, 32):
        d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
        fl = torch.full((B, 1), 418.3, device=dev)
        cd = torch.full((B, 1), 2.2, device=dev)
        tdf = torch.empty((B, 1, 128, 128, 128), device=dev)
        cnt = torch.empty_like(tdf)
        reps = 20

        def body():
            for _ in range(reps):
                cam_bp_lib.back_projection_forward(d, cd, fl, tdf, cnt)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body()
        res["b%d_us_per_image" % B] = round(event_us(graph.replay, 30, 3) / reps / B, 2)
    # the M2 pair at batch 1: cam_bp fwd + calc_prob fwd, 20 x in one graph
    from genre_shapehd_amd.toolbox.calc_prob.calc_prob._ext import calc_prob_lib
    d = torch.from_numpy(inputs.batch_depth(1)).to(dev)
    fl = torch.full((1, 1), 418.3, device=dev)
    cd = torch.full((1, 1), 2.2, device=dev)
    tdf = torch.empty((1, 1, 128, 128, 128), device=dev)
    cnt = torch.empty_like(tdf)
    p = torch.rand((1, 1, 128, 128, 256), device=dev).clamp_(1e-5, 1 - 1e-5)
    s = torch.empty_like(p)

    def pair():
        for _ in range(20):
            cam_bp_lib.back_projection_forward(d, cd, fl, tdf, cnt)
            calc_prob_lib.calc_prob_forward(p, s)
    pair()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        pair()
    res["m2_b1_us"] = round(event_us(graph.replay, 50, 5) / 20, 2)
    print("AB cam %s mode=%s %s" % (sys.argv[3] if len(sys.argv) > 3 else "", os.environ.get("GENRE_CAMBP_MODE", "auto"), res),
          flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker_bm(sys.argv[3]) if sys.argv[2] == "bm" else worker_cam()
    keep = "/tmp/libgenre_hip_default.so"
    shutil.copy(LIB, keep)
    if os.path.exists("/tmp/ab_bm_ref.pt"):
        os.remove("/tmp/ab_bm_ref.pt")
    try:
        if "--cam-variants" in sys.argv:
            for name in sorted(f[len("libgenre_hip_"):-3] for f in os.listdir(VAR) if f.endswith(".so")):
                shutil.copy(os.path.join(VAR, "libgenre_hip_%s.so" % name), LIB)
                subprocess.run([sys.executable, __file__, "--worker", "cam", name],
                               env=dict(os.environ, GENRE_CAMBP_MODE="brick"), timeout=300)
            return
        for mode in (() if "--no-cam" in sys.argv else ("scatter", "brick")):
            t0 = time.time()
            subprocess.run([sys.executable, __file__, "--worker", "cam"], env=dict(os.environ, GENRE_CAMBP_MODE=mode),
                           timeout=300)
            print("  (%.0f s)" % (time.time() - t0), flush=True)
        names = sorted(f[len("libgenre_hip_"):-3] for f in os.listdir(VAR) if f.endswith(".so"))
        for name in names:
            shutil.copy(os.path.join(VAR, "libgenre_hip_%s.so" % name), LIB)
            t0 = time.time()
            subprocess.run([sys.executable, __file__, "--worker", "bm", name], timeout=300)
            print("  (%.0f s)" % (time.time() - t0), flush=True)
    finally:
        shutil.copy(keep, LIB)


if __name__ == "__main__":
    main()
