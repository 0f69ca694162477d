"""Round-4 A/B harness for one GPU call (run on the MI355X box from the repo root):

  python tools/ab_round4.py                    every library in tools/variants/ (tools/build_variants.sh), one subprocess each
  python tools/ab_round4.py --worker m2 TAG    batch-1 M2 pair (cam_bp by value + calc_prob forward, HIP-graph replay), each
                                               kernel alone, the same two at batch 32; outputs compared with the first variant's
  python tools/ab_round4.py --worker bm TAG    batch-minor renderer at batch 32: forward / backward on the GenRe volume (all
                                               clamp masks zero) and on the `soft` volume (gradient everywhere, pre_scale 50)

Variant names starting with `sc` run the bm worker, all others the m2 worker; `base` runs both."""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "genre-shapehd_amd", "csrc", "libgenre_hip.so")
VAR = os.path.join(ROOT, "tools", "variants")


def event_us(fn, iters=50, warm=5):
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


def graph_of(body):
    import torch
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    return g


def worker_m2(tag):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch
    import inputs
    import genre_shapehd_amd  # noqa: F401
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    from genre_shapehd_amd.toolbox.calc_prob.calc_prob._ext import calc_prob_lib
    dev = torch.device("cuda:0")
    res = {}
    d = torch.from_numpy(inputs.sphere_depth(noise_seed=2)).to(dev)
    tdf = torch.empty((1, 1, 128, 128, 128), device=dev)
    cnt = torch.empty_like(tdf)
    torch.manual_seed(0)
    p = torch.rand((1, 1, 128, 128, 256), device=dev).clamp_(1e-5, 1 - 1e-5)
    s = torch.empty_like(p)
    reps = 20

    def cam():
        cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, tdf, cnt)

    def cp():
        calc_prob_lib.calc_prob_forward(p, s)

    def pair():
        for _ in range(reps):
            cam()
            cp()
    gp = graph_of(pair)
    gc = graph_of(lambda: [cam() for _ in range(reps)])
    gs = graph_of(lambda: [cp() for _ in range(reps)])
    best = lambda g: min(event_us(g.replay, 40, 5) for _ in range(3)) / reps      # noqa: E731
    res["m2_b1_us"] = round(best(gp), 3)
    res["cam_b1_us"] = round(best(gc), 3)
    res["cp_b1_us"] = round(best(gs), 3)
    res["frac"] = round(50593792 / res["m2_b1_us"] / 1e3 / 8000.0, 4)
    torch.cuda.synchronize()
    outs = {"tdf": tdf.clone().cpu(), "cnt": cnt.clone().cpu(), "s": s.clone().cpu()}
    B = 32
    d32 = torch.from_numpy(inputs.batch_depth(B)).to(dev)
    t32 = torch.empty((B, 1, 128, 128, 128), device=dev)
    c32 = torch.empty_like(t32)
    p32 = torch.rand((B, 1, 128, 128, 256), device=dev).clamp_(1e-5, 1 - 1e-5)
    s32 = torch.empty_like(p32)
    res["cam_b32_us"] = round(min(event_us(lambda: cam_bp_lib.back_projection_forward_const(d32, 2.2, 418.3, t32, c32), 20, 3) for _ in range(2)), 1)
    res["cp_b32_us"] = round(min(event_us(lambda: calc_prob_lib.calc_prob_forward(p32, s32), 20, 3) for _ in range(2)), 1)
    res["m2_b32_frac"] = round(B * 50593792 / (res["cam_b32_us"] + res["cp_b32_us"]) / 1e3 / 8000.0, 4)
    outs["t32"] = t32[::8].clone().cpu()
    ref_path = "/tmp/ab4_m2_ref.pt"
    if not os.path.exists(ref_path):
        torch.save(outs, ref_path)
        res["ref"] = "saved"
    else:
        ref = torch.load(ref_path)
        res["same"] = all(torch.equal(outs[k], ref[k]) for k in outs)
    print("AB4 m2 %-8s %s" % (tag, res), flush=True)


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
    g = torch.Generator(device="cpu").manual_seed(1)
    soft = _fused_render.empty_batch_minor(proj.shape, torch.float32, dev)
    soft.copy_((torch.rand(proj.shape, generator=g) * 0.9 + 0.05).to(dev))
    soft50 = soft * 0.02
    res, outs = {}, {}
    for name, vol, scale in (("soft", soft, 0.0), ("soft50", soft50, 50.0), ("genre", proj, 50.0)):
        fwd(vol, True, scale)
        bwd(scale)
        torch.cuda.synchronize()
        outs[name + "_out"] = out.clone().cpu()
        outs[name + "_gv"] = gvox.clone().cpu()
        res[name + "_fwd_us"] = round(event_us(lambda: fwd(vol, True, scale), 30, 3), 1)
        fwd(vol, True, scale)
        res[name + "_bwd_us"] = round(min(event_us(lambda: bwd(scale), 30, 3) for _ in range(2)), 1)
    res["max|g| soft50"] = float(outs["soft50_gv"].abs().max())
    res["max|g| genre"] = float(outs["genre_gv"].abs().max())
    ref_path = "/tmp/ab4_bm_ref.pt"
    if not os.path.exists(ref_path):
        torch.save(outs, ref_path)
        res["ref"] = "saved"
    else:
        ref = torch.load(ref_path)
        for k, v in outs.items():
            r = ref[k]
            den = max(1e-30, r.abs().max().item())
            res["d_" + k] = "%.1e" % ((v - r).abs().max().item() / den)
    print("AB4 bm %-8s %s" % (tag, res), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        return worker_bm(sys.argv[3]) if sys.argv[2] == "bm" else worker_m2(sys.argv[3])
    keep = "/tmp/libgenre_hip_default.so"
    shutil.copy(LIB, keep)
    for f in ("/tmp/ab4_m2_ref.pt", "/tmp/ab4_bm_ref.pt"):
        if os.path.exists(f):
            os.remove(f)
    names = sorted(f[len("libgenre_hip_"):-3] for f in os.listdir(VAR) if f.endswith(".so"))
    names = ["base"] + [n for n in names if n != "base"]
    try:
        for name in names:
            shutil.copy(os.path.join(VAR, "libgenre_hip_%s.so" % name), LIB)
            workers = ["m2", "bm"] if name == "base" else (["bm"] if name.startswith("sc") else ["m2"])
            for w in workers:
                t0 = time.time()
                subprocess.run([sys.executable, __file__, "--worker", w, name], timeout=400)
                print("  (%.0f s)" % (time.time() - t0), flush=True)
        if "--cam-tests" in sys.argv:              # the pixel-screen variant through the camera parity tests
            shutil.copy(os.path.join(VAR, "libgenre_hip_pxs.so"), LIB)
            subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_cam_bp.py", "tests/test_gpu_golden.py", "tests/test_gpu_fuzz.py",
                            "-q", "-m", "gpu", "-k", "camera or cam or layer or fuzz"], cwd=ROOT, timeout=600)
    finally:
        shutil.copy(keep, LIB)


if __name__ == "__main__":
    main()
