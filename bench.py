#!/usr/bin/env python3
"""bench.py -- hot-path throughput on MI355X (contract: see the task statement / DESIGN.md).

A "step" is ONE pass of BASELINE.json configs[1] over one batch of synthetic depth maps that is
already resident in HBM:

    depth [B,1,256,256] -> cam_bp fwd -> 128^3 TDF -> shift, x50, clamp -> render_spherical
    (trilinear sampling of 128x128 rays x 256 samples, calc_prob stop-probability scan, depth
    expectation) -> sph_pad -> [B,1,160,160]; then the backward of the same chain down to
    grad_depth (calc_prob bwd, sampling bwd, cam_bp bwd).

`value` = depth maps ("shapes") per second through that forward+backward chain, summed over
all ranks (weak scaling: every rank owns its own batch; the path needs no collective).
Besides the contract keys the JSON line carries
  roofline     : the dominant hand-written kernel of the step against the 8 TB/s HBM roof
                 (ALGORITHMIC bytes / measured kernel time, HIP events on the launch stream)
  m2           : BASELINE's second metric, cam_bp fwd + calc_prob fwd bytes / their time
  batch1       : the same at batch 1 (launch-bound regime; replayed from a HIP graph)
  cpu_baseline : the same step on the host (reference kernel bodies host-compiled, or the C
                 port), bounded sample, rank 0 only
  m1           : GenRe full-model forward passes per second (BASELINE's first metric), batch 1 and 8
  train        : one optimizer step of BASELINE configs[3] / configs[4] at their per-rank shard shapes (DDP when N > 1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import miopen_cache  # noqa: E402

miopen_cache.use()          # compiled MIOpen kernels of the networks (m1, train), if tools/warm_miopen.py left them in the tree
import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md: dense fp32 matrix peak (the networks of `m1` run fp32 on MIOpen)
BYTES_CAM_FWD = 256 * 256 * 4 + 2 * 128 ** 3 * 4          # depth + tdf + cnt          = 17 039 360
BYTES_CP_FWD = 2 * 128 * 128 * 256 * 4                    # prob in + stop out         = 33 554 432
BYTES_CP_BWD_FUSED = 4 * 128 * 128 * 256 * 4              # p, s, grad in + grad out   = 67 108 864
BYTES_RENDER_FUSED = 128 ** 3 * 4 + 128 * 128 * 4         # vox in + map out           =  8 454 144


def source_sha(files=None):
    """sha256 of kernel sources: identifies the code a PMC table was measured on (the GPU box has no .git).
    files=None: one hash over all of csrc/*.hip|*.hpp; else {file: hash} for the named files"""
    import hashlib
    csrc = os.path.join(ROOT, "genre-shapehd_amd", "csrc")
    names = sorted(n for n in os.listdir(csrc) if n.endswith((".hip", ".hpp")))
    if files is not None:
        out = {}
        for n in files:
            with open(os.path.join(csrc, n), "rb") as f:
                out[n] = hashlib.sha256(f.read()).hexdigest()[:16]
        return out
    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel_names, batch, sources=("common.hpp",)):
    """HBM bytes per launch of a kernel group from the newest committed PMC table (profiles/*_pmc_hbm_traffic.json,
    written by profiles/collect_pmc.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, corrected as
    MI355X_MICROARCH.md prescribes).  None -- never a stale number -- unless the table was measured on exactly the
    source files that define these kernels (`sources`) and on this batch size."""
    import glob
    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")))
    if not tables:
        return None, None
    with open(tables[-1]) as f:
        t = json.load(f)
    name = os.path.basename(tables[-1])
    mine = source_sha(sources)
    theirs = t.get("source_sha_by_file", {})
    stale = [n for n in sources if theirs.get(n) != mine[n]]
    if stale or t.get("batch") != batch:
        return None, "%s was measured on other versions of %s / batch %s" % (name, stale, t.get("batch"))
    total = 0.0
    for k in kernel_names:
        row = t["kernels"].get(k)
        if row is None:
            return None, "%s has no row for %s" % (name, k)
        total += row["hbm_bytes"]
    return total, "profiles/" + name


def rocprof_m2():
    """the two M2 kernels in the newest committed rocprofv3 trace (profiles/*_kernel_stats_phases.txt: profiles/pmc_targets.py, the
    same inputs, but each launch there follows calc_prob's 2 GB backward stream instead of its own kind) -- the figure a reader
    of the trace finds, beside the event figures above"""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_stats_phases.txt")))
    out = {"what": "rocprofv3 kernel trace of profiles/pmc_targets.py (same inputs; every launch there follows other kernels' streams, "
                   "not its own kind back to back)"}
    if not files:
        return out
    out["file"] = "profiles/" + os.path.basename(files[-1])
    for key, pat in (("cam_brick_kernel<false, true>", "cam_brick_kernel<false, true>"), ("stop_fwd_vec4_kernel<true>", "stop_fwd_vec4_kernel<true>")):
        for line in open(files[-1]):
            if pat in line:
                m = re.search(r"\)?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+[\d.]+\s+\d+\s+\d+\s+\d+\s*$", line)
                if m:
                    out[key] = {"calls": int(m.group(1)), "avg_us": float(m.group(3)), "min_us": float(m.group(4)), "max_us": float(m.group(5))}
                break
    return out


def miopen_find_db_state():
    """`shipped` when MIOpen reads its find-db from the tree (genre-shapehd_amd/.miopen/db/*.txt, tracked since round 5: the measured
    solver choices of tools/warm_miopen.py -- worth 154 -> 180 forward passes/s in round 4), `absent` when a clone has none and
    MIOpen ranks solvers by its built-in estimates"""
    import glob
    db = os.environ.get("MIOPEN_USER_DB_PATH")
    n = len(glob.glob(os.path.join(db, "*.ufdb.txt"))) if db else 0
    return {"state": "shipped" if n else "absent", "path": os.path.relpath(db, ROOT) if db else None,
            "kernel_cache": "present" if glob.glob(os.path.join(os.environ.get("MIOPEN_CUSTOM_CACHE_DIR", "/nonexistent"), "*.ukdb")) else
                            "absent (MIOpen compiles its kernels on first use: minutes, outside every timed region)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="depth maps per GPU per step")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="force the reference op sequence in render_spherical")
    ap.add_argument("--layout", choices=["bm", "std"], default="bm",
                    help="memory order of the projected volume between cam_bp and the renderer: bm = image index fastest "
                         "(batch-minor tile renderer, batches >= 16), std = the reference's NCXYZ")
    ap.add_argument("--no-m1", action="store_true", help="skip the GenRe whole-model forward (M1)")
    ap.add_argument("--no-train", action="store_true", help="skip the configs[3]/[4] train-step timings (`train`)")
    ap.add_argument("--train-steps", type=int, default=8, help="timed optimizer steps per train config")
    ap.add_argument("--train-configs", default="all",
                    help="comma-separated subset of shapehd,wgangp,genre (or `all`, the default): the train steps to time")
    ap.add_argument("--eager", action="store_true", help="time eager launches of the step instead of a HIP-graph replay")
    ap.add_argument("--stub", action="store_true", help="launcher self-test: a trivial CPU step over gloo, no GPU")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) without a torchrun environment: re-execute under torch.distributed.run, one
    rank per GPU, and hand its exit code back"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def stub_main(args):
    """the launcher path with a trivial CPU step (tests/test_bench_launcher.py): init from the torchrun environment,
    barrier-bracketed timing, max over ranks, one JSON line from rank 0"""
    from importlib import util as _u
    spec = _u.spec_from_file_location("dist_utils", os.path.join(ROOT, "genre-shapehd_amd", "dist_utils.py"))
    du = _u.module_from_spec(spec)
    spec.loader.exec_module(du)
    rank, _, world = du.env_rank_world()
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    dist = du.init_from_env("gloo")
    x = torch.ones(64, 64)
    # the hot path's sharding and per-rank bookkeeping as main() does it: every rank owns a contiguous shard of the job's
    # world * batch items (no data-path collective), times its own loop, and the line carries the max-over-ranks figure beside
    # one figure per rank
    lo, hi = du.shard_bounds(world * args.batch, rank, world)
    du.fence(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = (x @ x).clamp_(max=1.0)
    du.fence(dist)
    local = time.perf_counter() - t0
    el = du.max_over_ranks(dist, local)
    per_rank_t = du.per_rank(dist, local)
    shards = du.per_rank(dist, float(lo * (1 << 20) + hi))                # (two small ints through the float64 all-gather)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": world * args.batch * args.steps / el, "unit": "shapes/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "stub": True, "scaling": "weak",
                          "rccl_ranks": world, "ms_per_step": el * 1e3 / args.steps,
                          "hot_path": {"batch_per_gpu": args.batch,
                                       "per_rank_shapes_per_s": [args.batch * args.steps / t for t in per_rank_t],
                                       "shards": [[int(v) >> 20, int(v) & ((1 << 20) - 1)] for v in shards]}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


class HotPath(torch.nn.Module):
    """configs[1] with the product ops (depth_pred_with_sph_inpaint.py:120-126)."""

    def __init__(self, G, fused, batch_minor=False):
        super().__init__()
        self.G = G
        self.cam = G.Camera_back_projection_layer(batch_minor=batch_minor)
        self.render = G.render_spherical(fused=fused)

    def forward(self, depth):
        proj = self.cam(depth)                                        # fl=418.3, cam_dist=2.2, 1-128*tdf
        # == sph_pad(render(clamp(proj*50, 1e-5, 1-1e-5)), 16)  (:124-126); the fused renderer folds both in, the
        # reference-op-sequence mode runs them as separate torch ops
        return self.render(proj, pre_scale=50.0, pad=16)


def event_time_us(fn, iters, warm, min_seconds=0.5):
    """average duration of fn() in microseconds, HIP events on the current (launch) stream; the measured loop runs for
    at least min_seconds (so that an SMI sampler sees the GPU busy and clocks settle)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if min_seconds:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            fn()
        b.record()
        torch.cuda.synchronize()
        est = max(a.elapsed_time(b) / 3 * 1e-3, 1e-7)
        iters = int(min(max(iters, min_seconds / est), 20000))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def kernel_table(G, dev, B):
    """time the hand-written kernels of the step in isolation (back-to-back launches)"""
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    from genre_shapehd_amd.toolbox.calc_prob.calc_prob._ext import calc_prob_lib
    import inputs
    d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
    fl = torch.full((B, 1), 418.3, device=dev)
    cd = torch.full((B, 1), 2.2, device=dev)
    tdf = torch.empty((B, 1, 128, 128, 128), device=dev)
    cnt = torch.empty_like(tdf)
    p = torch.rand((B, 1, 128, 128, 256), device=dev).clamp_(1e-5, 1 - 1e-5)
    s = torch.empty_like(p)
    g = torch.randn_like(p)
    o = torch.empty_like(p)
    iters = 20
    rows = {}
    t = event_time_us(lambda: cam_bp_lib.back_projection_forward(d, cd, fl, tdf, cnt), iters, 5)
    rows["cam_bp_fwd"] = dict(us=t, bytes=B * BYTES_CAM_FWD, kernels="cam_brick_kernel (dense NCXYZ outputs: one launch)",
                              pmc=["cam_brick_kernel"], src=("common.hpp", "cam_bp.hip"))
    t = event_time_us(lambda: calc_prob_lib.calc_prob_forward(p, s), iters, 5)
    rows["calc_prob_fwd"] = dict(us=t, bytes=B * BYTES_CP_FWD, kernels="stop_fwd_vec4_kernel", pmc=["stop_fwd_vec4_kernel<true>" if B * BYTES_CP_FWD // 2 > (128 << 20) else "stop_fwd_vec4_kernel<false>"],
                                 src=("common.hpp", "wave_scan.hpp", "calc_prob.hip"))
    # the M2 pair as a pair: cam_bp forward and calc_prob forward ALTERNATING on one stream, each finding its lines evicted by the
    # other's 0.5 / 1 GB (what a rocprofv3 trace of profiles/pmc_targets.py sees: VERDICT r5 weak 8 -- the back-to-back figure of
    # cam_brick_kernel, 120 us, sits below the trace's 130-140 us)
    def m2_pair():
        cam_bp_lib.back_projection_forward(d, cd, fl, tdf, cnt)
        calc_prob_lib.calc_prob_forward(p, s)
    rows["m2_pair_alternating"] = dict(us=event_time_us(m2_pair, iters, 5), bytes=B * (BYTES_CAM_FWD + BYTES_CP_FWD),
                                       kernels="cam_brick_kernel, stop_fwd_vec4_kernel alternating on one stream")
    t = event_time_us(lambda: calc_prob_lib.calc_prob_backward_fused(p, s, g, o), iters, 5)
    rows["calc_prob_bwd_fused"] = dict(us=t, bytes=B * BYTES_CP_BWD_FUSED, kernels="stop_bwd_vec4_kernel<fused>",
                                       pmc=["stop_bwd_vec4_kernel<true>"], src=("common.hpp", "wave_scan.hpp", "calc_prob.hip"))
    from genre_shapehd_amd.toolbox import _fused_render
    fused_ok = _fused_render.available()
    if fused_ok:
        render_lib = _fused_render._loader().render_lib
        mod = G.render_spherical(fused=True).to(dev)
        dirs = mod._dirs64.view(torch.float32)
        out = torch.empty((B, 1, 128, 128), device=dev)
        gout = torch.randn_like(out)
        gvox = torch.empty_like(tdf)
        T = _fused_render.tables_for(tdf.shape, dev, mod._dirs64, mod.z_res)
        vbuf = torch.empty((B * 128 * 128 * mod.z_res,), device=dev)
        scratch = torch.empty((vbuf.numel() + max(4, B),), device=dev)
        live = torch.empty((B * (1 + 512),), dtype=torch.int32, device=dev)      # the clamp's pass words: what autograd passes
        # STANDARD LAYOUT (NCXYZ: what every BASELINE config below 16 images per GPU runs).  The volume as the step's layer hands
        # it over: dense, with the camera brick kernel's occupancy words (one per image and 8x8x32-voxel cell); the segment
        # forward (csrc/sph_render_seg.hip, round 6) copies constants for tiles it knows to be empty instead of reading them
        layer_std = G.Camera_back_projection_layer().to(dev)
        with torch.no_grad():
            proj = layer_std(d)
        S = _fused_render.seg_tables_for(proj.shape, dev, mod._dirs64, mod.depth_weight)
        ps_std = torch.empty((B * S["smax"] * 128 * 128 * 2,), device=dev)
        occ, pe_std, cell = _fused_render.occupancy_hint_std(proj, S, mod._dirs64, mod.depth_weight, 50.0, render_lib, with_grad=True)
        tiles_live_std = 1.0
        if occ is not None:          # share of the 16^3 tiles (brick + high halo) that overlap an occupied cell: what is read
            o8 = (occ != 0).float().view(B, 1, 16, 16, 4)
            zc = torch.stack([o8[..., [0, 0, 1, 1, 2, 2, 3, 3]], o8[..., [0, 1, 1, 2, 2, 3, 3, 3]]]).amax(0)      # z: tile bz -> cells bz//2, (bz+1)//2
            t8 = torch.nn.functional.max_pool3d(torch.nn.functional.pad(zc, (0, 0, 0, 1, 0, 1)), (3, 3, 1), stride=(2, 2, 1))
            tiles_live_std = float((t8 > 0).float().mean())

        def seg_fwd(vol, hint=True, lv=live):
            render_lib.render_seg_forward(vol, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"],
                                          ps_std, 50.0, lv, *((occ, pe_std, cell) if hint else (None, None, 0)))
        seg_pmc = ["seg_sample_kernel<true, false, true>", "seg_combine_kernel<256>"]       # (pmc_targets.py: with saved samples)
        rows["render_fwd_fused"] = dict(us=event_time_us(lambda: seg_fwd(proj), iters, 5), bytes=B * BYTES_RENDER_FUSED,
                                        bytes_needed=int(B * (tiles_live_std * 128 ** 3 * 4 + 128 * 128 * 4)),
                                        tiles_live_frac=tiles_live_std,
                                        kernels="seg_sample_kernel+seg_combine_kernel on GenRe's volume, %.0f %% of the tiles live "
                                                "(occupancy words of the camera brick kernel)" % (100 * tiles_live_std),
                                        pmc=[k + "@genre" for k in seg_pmc], src=("common.hpp", "render_common.hpp", "sph_render_seg.hip"))
        rows["render_fwd_fused_dense"] = dict(us=event_time_us(lambda: seg_fwd(proj, False), iters, 5), bytes=B * BYTES_RENDER_FUSED,
                                              kernels="the same volume WITHOUT the occupancy words: every tile is read",
                                              pmc=["seg_sample_kernel<true, false, false>@dense", "seg_combine_kernel<256>@dense"],
                                              src=("common.hpp", "render_common.hpp", "sph_render_seg.hip"))
        bwd_pmc = ["seg_combine_bwd_kernel<256, 8>", "seg_scatter_kernel<%d>" % (256 if B >= 16 else 512), "seg_halo_kernel"]
        bwd_src = ("common.hpp", "render_common.hpp", "sph_render_seg.hip")
        tr_std = _fused_render.seg_tr_scratch(ps_std, proj, mod._dirs64)
        halo_std = _fused_render.seg_halo_scratch(S, proj)
        vseg = _fused_render.seg_v_scratch(S, B, dev)                               # saved sample values: 16 floats per segment

        def seg_fwd_grad(vol, hint=True):
            # the forward as autograd runs it when a gradient is wanted: + the raw sample values of the tiles a gradient can
            # come back through (GenRe's volume: none)
            render_lib.render_seg_forward(vol, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"],
                                          ps_std, 50.0, live, *((occ, pe_std, cell) if hint else (None, None, 0)), vseg)

        def std_bwd(vol, lv):
            # per-ray chains, then one pass over the segments (dL/dp from the saved values, scattered into the bricks' tiles)
            render_lib.render_seg_backward(vol, dirs, mod.depth_weight, gout, gvox, S["bwd_rows"], S["segs"], S["ray_nseg"],
                                           S["ray_pre"], S["line_w"], ps_std, tr_std, vseg, halo_std, 50.0, lv)
        # GenRe's own volume: the clamp blocks every voxel, the group writes grad_vox = 0 (billed with the bytes it moves) ...
        rows["render_fwd_fused_grad"] = dict(us=event_time_us(lambda: seg_fwd_grad(proj), iters, 5), bytes=B * BYTES_RENDER_FUSED,
                                             kernels="the same with a gradient wanted (GenRe's volume: no tile's samples are saved)")
        seg_fwd_grad(proj)
        rows["render_bwd_fused"] = dict(us=event_time_us(lambda: std_bwd(proj, live), iters, 5), bytes=B * 128 ** 3 * 4,
                                        kernels="seg_combine_bwd_kernel (returns at once) + seg_scatter_kernel (every row writes its "
                                                "brick's zeros) + seg_halo_kernel (returns) on GenRe's volume (the clamp blocks every voxel)",
                                        pmc=[k + "@genre" for k in bwd_pmc], src=bwd_src)
        # ... and the same kernels where they do work: the soft volume (every sample passes the clamps)
        gs = torch.Generator(device="cpu").manual_seed(1)
        soft = ((torch.rand(proj.shape, generator=gs) * 0.9 + 0.05) * 0.02).to(dev)
        rows["render_fwd_fused_soft"] = dict(us=event_time_us(lambda: seg_fwd(soft, False), iters, 5), bytes=B * BYTES_RENDER_FUSED,
                                             kernels="seg_sample_kernel+seg_combine_kernel (soft volume)",
                                             pmc=[k + "@soft" for k in seg_pmc], src=("common.hpp", "render_common.hpp", "sph_render_seg.hip"))
        rows["render_fwd_fused_soft_grad"] = dict(us=event_time_us(lambda: seg_fwd_grad(soft, False), iters, 5),
                                                  bytes=B * BYTES_RENDER_FUSED,
                                                  kernels="the same with a gradient wanted: + 4 B per sample of saved values")
        seg_fwd_grad(soft, False)
        rows["render_bwd_fused_soft"] = dict(us=event_time_us(lambda: std_bwd(soft, live), iters, 5),
                                             bytes=B * (BYTES_RENDER_FUSED + 128 ** 3 * 4),
                                             kernels="seg_combine_bwd_kernel + seg_scatter_kernel + seg_halo_kernel on the soft volume "
                                                     "(gradient everywhere)",
                                             pmc=[k + "@soft" for k in bwd_pmc], src=bwd_src)
        del soft
        seg_fwd(proj)
        if B >= 16:     # batch-minor tile renderer (csrc/sph_render_bm.hip): the volume with the image index fastest
            layer = G.Camera_back_projection_layer(batch_minor=True).to(dev)
            with torch.no_grad():
                proj_bm = layer(d)
            # the camera forward the batch-minor STEP runs (the layer's call: camera by value): image-minor volumes have no
            # contiguous z rows for cam_brick_kernel, so fill + the deterministic leader pass (round 5: no atomics, bit-identical
            # to a serial evaluation of the reference); `cam_bp_fwd_bm_atomics` = the path it replaced there and that tensor
            # cameras still take (fill, tile scatter with global float atomics, per-pixel normalise)
            cnt_bm = _fused_render.empty_batch_minor(proj_bm.shape, torch.float32, dev)
            t = event_time_us(lambda: cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, proj_bm, cnt_bm, shifted=True), iters, 5)
            rows["cam_bp_fwd_bm"] = dict(us=t, bytes=B * BYTES_CAM_FWD, kernels="fill2_vec4_kernel+cam_leader_kernel<2>",
                                         pmc=["fill2_vec4_kernel", "cam_leader_kernel<2>"], src=("common.hpp", "cam_bp.hip"))
            # ... and as the LAYER calls it in the step: cnt kept for the layer's own backward only, hence written only where a
            # point landed (half of the fill is not written), plus the occupancy words for the renderer
            tl = _fused_render.new_brick_words(B, 128, dev)
            t = event_time_us(lambda: cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, proj_bm, cnt_bm, shifted=True,
                                                                                tile_live=tl, sparse_cnt=True), iters, 5)
            cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, proj_bm, cnt_bm, shifted=True)     # (cnt dense again)
            rows["cam_bp_fwd_bm_layer"] = dict(us=t, bytes=B * (256 * 256 * 4 + 128 ** 3 * 4),
                                               kernels="fill1_vec4_kernel+cam_leader_kernel<2> (cnt only where a point landed; "
                                                       "occupancy words)",
                                               pmc=["fill1_vec4_kernel", "cam_leader_kernel<2>"], src=("common.hpp", "cam_bp.hip"))
            t = event_time_us(lambda: cam_bp_lib.back_projection_forward_shifted(d, cd, fl, proj_bm, cnt_bm), iters, 5)
            rows["cam_bp_fwd_bm_atomics"] = dict(us=t, bytes=B * BYTES_CAM_FWD,
                                                 kernels="fill2_vec4_kernel+scatter_tile_kernel<false>+normalise_tile_kernel<false>")
            cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, proj_bm, cnt_bm, shifted=True)
            TB = _fused_render.bm_tables_for(proj_bm.shape, dev, mod._dirs64, mod.depth_weight)
            groups = -(-B // 32)
            ps = torch.empty((groups * TB["segs"].shape[0] * 64,), device=dev)
            tr = torch.empty((ps.numel(),), device=dev)
            stash = torch.empty((groups * TB["rec_f"].shape[0] * 32,), device=dev)
            mask = torch.empty((groups * 128 ** 3 + groups,), dtype=torch.int32, device=dev)
            out_p = torch.empty((B, 1, 160, 160), device=dev)
            gout_p = torch.randn_like(out_p)
            gvox_bm = _fused_render.empty_batch_minor(proj_bm.shape, torch.float32, dev)

            # the volume as the step's layer hands it over: with the leader pass's occupancy words (which bricks of the group hold
            # anything but the fill value); the sampler copies constants for tiles it knows to be empty instead of reading them.
            # (The timing calls above re-wrote proj_bm through the raw C ABI -- same depth maps, same values -- and a raw write
            # drops a volume's hint (_loader._call, round 6): the words `tl` of the last hinted call are hung on it again.)
            _fused_render.attach_hint(proj_bm, tl, 128)
            blocked = _fused_render.provably_blocked(proj_bm, 50.0)     # the step's renderer saves nothing, its backward launches nothing
            words, ps_empty = _fused_render.occupancy_hint(proj_bm, TB, 50.0, render_lib, with_grad=True)
            # (None when GENRE_CAMBP_MODE pins another camera forward: every tile is read then)
            tiles_live = (words != 0).float().mean().item() if words is not None else 1.0   # share of the tiles that are read

            def bm_fwd(save, hint=True):
                render_lib.render_bm_forward(proj_bm, out_p, TB["segs"], TB["rec_f"], TB["fwd_rows"], TB["ray_ptr"],
                                             TB["ray_seg"], TB["ray_pre"], ps, stash if save else None,
                                             mask if save else None, 50.0, words if hint else None, ps_empty if hint else None)
            # `bytes` = the ALGORITHMIC bytes of the operator (volume in + map out, 8 454 144 B per image) -- the contract's
            # definition of roofline.achieved; `bytes_needed` = what the group has to move given the occupancy words (the live
            # tiles' voxels + the map): both rates are in the line, so that nobody reads the first as bytes the kernel touched
            rows["render_fwd_bm"] = dict(us=event_time_us(lambda: bm_fwd(True), iters, 5), bytes=B * BYTES_RENDER_FUSED,
                                         bytes_needed=int(B * (tiles_live * 128 ** 3 * 4 + 128 * 128 * 4)),
                                         tiles_live_frac=tiles_live,
                                         kernels="bm_sample_kernel+bm_combine_fwd_kernel on GenRe's volume, %.0f %% of the tiles live "
                                                 "(occupancy words of the camera forward)" % (100 * tiles_live),
                                         pmc=["bm_sample_kernel<true, true, true, 1024>@genre", "bm_combine_fwd_kernel@genre"],
                                         src=("common.hpp", "sph_render_bm.hip"))
            # ... and as the STEP runs it when the layer's value range proves on the host that the x50 clamp blocks every voxel
            # (toolbox/_fused_render.py: provably_blocked -- GenRe's own chain): nothing is saved for a backward that launches nothing
            rows["render_fwd_bm_nosave"] = dict(us=event_time_us(lambda: bm_fwd(False), iters, 5),
                                                bytes=rows["render_fwd_bm"]["bytes"], bytes_needed=rows["render_fwd_bm"]["bytes_needed"],
                                                tiles_live_frac=tiles_live,
                                                kernels="bm_sample_kernel+bm_combine_fwd_kernel on GenRe's volume without saved state "
                                                        "(inference; the step too when the clamp provably blocks every voxel), "
                                                        "%.0f %% of the tiles live" % (100 * tiles_live),
                                                pmc=["bm_sample_kernel<true, false, true, 1024>", "bm_combine_fwd_kernel@nosave"],
                                                src=("common.hpp", "sph_render_bm.hip"))
            rows["render_fwd_bm_dense"] = dict(us=event_time_us(lambda: bm_fwd(True, False), iters, 5), bytes=B * BYTES_RENDER_FUSED,
                                               kernels="the same volume WITHOUT the occupancy words: every tile is read")
            def bm_bwd_scatter():
                render_lib.render_bm_backward(gout_p, gvox_bm, TB["segs"], TB["ray_ptr"], TB["ray_seg"], TB["ray_pre"],
                                              TB["ent"], TB["rec_b"], TB["bwd_rows"], mod.depth_weight, ps, tr, stash,
                                              mask, 50.0, TB["pull_code"])

            # The same kernels on a volume whose every sample passes the clamps (`soft`: uniform(0.05, 0.95) / 50 under
            # pre_scale 50 -- the instantiation the step runs, with a gradient everywhere).  On GenRe's own volume the clamp
            # blocks every voxel (saturated or empty, depth_pred_with_sph_inpaint.py:124), the backward is identically zero
            # and the kernels only write zeros (render_bwd_bm below): the roofline of the kernel group is quoted on `soft`.
            gsoft = torch.Generator(device="cpu").manual_seed(1)
            soft_bm = _fused_render.empty_batch_minor(proj_bm.shape, torch.float32, dev)
            soft_bm.copy_(((torch.rand(proj_bm.shape, generator=gsoft) * 0.9 + 0.05) * 0.02).to(dev))

            def bm_fwd_soft():
                render_lib.render_bm_forward(soft_bm, out_p, TB["segs"], TB["rec_f"], TB["fwd_rows"], TB["ray_ptr"],
                                             TB["ray_seg"], TB["ray_pre"], ps, stash, mask, 50.0)
            bm_bwd_pmc = ["bm_combine_bwd_kernel", "bm_zero_shared_kernel<4, 8, 8>", "bm_scatter_kernel<true, 4, 8, 8, 768>"]
            rows["render_bwd_bm"] = dict(
                us=event_time_us(bm_bwd_scatter, iters, 5),
                bytes=B * (BYTES_RENDER_FUSED + 128 ** 3 * 4),
                kernels="bm_combine_bwd_kernel+bm_zero_shared_kernel+bm_scatter_kernel",
                pmc=[k + "@genre" for k in bm_bwd_pmc],
                src=("common.hpp", "sph_render_bm.hip"))
            bm_fwd(True)                                            # (the saved state of the GenRe volume again)
            rows["render_bwd_bm"]["us"] = event_time_us(bm_bwd_scatter, iters, 5)
            # on GenRe's own volume the clamp blocks every voxel: the group reads B group words and writes grad_vox = 0 --
            # it is billed with the bytes it MOVES there, not with the algorithmic bytes of a backward it does not compute
            rows["render_bwd_bm"]["bytes"] = B * 128 ** 3 * 4
            rows["render_bwd_bm"]["kernels"] += " on GenRe's volume (clamp blocks every voxel: writes zeros)"
            rows["render_fwd_bm_soft"] = dict(us=event_time_us(bm_fwd_soft, iters, 5), bytes=B * BYTES_RENDER_FUSED,
                                              kernels="bm_sample_kernel+bm_combine_fwd_kernel (soft volume)",
                                              pmc=["bm_sample_kernel<true, true, false, 1024>@soft", "bm_combine_fwd_kernel@soft"],
                                              src=("common.hpp", "sph_render_bm.hip"))
            bm_fwd_soft()
            rows["render_bwd_bm_soft"] = dict(
                us=event_time_us(bm_bwd_scatter, iters, 5),
                bytes=B * (BYTES_RENDER_FUSED + 128 ** 3 * 4),
                kernels="bm_combine_bwd_kernel+bm_zero_shared_kernel+bm_scatter_kernel on the soft volume (gradient everywhere)",
                pmc=[k + "@soft" for k in bm_bwd_pmc], src=("common.hpp", "sph_render_bm.hip"))
            bm_fwd(True)
    if fused_ok and B >= 16:
        # The same groups IN STEP ORDER (camera forward -> renderer forward -> renderer backward -> camera backward, eager
        # launches, HIP events around ONE group per pass): what a group takes when the launches in front of it have just streamed
        # 0.3-0.5 GB through the caches -- the rocprofv3 trace of profiles/pmc_targets.py sees the same (cold tables and records),
        # the back-to-back figure above is the warm one.  Both are in the line (`us`, `us_in_step_order`).
        gd = torch.empty_like(d)
        gfl, gcd = torch.empty((B, 1), device=dev), torch.empty((B, 1), device=dev)
        zero_word = torch.zeros((1,), dtype=torch.int32, device=dev)
        seq = [("cam_bp_fwd_bm_layer", lambda: cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, proj_bm, cnt_bm, shifted=True,
                                                                                        tile_live=tl, sparse_cnt=True))]
        if blocked:
            # the step as its autograd chain runs it on GenRe's volume: the layer's value range proves on the host that the x50 clamp
            # blocks every voxel -- the renderer saves nothing, its backward launches nothing (the gradient is a stride-0 view of one
            # zero), the layer's backward gets "every image's gradient is zero" in one word
            seq += [("render_fwd_bm_nosave", lambda: bm_fwd(False)),
                    ("cam_bp_bwd_bm", lambda: cam_bp_lib.back_projection_backward_hinted(d, fl, cd, cnt_bm, gvox_bm, gd, gcd, gfl, zero_word,
                                                                                         0, 0, 1 << 30))]
        else:
            # (GENRE_LAZY_ZERO_GRAD=0: the kernels find the zeros themselves -- the renderer's backward writes them and hangs "this
            # group's gradient is identically zero", the trailing words of its clamp mask, on the gradient it returns)
            seq += [("render_fwd_bm", lambda: bm_fwd(True)),
                    ("render_bwd_bm", bm_bwd_scatter),
                    ("cam_bp_bwd_bm", lambda: cam_bp_lib.back_projection_backward_hinted(d, fl, cd, cnt_bm, gvox_bm, gd, gcd, gfl, mask, 1,
                                                                                         groups * 128 ** 3, 32))]
        seq += [("cam_bp_bwd_bm_nohint", lambda: cam_bp_lib.back_projection_backward_shifted(d, fl, cd, cnt_bm, gvox_bm, gd, gcd, gfl))]
        rows["_step_groups"] = [name for name, _ in seq[:-1]]
        for _ in range(3):
            for _, fn in seq:
                fn()
        torch.cuda.synchronize()
        marks = {name: [] for name, _ in seq}
        for _ in range(20):
            for name, fn in seq:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                marks[name].append((a, b))
        torch.cuda.synchronize()
        for name, evs in marks.items():
            us = sum(a.elapsed_time(b) for a, b in evs) * 1e3 / len(evs)
            if name in rows:
                rows[name]["us_in_step_order"] = us
            else:
                rows[name] = dict(us=us, us_in_step_order=us, bytes=B * 670000,       # depth + grad_depth + 8 B per in-grid point
                                  kernels=("cam_backward_kernel (+ its two scalar zero fills), timed in step order only; " +
                                           ("with the renderer's zero-gradient words, as the step runs it: every group of GenRe's "
                                            "chain is blocked by the clamp, grad_depth = 0 is written without reading anything"
                                            if name == "cam_bp_bwd_bm" else "without the words: every pixel's gather and arithmetic")))
        cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, proj_bm, cnt_bm, shifted=True)     # (cnt dense again)
    for r in rows.values():
        if not isinstance(r, dict):         # ("_step_groups": the names of the step's groups)
            continue
        r["GBs"] = r["bytes"] / r["us"] / 1e3
        if "bytes_needed" in r:
            r["GBs_needed"] = r["bytes_needed"] / r["us"] / 1e3
    # forward-only chain (inference) at this batch size, standard layout and batch-minor layout
    if fused_ok:
        with torch.no_grad():
            for name, bm in (("chain_fwd", False), ("chain_fwd_batch_minor", True)):
                net = HotPath(G, True, batch_minor=bm).to(dev)
                t = event_time_us(lambda: net(d), iters, 3)
                rows[name] = dict(us=t, bytes=B * (BYTES_CAM_FWD + BYTES_RENDER_FUSED), GBs=B * (BYTES_CAM_FWD + BYTES_RENDER_FUSED) / t / 1e3,
                                  kernels="cam_bp forward (shifted) + fused render forward (+pad)")
    # Chamfer forward, both directions, B x 2048 x 2048 (configs[0]'s cloud size): fp32-VALU bound, 8 flops per
    # pair (3 sub, 3 mul, 2 add -- no fma: the distance must round like the reference's expression)
    from genre_shapehd_amd.toolbox.nndistance._ext import my_lib
    n = 2048
    a = torch.rand((B, n, 3), device=dev)
    b = torch.rand((B, n, 3), device=dev)
    d1 = torch.empty((B, n), device=dev); d2 = torch.empty((B, n), device=dev)
    i1 = torch.empty((B, n), device=dev, dtype=torch.int32); i2 = torch.empty_like(i1)
    t = event_time_us(lambda: my_lib.nnd_forward_cuda(a, b, d1, d2, i1, i2), iters, 5)
    flops = 2 * B * n * n * 8
    rows["nnd_fwd"] = dict(us=t, bytes=B * n * (2 * 12 + 2 * 8), kernels="nnd_forward_kernel", TFLOPs=flops / t / 1e6,
                           frac_fp32_valu=flops / t / 1e6 / 157.3, pairs=2 * B * n * n)
    rows["nnd_fwd"]["GBs"] = rows["nnd_fwd"]["bytes"] / t / 1e3
    # configs[0] on the GPU: one 2048 x 2048 pair, replayed from a HIP graph (launch cost excluded)
    a1, b1 = a[:1].contiguous(), b[:1].contiguous()
    o = [torch.empty((1, n), device=dev) for _ in range(2)] + [torch.empty((1, n), device=dev, dtype=torch.int32) for _ in range(2)]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        my_lib.nnd_forward_cuda(a1, b1, o[0], o[1], o[2], o[3])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(20):
            my_lib.nnd_forward_cuda(a1, b1, o[0], o[1], o[2], o[3])
    rows["nnd_fwd"]["cfg0_us"] = event_time_us(graph.replay, 20, 3) / 20
    return rows


def batch1_graph(G, dev):
    """cam_bp fwd + calc_prob fwd at batch 1, replayed from a HIP graph (no host launch cost)"""
    from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib
    from genre_shapehd_amd.toolbox.calc_prob.calc_prob._ext import calc_prob_lib
    import inputs
    d = torch.from_numpy(inputs.sphere_depth(noise_seed=2)).to(dev)
    fl = torch.full((1, 1), 418.3, device=dev)
    cd = torch.full((1, 1), 2.2, device=dev)
    tdf = torch.empty((1, 1, 128, 128, 128), device=dev)
    cnt = torch.empty_like(tdf)
    p = torch.rand((1, 1, 128, 128, 256), device=dev).clamp_(1e-5, 1 - 1e-5)
    s = torch.empty_like(p)
    reps = 20

    # the camera as Camera_back_projection_layer passes it when called with Python floats (the reference's default call,
    # camera_backprojection_module.py:12-21): by value (genre_back_projection_forward_const); `tensor_camera` below
    # repeats the measurement with fl / cam_dist tensors (genre_back_projection_forward)
    def cam_fwd():
        cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, tdf, cnt)

    def body(cam=cam_fwd):
        for _ in range(reps):
            cam()
            calc_prob_lib.calc_prob_forward(p, s)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        body()
    us = event_time_us(graph.replay, 20, 3) / reps
    nbytes = BYTES_CAM_FWD + BYTES_CP_FWD
    res = dict(us_per_image=us, GBs=nbytes / us / 1e3, frac=nbytes / us / 1e3 / HBM_PEAK_GBS,
               launches_per_image=2, note="HIP-graph replay of 20x(cam_bp fwd [cam_brick_kernel, camera by value] + "
                                          "calc_prob fwd), batch 1, one stream")
    try:
        gt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gt):
            body(lambda: cam_bp_lib.back_projection_forward(d, cd, fl, tdf, cnt))
        ust = event_time_us(gt.replay, 20, 3) / reps
        res["tensor_camera"] = dict(us_per_image=ust, frac=nbytes / ust / 1e3 / HBM_PEAK_GBS,
                                    note="same, fl / cam_dist read from [1,1] tensors")
    except Exception as e:      # pragma: no cover
        res["tensor_camera"] = dict(error=str(e)[:200])
    try:
        # The same 20 + 20 calls as TWO request streams inside one graph: cam_bp's latency-bound kernel of image
        # i+1 run beside calc_prob's bandwidth-bound kernel of image i (a server pipelining consecutive batch-1
        # requests over two HIP streams).  Reported beside the serial figure, never instead of it.
        s2 = torch.cuda.Stream()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            main = torch.cuda.current_stream()
            s2.wait_stream(main)
            with torch.cuda.stream(s2):
                for _ in range(reps):
                    calc_prob_lib.calc_prob_forward(p, s)
            for _ in range(reps):
                cam_fwd()
            main.wait_stream(s2)
        us2 = event_time_us(g2.replay, 20, 3) / reps
        res["two_streams"] = dict(us_per_image=us2, GBs=nbytes / us2 / 1e3, frac=nbytes / us2 / 1e3 / HBM_PEAK_GBS,
                                  note="cam_bp and calc_prob of consecutive requests on two HIP streams")
    except Exception as e:      # pragma: no cover
        res["two_streams"] = dict(error=str(e)[:200])
    try:        # forward of the whole configs[1] chain at batch 1, also from a graph (never fatal for the bench line)
        chain = HotPath(G, True).to(dev)
        with torch.no_grad():
            for _ in range(3):
                chain(d)
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                chain(d)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2):
                for _ in range(reps):
                    out = chain(d)
            res["chain_fwd_us_per_image"] = event_time_us(g2.replay, 20, 3) / reps
            # the same forward chain at batch 8 (SURVEY 8d M1 quotes batch 1 and batch 8), eager launches
            d8 = torch.from_numpy(inputs.batch_depth(8)).to(dev)
            for _ in range(3):
                chain(d8)
            res["chain_fwd_b8_us_per_image"] = event_time_us(lambda: chain(d8), 20, 3) / 8
    except Exception as e:      # pragma: no cover
        res["chain_fwd_us_per_image"] = None
        res["chain_fwd_error"] = str(e)[:200]
    try:        # all of GenRe's geometry between its networks, forward, batch 1, from a HIP graph:
        # get_abs_depth -> cam_bp -> x50/clamp -> render_spherical -> sph_pad  [net2]  crop/1-x -> spherical
        # back-projection -> refiner input [1,2,128^3]   (depth_pred_with_sph_inpaint.py:120-142,
        # genre_full_model.py:122-143); the spherical map net2 would return is stood in by its input
        from genre_shapehd_amd.callers import GenReGeometry
        geo = GenReGeometry().to(dev)
        rng = np.random.default_rng(3)
        pred = torch.from_numpy(rng.uniform(20, 80, (1, 1, 256, 256)).astype(np.float32)).to(dev)
        sil = torch.zeros((1, 1, 256, 256), device=dev)
        sil[:, :, 64:192, 64:192] = 100.0
        mm = torch.tensor([[1.8, 2.6]], device=dev)

        def geometry():
            dd = geo.get_abs_depth(pred, mm, sil)
            proj50, sph_in = geo.depth_to_spherical(dd)
            return geo.refiner_input(sph_in, proj50)

        with torch.no_grad():
            for _ in range(3):
                geometry()
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                geometry()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g3 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g3):
                for _ in range(reps):
                    keep = geometry()
            res["genre_geometry_fwd_us_per_image"] = event_time_us(g3.replay, 20, 3) / reps
    except Exception as e:      # pragma: no cover
        res["genre_geometry_fwd_us_per_image"] = None
        res["genre_geometry_error"] = str(e)[:200]
    return res


def hot_path_batch1(G, dev, reps=20):
    """configs[1] AS BASELINE.json STATES IT: one 256x256 depth map, forward AND backward (the reference tests at batch 1,
    scripts/test_genre.sh:30; chain models/depth_pred_with_sph_inpaint.py:120-126), the reference's NCXYZ layout, replayed from
    a HIP graph of `reps` forward+backward passes.  Two volumes: GenRe's own (clamp(proj * 50): every voxel saturated or
    empty, the gradient through render_spherical identically zero, as in the reference) and the same chain with the
    pre-scale at 0.9 instead of 50 (occupied voxels 0.12 ... 0.9: the clamp passes them -- a live gradient through the
    renderer's backward down to the depth map)."""
    import inputs
    res = {"what": "configs[1] at its stated size: ONE 256x256 depth map -> cam_bp -> 128^3 -> clamp(x s) -> render_spherical -> "
                   "sph_pad -> 160x160, forward + backward to grad_depth, NCXYZ layout, HIP-graph replay of %d passes" % reps}
    d = torch.from_numpy(inputs.sphere_depth(noise_seed=2)).to(dev).requires_grad_(True)
    gout = torch.randn((1, 1, 160, 160), device=dev)
    net = HotPath(G, True).to(dev)
    nbytes_f = BYTES_CAM_FWD + BYTES_RENDER_FUSED
    nbytes_b = BYTES_RENDER_FUSED + 128 ** 3 * 4 + BYTES_CAM_FWD
    for name, scale in (("genre_volume", 50.0), ("live_gradient", 0.9)):
        def one():
            d.grad = None
            out = net.render(net.cam(d), pre_scale=scale, pad=16)
            out.backward(gout)
        try:
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                one()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            d.grad = None
            with torch.cuda.graph(g):
                for _ in range(reps):
                    one()
            us = event_time_us(g.replay, 20, 3) / reps
            res[name] = {"us_per_image_fwd_bwd": us, "shapes_per_s": 1e6 / us, "pre_scale": scale,
                         "grad_depth_absmax": float(d.grad.abs().max()),
                         "GBs_algorithmic": (nbytes_f + nbytes_b) / us / 1e3,
                         "frac_of_8TBs": (nbytes_f + nbytes_b) / us / 1e3 / HBM_PEAK_GBS}
            del g
        except Exception as e:      # pragma: no cover -- never fatal for the bench line
            res[name] = {"error": repr(e)[:300]}
    return res


def m1_capture(G, dev, batches=(1, 8)):
    """BASELINE.json metric, first half: GenRe forward passes per second, 256x256 RGB -> 128^3 voxels, on one GPU.
    The reference's full model (models/genre_full_model.py:116-132: MarrNet-1, the geometric ops, the inpainting
    U-ResNet, the spherical back-projection, Unet_3D) with seeded random weights (no checkpoint ships),
    eval mode, no_grad; each batch size captured once in a HIP graph.  -> {batch: (replay, eager)}"""
    from genre_shapehd_amd.models import GenReNet, GenReInference
    torch.manual_seed(0)
    net = GenReNet().to(dev).eval()
    out = {}
    for n in batches:
        inf = GenReInference(net, device=dev, graph=True)
        rgb = torch.rand(n, 3, 256, 256, device=dev)
        sil = torch.zeros(n, 1, 256, 256, device=dev)
        sil[:, :, 48:208, 48:208] = 100.0
        inf.predict(rgb, sil)                                             # capture
        g = inf._captured[tuple(rgb.shape)][0]
        eager = GenReInference(net, device=dev, graph=False)
        out[n] = (g.replay, lambda eager=eager, rgb=rgb, sil=sil: eager.predict(rgb, sil), inf)
    return out


def m1_table(cap):
    res = {"what": "GenRe full-model forward (3 networks + geometric ops), random weights, fp32, HIP-graph replay",
           "target_fwd_per_s_batch1": 50.0}
    # FLOPs of one forward, counted with torch.utils.flop_counter over the same call (convolutions + linears of the three
    # networks; profiles/r05a_m1_flops.json: 102.28 GFLOP per image, 78 of them in Unet_3D, 53.7 in its 8^3-kernel
    # ConvTranspose3d) against the dense fp32 MFMA peak (MI355X_MICROARCH.md: 157.3 TFLOP/s)
    res["fp32_mfma_peak_TFLOPs"] = FP32_MFMA_PEAK_TFLOPS
    for n, (replay, eager, inf) in cap.items():
        us = event_time_us(replay, 10, 2)
        us_eager = event_time_us(eager, 10, 2)
        row = {"ms_per_forward": us / 1e3, "shapes_per_s": n / us * 1e6, "eager_ms": us_eager / 1e3}
        try:
            from torch.utils.flop_counter import FlopCounterMode
            with FlopCounterMode(display=False) as fc:
                eager()
            row["flops_per_forward"] = int(fc.get_total_flops())
            row["TFLOPs"] = row["flops_per_forward"] / us / 1e6
            row["frac_fp32_mfma"] = row["TFLOPs"] / FP32_MFMA_PEAK_TFLOPS
        except Exception as e:      # pragma: no cover
            row["flops_error"] = repr(e)[:200]
        res["batch%d" % n] = row
    return res


def train_bench(dev, dist, du, world, rank, steps, which=("shapehd", "genre")):
    """BASELINE.json configs[3] / configs[4] as per-rank shards of their 8-GPU batches: one optimizer step of
      shapehd_b8      ShapeHD fine-tuning, batch 64 / 8 = 8 per rank (models/shapehd.py:82-118, marrnet2.py:46-54)
      wgangp_b8       3-D WGAN-GP critic + generator step, batch 8 per rank (models/wgangp.py:77-164)
      genre_joint_b4  GenRe joint fine-tuning through the projections + Chamfer, batch 32 / 8 = 4 per rank
                      (depth_pred_with_sph_inpaint.py:114-118, genre_full_model.py:117-121)
    at the reference's network widths, fp32, Adam, synthetic seeded batches resident on the device, every trainable
    network under DistributedDataParallel when world > 1 (RCCL all-reduce overlapped with the backward).  Runs on EVERY
    rank; timing = barrier + synchronize on both sides, max over ranks; samples/s = world * batch * steps / time."""
    from genre_shapehd_amd import train as T
    from genre_shapehd_amd.models import shapehd as MS
    from genre_shapehd_amd.models.genre import GenReNet, GenReOptions
    res = {"what": "one optimizer step per config at the per-rank shard of the 8-GPU batch, reference widths, fp32, "
                   "DDP over %d rank(s)" % world, "steps": steps}

    def fence():
        du.fence(dist, torch.cuda.synchronize)

    def agree(ok):
        return du.all_agree(dist, ok, dev)

    def timed(name, batch, fn, note):
        """Failures are made COLLECTIVE: a rank whose warm-up or timed loop raised says so in a MIN all-reduce that
        every rank enters, and all of them drop the config together -- nobody is left waiting in a barrier or in DDP's
        all-reduce for a rank that has moved on.  (A rank that dies INSIDE a step is bounded by the process group's time-out,
        dist_utils.init_from_env.)"""
        err = None
        try:
            for _ in range(2):
                fn()                                                    # MIOpen find, allocator, DDP bucket build
        except Exception as e:      # pragma: no cover
            err = repr(e)[:300]
        if not agree(err is None):
            res[name] = {"error": err or "another rank failed in the warm-up steps"}
            torch.cuda.empty_cache()
            return
        fence()
        t0 = time.perf_counter()
        try:
            for _ in range(steps):
                fn()
        except Exception as e:      # pragma: no cover
            err = repr(e)[:300]
        if not agree(err is None):
            res[name] = {"error": err or "another rank failed in the timed steps"}
            torch.cuda.empty_cache()
            return
        fence()
        mine = time.perf_counter() - t0
        el = du.max_over_ranks(dist, mine, dev)
        res[name] = {"batch_per_gpu": batch, "ms_per_step": el * 1e3 / steps,
                     "samples_per_s": world * batch * steps / el,
                     "per_rank_samples_per_s": [batch * steps / t for t in du.per_rank(dist, mine, dev)], "what": note}
        torch.cuda.empty_cache()

    to = lambda ns: type(ns)(**{k: v.to(dev) for k, v in vars(ns).items()})       # noqa: E731
    torch.manual_seed(1234)                                             # identical initial weights on every rank

    # Every config builds its networks inside its own try block and the ranks agree on the outcome before anyone runs a step:
    # a failure (out of memory, a DDP construction error) is reported in the JSON line under that config's key and never
    # takes the bench line -- or another rank -- with it.
    def config(name, build):
        """`build` is a generator: everything local to the rank (networks, batch, optimizer) before its `yield`, the DDP
        wrappers -- whose construction is itself collective -- behind it; the ranks agree in between"""
        holder = {}
        gen = build()
        try:
            if os.environ.get("GENRE_BENCH_INJECT_FAILURE") == "%d:%s" % (rank, name):      # tests/test_bench_launcher.py
                raise RuntimeError("injected failure on rank %d" % rank)
            next(gen)
        except Exception as e:      # pragma: no cover
            holder["err"] = repr(e)[:300]
        if not agree("err" not in holder):
            res[name] = {"error": holder.get("err", "another rank failed to build this config")}
            torch.cuda.empty_cache()
            return
        try:
            next(gen)
            holder["err"] = "builder did not finish"
        except StopIteration as fin:
            holder["step"] = fin.value
        except Exception as e:      # pragma: no cover
            holder["err"] = repr(e)[:300]
        if agree("step" in holder):
            timed(name, *holder["step"])
        else:
            res[name] = {"error": holder.get("err", "another rank failed to wrap this config")}
        torch.cuda.empty_cache()

    def shapehd():      # configs[3]
        ins, vox = T.sketch_batch(8, "cpu", seed=500 + rank)
        ins, vox = to(ins), vox.to(dev)
        net = MS.ShapeHDNet().to(dev).train()
        optim = torch.optim.Adam(net.marrnet2.parameters(), lr=1e-4, betas=(0.5, 0.9))
        yield
        model = T.ddp(net, dev, dist)
        return (8, lambda: T.shapehd_train_step(model, optim, ins, vox, 1e-3),
                "MarrNet-2 fine-tuned against the frozen 3-D critic: forward (incl. frozen copy + critic), backward, Adam")

    def wgangp():       # configs[3]'s critic
        _, vox = T.sketch_batch(8, "cpu", seed=500 + rank)
        vox = vox.to(dev)
        gan = MS.WGANGP(lr=1e-4)
        gan.net_g.to(dev), gan.net_d.to(dev)
        yield
        gan.net_g, gan.net_d = T.ddp(gan.net_g, dev, dist), T.ddp(gan.net_d, dev, dist)
        return (8, lambda: gan.train_on_batch(0, vox),
                "critic step (real, fake, second-order gradient penalty) + generator step")

    def genre():        # configs[4]
        gopt = GenReOptions(joint_train=True)
        net = GenReNet(gopt).to(dev).train()
        with torch.no_grad():                                           # a depth range that puts the surface in the cube
            head = net.depth_and_inpaint.net1.decoder_minmax[9]
            head.weight.zero_()
            head.bias.copy_(torch.tensor([1.9, 2.4]))
        optim = torch.optim.Adam(net.parameters(), lr=1e-6, betas=(0.5, 0.9))
        gin, gt = T.genre_batch(4, "cpu", seed=600 + rank)
        gin, gt = to(gin), to(gt)
        yield
        model = T.ddp(net, dev, dist)
        return (4, lambda: T.genre_train_step(model, optim, gin, gt, gopt, chamfer_weight=0.1),
                "all three modules + cam_bp / render_spherical / spherical back-projection / Chamfer in the graph, Adam")

    for name, key, build in (("shapehd", "shapehd_b8", shapehd), ("wgangp", "wgangp_b8", wgangp), ("genre", "genre_joint_b4", genre)):
        if name in which:
            config(key, build)
    return res


def cpu_baseline(budget_s):
    """the same step on the host cores: reference kernel bodies (oracle/_ref) if they travelled
    with the snapshot, else the C port; torch CPU ops (1 thread) for grid_sample/matmul."""
    import inputs
    from oracle.oracle import Oracle, Reference, reference_available
    from oracle.torch_oracle import HotPathCPU
    torch.set_num_threads(1)
    backend = Reference() if reference_available() else Oracle()
    hp = HotPathCPU(backend)
    g = torch.from_numpy(np.random.default_rng(0).standard_normal((1, 1, 160, 160)).astype(np.float32))
    depths = inputs.batch_depth(4)
    hp.forward_backward(torch.from_numpy(depths[:1]), g)               # warm-up (grid tables, page-in)
    n, t0 = 0, time.perf_counter()
    while True:
        hp.forward_backward(torch.from_numpy(depths[n % 4:n % 4 + 1]), g)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    res = dict(value=n / el, unit="shapes/s", cores=1, kind=backend.kind,
               what="the configs[1] forward+backward chain on the host: pairs with `hot_path` / `hot_path_batch1` (shapes/s of the "
                    "same chain on the GPU), NOT with the top-level `value` (the GenRe full-model forward, whose networks have "
                    "no host leg here)",
               sample="%d depth maps fwd+bwd through the same chain in %.1f s, 1 thread" % (n, el))
    # the same chain on several host cores at once (one image per thread, intra-op threads stay at 1), so that the
    # host figure is not artificially weak; bounded to 16 threads / ~8 s
    try:
        import concurrent.futures as cf
        cores = max(1, min(os.cpu_count() or 1, 16))
        if cores > 1:
            hps = [HotPathCPU(backend) for _ in range(cores)]
            deadline = time.perf_counter() + min(8.0, budget_s)

            def work(i):
                k = 0
                while True:
                    hps[i].forward_backward(torch.from_numpy(depths[(i + k) % 4:(i + k) % 4 + 1]), g)
                    k += 1
                    if time.perf_counter() >= deadline:
                        return k
            t1 = time.perf_counter()
            with cf.ThreadPoolExecutor(cores) as ex:
                total = sum(ex.map(work, range(cores)))
            res["all_cores"] = dict(value=total / (time.perf_counter() - t1), unit="shapes/s", cores=cores,
                                    sample="%d depth maps over %d threads" % (total, cores))
    except Exception as e:      # pragma: no cover -- never fatal for the bench line
        res["all_cores"] = dict(value=None, error=str(e)[:200])
    # configs[0]: Chamfer on two 2048-point clouds, the reference's CPU path (my_lib.c nnsearch x2), 1 thread
    try:
        x1, x2 = inputs.clouds(1, 2048, 2048, seed1=0, seed2=1)
        backend.nnd_forward(x1, x2)
        t1 = time.perf_counter()
        for _ in range(5):
            backend.nnd_forward(x1, x2)
        res["nnd_cfg0_ms"] = (time.perf_counter() - t1) / 5 * 1e3
        # the same pair through the PRODUCT's host entry points (csrc/nnd_host.hip = the reference's my_lib.nnd_forward,
        # the path its NNDFunction takes for CPU tensors), multi-threaded
        import genre_shapehd_amd as G
        a, b = torch.from_numpy(x1), torch.from_numpy(x2)
        G.nndistance_w_idx(a, b)
        t1 = time.perf_counter()
        for _ in range(20):
            G.nndistance_w_idx(a, b)
        res["nnd_cfg0_product_host_ms"] = (time.perf_counter() - t1) / 20 * 1e3
    except Exception as e:      # pragma: no cover
        res.setdefault("nnd_cfg0_ms", None)
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return respawn_under_torchrun(args)            # `python bench.py --gpus N`: one rank per GPU over RCCL
    if args.stub:
        return stub_main(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import genre_shapehd_amd as G
    from genre_shapehd_amd import dist_utils
    dist = dist_utils.init_from_env("nccl", dev)           # RCCL; None when WORLD_SIZE == 1
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    import inputs
    from genre_shapehd_amd.toolbox import _fused_render
    fused = (not args.unfused) and _fused_render.available()
    B = args.batch
    bm = fused and args.layout == "bm" and B >= 16
    model = HotPath(G, fused, batch_minor=bm).to(dev)
    depth = torch.from_numpy(inputs.batch_depth(B, seed=100 + 1000 * rank)).to(dev).requires_grad_(True)
    grad_out = torch.randn((B, 1, 160, 160), device=dev)

    def step():
        depth.grad = None
        out = model(depth)
        out.backward(grad_out)

    def fence():
        dist_utils.fence(dist, torch.cuda.synchronize)

    step()                                  # set-up pass: builds the geometry tables on the host; never timed
    # The step is ~13 kernels of 5 ... 600 us: launched eagerly, the host-side gaps between them (autograd bookkeeping,
    # allocator, ctypes) are ~10 % of it.  Capture forward + backward once in a HIP graph and replay it -- the same
    # kernels on the same data, without the gaps (--eager times the plain launches instead).
    run, launch = step, "eager launches"
    graph_holder = []
    if not args.eager:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            depth.grad = None
            with torch.cuda.graph(graph):
                out_g = model(depth)
                out_g.backward(grad_out)
            graph_holder.append(graph)
            run, launch = graph.replay, "HIP-graph replay of forward + backward"
        except Exception as e:      # pragma: no cover -- fall back to eager launches
            launch = "eager launches (graph capture failed: %s)" % str(e)[:120]
            torch.cuda.synchronize()
    def timed_steps(fn):
        """W untimed + exactly K timed calls, barrier + synchronize on both sides, the maximum over ranks"""
        for _ in range(args.warmup):
            fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        fence()
        timed_steps.last_local = time.perf_counter() - t0             # this rank's own clock (per-rank figures of the line)
        return dist_utils.max_over_ranks(dist, timed_steps.last_local, dev)

    hot_elapsed = timed_steps(run)
    hot_per_rank = dist_utils.per_rank(dist, timed_steps.last_local, dev)
    hot = {"what": "configs[1]: 256x256 depth -> cam_bp -> 128^3 voxel -> clamp(x50) -> render_spherical (calc_prob) -> 160x160 "
                   "spherical map, forward + backward, inputs resident in HBM",
           "shapes_per_s": world * B * args.steps / hot_elapsed, "ms_per_step": hot_elapsed * 1e3 / args.steps,
           "batch_per_gpu": B, "steps": args.steps, "step_launch": launch,
           "per_rank_shapes_per_s": [B * args.steps / t for t in hot_per_rank],
           "render_spherical": "fused" if fused else "reference op sequence (grid_sample + CalcStopProb)",
           "volume_layout": "batch-minor (image index fastest)" if bm else "NCXYZ",
           "note": "on GenRe's own volume the x50 clamp blocks every voxel: the gradient through render_spherical is identically "
                   "zero, as in the reference.  The camera layer hangs its value range ({0} u [0.13, 1]) on the volume, the fused "
                   "renderer sees on the HOST that clamp(x * 50) blocks both ends, saves nothing for a backward and launches "
                   "nothing in it: the gradient is a stride-0 view of one zero carrying 'every image's gradient is zero' for the "
                   "camera layer's backward (toolbox/_fused_render.py: provably_blocked; GENRE_LAZY_ZERO_GRAD=0 switches it off -- "
                   "the kernels then find and write the same zeros: kernels.render_bwd_bm / render_bwd_fused).  "
                   "kernels.render_bwd_bm_soft and `roofline_soft` describe the backward kernels where they do work",
           "renderer_backward": "none launched (volume provably blocked)" if os.environ.get("GENRE_LAZY_ZERO_GRAD", "1") != "0"
                                else "kernels write the zeros"}
    # ---- the headline: BASELINE.json's metric -- GenRe full-model forward passes per second, batch 1 per GPU -------
    del graph_holder[:]
    torch.cuda.empty_cache()
    cap, value, ms_per_step, m1_err, m1_per_rank = None, None, None, None, None
    if not args.no_m1:
        try:
            cap = m1_capture(G, dev)
        except Exception as e:      # pragma: no cover
            m1_err = repr(e)[:300]
        ok = dist_utils.all_agree(dist, cap is not None, dev)
        if ok:
            el = timed_steps(cap[1][0])
            value, ms_per_step = world * 1 * args.steps / el, el * 1e3 / args.steps
            m1_per_rank = [args.steps / t for t in dist_utils.per_rank(dist, timed_steps.last_local, dev)]
        elif m1_err is None:
            m1_err = "another rank failed to build the GenRe forward"
    if value is None:                      # --no-m1 (or the networks could not be built): the hot-path step is what was timed
        value, ms_per_step = hot["shapes_per_s"], hot["ms_per_step"]
    train = None
    if not args.no_train:           # every rank takes part (DDP all-reduce); reported in the same JSON line
        torch.cuda.empty_cache()
        which = ("shapehd", "wgangp", "genre") if args.train_configs == "all" else tuple(args.train_configs.split(","))
        train = train_bench(dev, dist, dist_utils, world, rank, args.train_steps, which)

    if rank == 0:
        rows = kernel_table(G, dev, B)
        # kernel groups that are part of the hot-path step in this mode
        if not fused:
            in_step = ["cam_bp_fwd", "calc_prob_fwd", "calc_prob_bwd_fused"]
        elif bm:
            in_step = rows.pop("_step_groups", ["cam_bp_fwd_bm_layer", "render_fwd_bm", "render_bwd_bm", "cam_bp_bwd_bm"])
        else:
            # (render_bwd_fused: only when GENRE_LAZY_ZERO_GRAD=0 -- otherwise the clamp provably blocks every voxel of the step's
            # volume and the renderer's backward launches nothing: toolbox/_fused_render.py: provably_blocked)
            in_step = ["cam_bp_fwd", "render_fwd_fused"] + (["render_bwd_fused"] if os.environ.get("GENRE_LAZY_ZERO_GRAD", "1") == "0" else [])
        rows.pop("_step_groups", None)
        for k in ("m2_pair_alternating",):
            rows[k]["GBs"] = rows[k]["bytes"] / rows[k]["us"] / 1e3
        # `roofline` = the slowest hand-written kernel group OF THE TIMED hot-path step, on the volume the step renders (round 5;
        # the block must describe a timed region).  The renderer's backward where it does work -- the soft volume,
        # which no timed step renders -- keeps its own block, `roofline_soft`.
        dom_name = max(in_step, key=lambda k: rows[k].get("us_in_step_order", rows[k]["us"]))
        dom = rows[dom_name]
        traffic, traffic_src = pmc_traffic(dom.get("pmc_in_step", dom.get("pmc", [])), B, dom.get("src", ("common.hpp",)))

        def roof(name, row, tr, tr_src):
            r = {"bound": "hbm", "kernel": name + " (" + row["kernels"] + ")", "achieved": row["GBs"], "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": row["GBs"] / HBM_PEAK_GBS, "traffic": tr, "traffic_unit": "bytes/launch",
                 "traffic_source": tr_src, "algorithmic_bytes_per_launch": row["bytes"], "avg_launch_us": row["us"]}
            if "us_in_step_order" in row:
                # the headline figures of the block are the group BETWEEN the step's other launches (cold tables: what the step pays
                # and what the rocprofv3 trace of profiles/pmc_targets.py shows); back-to-back launches of the group alone keep
                # their tables warm and are reported beside it
                r["avg_launch_us_back_to_back"], r["frac_back_to_back"] = row["us"], row["GBs"] / HBM_PEAK_GBS
                r["avg_launch_us"] = row["us_in_step_order"]
                r["achieved"] = row["bytes"] / row["us_in_step_order"] / 1e3
                r["frac"] = r["achieved"] / HBM_PEAK_GBS
                r["timing"] = "HIP events around the group inside eager passes of the step's launch order (20 passes)"
            if "bytes_needed" in row:       # the occupancy words let the group skip tiles: what it must move is less than the operator's bytes
                r["bytes_needed_per_launch"] = row["bytes_needed"]
                r["achieved_on_bytes_needed"] = row["bytes_needed"] / r["avg_launch_us"] / 1e3
                r["frac_on_bytes_needed"] = r["achieved_on_bytes_needed"] / HBM_PEAK_GBS
                r["tiles_live_frac"] = row.get("tiles_live_frac")
                if r["frac_on_bytes_needed"] < 0.15:
                    # on the bytes it really has to move the group is nowhere near the HBM roof: it is bound by dependent memory
                    # round trips and by how many workgroups a CU holds (per-workgroup timelines: profiles/r06_ab_experiments.txt)
                    r["bound"] = "latency"
                    r["bound_note"] = ("frac / peak are kept against the 8 TB/s HBM roof as the contract defines them; the group is "
                                       "latency-bound (frac_on_bytes_needed < 0.15)")
            return r
        roofline = roof(dom_name, dom, traffic, traffic_src)
        roofline["in_timed_step"] = "hot_path (batch %d per GPU, %s)" % (B, "batch-minor volume" if bm else "NCXYZ volume")
        roofline_soft = None
        if bm and "render_bwd_bm_soft" in rows:
            sr = rows["render_bwd_bm_soft"]
            st, st_src = pmc_traffic(sr.get("pmc", []), B, sr.get("src", ("common.hpp",)))
            roofline_soft = roof("render_bwd_bm_soft", sr, st, st_src)
            roofline_soft["in_timed_step"] = None
            roofline_soft["note"] = ("the renderer's backward on a volume whose every sample passes the clamps; no timed step renders "
                                     "such a volume (on GenRe's own the group writes zeros: kernels.render_bwd_bm)")
        # M2 at batch B: the pair launched alternately (each kernel after the other's traffic: what a trace sees); the sum of the two
        # back-to-back figures is reported beside it
        m2_us_b2b = rows["cam_bp_fwd"]["us"] + rows["calc_prob_fwd"]["us"]
        m2_us = rows["m2_pair_alternating"]["us"]
        m2_bytes = rows["cam_bp_fwd"]["bytes"] + rows["calc_prob_fwd"]["bytes"]
        b1 = batch1_graph(G, dev)
        headline_is_m1 = cap is not None and ms_per_step is not None and value != hot["shapes_per_s"]
        out = {
            "metric": "GenRe fwd shapes/sec (256\u00b2 RGB\u2192128\u00b3 vox) @1 GPU; cam_bp+calc_prob HBM GB/s vs roofline"
                      if headline_is_m1 else "hot-path shapes/sec (configs[1] fwd+bwd); GenRe forward not timed (--no-m1)",
            "value": value, "unit": "shapes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[2]-shape GenRe full_model forward (MarrNet-1 + depth_pred_with_sph_inpaint + voxel "
                                    "refiner, geometric ops in-stream), batch 1 per GPU, HIP-graph replay, seeded random weights "
                                    "(batch 8 in m1.batch8); the configs[1] hot-path step is `hot_path`") if headline_is_m1
                       else hot["what"],
                       "batch_per_gpu": 1 if headline_is_m1 else B, "target_fwd_per_s": 50.0,
                       "parallelism": "batch-sharded x%d, no collective" % world},
            "hot_path": hot,
            "roofline": roofline,
            "roofline_soft": roofline_soft,
            "roofline_m2": {"what": "BASELINE.json's second quantity: cam_bp fwd + calc_prob fwd, algorithmic bytes (50 593 792 "
                                    "per image) / time against 8 TB/s; target >= 0.40 at batch 1",
                            "batch1": {"frac": b1["frac"], "GBs": b1["GBs"], "us_per_image": b1["us_per_image"],
                                       "launch": "HIP-graph replay, one stream", "target_frac": 0.40},
                            "batch%d" % B: {"frac": m2_bytes / m2_us / 1e3 / HBM_PEAK_GBS, "GBs": m2_bytes / m2_us / 1e3,
                                            "us_per_image": m2_us / B, "us": m2_us,
                                            "timing": "HIP events, the two kernels launched alternately on one stream",
                                            "us_sum_of_back_to_back_figures": m2_us_b2b,
                                            "frac_sum_of_back_to_back_figures": m2_bytes / m2_us_b2b / 1e3 / HBM_PEAK_GBS,
                                            "rocprof": rocprof_m2()}},
            "m2": {"what": "cam_bp fwd + calc_prob fwd, algorithmic bytes / time, batch %d" % B,
                   "achieved": m2_bytes / m2_us / 1e3, "unit": "GB/s", "frac": m2_bytes / m2_us / 1e3 / HBM_PEAK_GBS,
                   "us_per_image": m2_us / B},
            "m2_batch1": {"what": "cam_bp fwd + calc_prob fwd at batch 1 (BASELINE.json: >= 40 % of 8 TB/s), HIP-graph replay",
                          "achieved": b1["GBs"], "unit": "GB/s", "frac": b1["frac"], "us_per_image": b1["us_per_image"],
                          "target_frac": 0.40, "two_streams": b1.get("two_streams")},
            "kernels": {k: dict({"us": round(v["us"], 2), "GBs": round(v["GBs"], 1), "in_step": k in in_step},
                                **({"us_in_step_order": round(v["us_in_step_order"], 2)} if "us_in_step_order" in v else {}),
                                **({"GBs_needed": round(v["GBs_needed"], 1), "tiles_live_frac": round(v["tiles_live_frac"], 3)}
                                   if "GBs_needed" in v else {}))
                        for k, v in rows.items()},
            "nnd": {"what": "Chamfer forward, both directions, %d x 2048 x 2048; 8 fp32 flops per pair, no fma" % B,
                    "TFLOPs": rows["nnd_fwd"]["TFLOPs"], "peak": 157.3, "frac": rows["nnd_fwd"]["frac_fp32_valu"],
                    "pairs_per_s": rows["nnd_fwd"]["pairs"] / rows["nnd_fwd"]["us"] * 1e6,
                    "cfg0_us": rows["nnd_fwd"]["cfg0_us"],
                    "cfg0_note": "configs[0] cloud pair (1 x 2048 x 2048) on the GPU, HIP-graph replay; the reference's "
                                 "CPU path for the same pair is cpu_baseline.nnd_cfg0_ms"},
            "forward_only": {"what": "configs[1] forward chain (no grad), batch %d, eager launches" % B,
                             "standard_layout_us": rows.get("chain_fwd", {}).get("us"),
                             "batch_minor_layout_us": rows.get("chain_fwd_batch_minor", {}).get("us"),
                             "shapes_per_s": (B / rows["chain_fwd_batch_minor"]["us"] * 1e6) if "chain_fwd_batch_minor" in rows else None},
            "batch1": b1,
            "hot_path_batch1": hot_path_batch1(G, dev),
            "rccl_ranks": world,
            "miopen_find_db": miopen_find_db_state(),
        }
        if not args.no_m1:
            if cap is not None:
                try:
                    m1 = m1_table(cap)
                    m1["timed_region"] = {"steps": args.steps, "warmup": args.warmup, "ms_per_forward": ms_per_step,
                                          "shapes_per_s": value, "per_rank_fwd_per_s": m1_per_rank,
                                          "note": "the bench line's `value`: barrier-bracketed, max over ranks"}
                    m1["kernel_trace"] = ("profiles/r05a_m1_b1_kernel_stats.txt, r05a_m1_b8_kernel_stats.txt (rocprofv3 "
                                          "--kernel-trace --stats of profiles/m1_target.py: no naive_conv_* kernel in the forward)")
                    geo = b1.get("genre_geometry_fwd_us_per_image")
                    if geo:
                        m1["geometry_us_batch1"] = geo
                        m1["geometry_share_batch1"] = geo / (m1["batch1"]["ms_per_forward"] * 1e3)
                    out["m1"] = m1
                except Exception as e:          # pragma: no cover -- never fatal for the bench line
                    out["m1"] = {"error": str(e)[:300]}
            else:
                out["m1"] = {"error": m1_err}
        if train is not None:
            out["train"] = train
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
