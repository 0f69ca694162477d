"""Runs each hand-written kernel of the step a few times at batch 32 so that
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes)
can attribute HBM-side traffic per dispatch.  Usage on the GPU box:
  cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE -d OUT -o fetch -- python profiles/pmc_targets.py
  cd /tmp && TMPDIR=/tmp rocprofv3 --pmc WRITE_SIZE -d OUT -o write -- python profiles/pmc_targets.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import inputs  # noqa: E402
import genre_shapehd_amd as G  # noqa: E402
from genre_shapehd_amd.toolbox import _fused_render  # noqa: E402
from genre_shapehd_amd.toolbox.cam_bp.cam_bp._ext import cam_bp_lib  # noqa: E402
from genre_shapehd_amd.toolbox.calc_prob.calc_prob._ext import calc_prob_lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 3        # (the kernel-trace pass of collect_pmc.sh asks for more)
dev = torch.device("cuda:0")
d = torch.from_numpy(inputs.batch_depth(B)).to(dev)
fl = torch.full((B, 1), 418.3, device=dev)
cd = torch.full((B, 1), 2.2, device=dev)
tdf = torch.empty((B, 1, 128, 128, 128), device=dev)
cnt = torch.empty_like(tdf)
p = torch.rand((B, 1, 128, 128, 256), device=dev).clamp_(1e-5, 1 - 1e-5)
s = torch.empty_like(p)
g = torch.randn_like(p)
o = torch.empty_like(p)
mod = G.render_spherical(fused=True).to(dev)
dirs = mod._dirs64.view(torch.float32)
T = _fused_render.tables_for(tdf.shape, dev, mod._dirs64, 256)
out = torch.empty((B, 1, 128, 128), device=dev)
gout = torch.randn_like(out)
vbuf = torch.empty((B * 128 * 128 * 256,), device=dev)
scratch = torch.empty((vbuf.numel() + max(4, B),), device=dev)
gvox = torch.empty_like(tdf)
lib = _fused_render._loader().render_lib
# batch-minor tile renderer (csrc/sph_render_bm.hip)
layer = G.Camera_back_projection_layer(batch_minor=True).to(dev)
with torch.no_grad():
    proj_bm = layer(d)
TB = _fused_render.bm_tables_for(proj_bm.shape, dev, mod._dirs64, mod.depth_weight) if B >= 16 else None
cnt_bm = _fused_render.empty_batch_minor(proj_bm.shape, torch.float32, dev) if B >= 16 else None
if TB is not None:
    groups = -(-B // 32)
    ps = torch.empty((groups * TB["segs"].shape[0] * 64,), device=dev)
    trs = torch.empty((ps.numel(),), device=dev)
    stash = torch.empty((groups * TB["rec_f"].shape[0] * 32,), device=dev)
    mask = torch.empty((groups * 128 ** 3 + groups,), dtype=torch.int32, device=dev)
    out_p = torch.empty((B, 1, 160, 160), device=dev)
    gout_p = torch.randn_like(out_p)
    gvox_bm = _fused_render.empty_batch_minor(proj_bm.shape, torch.float32, dev)
# The renderers are profiled in PHASES (profiles/phases.py): first on GenRe's own volume -- what the timed steps render, as the
# step's layer hands it over, with its occupancy words; the x50 clamp blocks every voxel there, the backward kernels write zeros
# (bench.py: kernels.render_fwd_fused / render_fwd_bm / render_bwd_*, and `roofline`) --, the segment forward once more on the
# same volume WITHOUT the words (@dense), then on the SOFT volume (every sample passes the clamps: a gradient everywhere;
# kernels.*_soft, `roofline_soft`).  The summarisers split every renderer kernel's dispatches by launch order.
live = torch.empty((B * (1 + 512),), dtype=torch.int32, device=dev)
layer_std = G.Camera_back_projection_layer().to(dev)
with torch.no_grad():
    proj_std = layer_std(d)                  # dense NCXYZ, with the camera brick kernel's cell words
S = _fused_render.seg_tables_for(proj_std.shape, dev, mod._dirs64, mod.depth_weight)
ps_std = torch.empty((B * S["smax"] * 128 * 128 * 2,), device=dev)
occ_std = _fused_render.occupancy_hint_std(proj_std, S, mod._dirs64, mod.depth_weight, 50.0, lib, with_grad=True)


tr_std = _fused_render.seg_tr_scratch(ps_std, proj_std, mod._dirs64)
halo_std = _fused_render.seg_halo_scratch(S, proj_std)
vseg = _fused_render.seg_v_scratch(S, B, dev)


def seg_fwd(vol, hint, save=True):
    lib.render_seg_forward(vol, dirs, mod.depth_weight, out, S["seg_rows"], S["segs"], S["ray_nseg"], S["ray_pre"], S["line_w"], ps_std, 50.0,
                           live, *(occ_std if hint else (None, None, 0)), vseg if save else None)
gsoft = torch.Generator(device="cpu").manual_seed(1)
soft_std = ((torch.rand(tdf.shape, generator=gsoft) * 0.9 + 0.05) * 0.02).to(dev)
soft_bm = None
if TB is not None:
    soft_bm = _fused_render.empty_batch_minor(proj_bm.shape, torch.float32, dev)
    soft_bm.copy_(soft_std)
for _ in range(ITERS):
    cam_bp_lib.back_projection_forward(d, cd, fl, tdf, cnt)              # (the call and the inputs of bench.py: kernels.cam_bp_fwd)
    calc_prob_lib.calc_prob_forward(p, s)
    calc_prob_lib.calc_prob_backward_fused(p, s, g, o)
    if TB is not None:
        cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, proj_bm, cnt_bm, shifted=True)   # image-minor volumes: fill + leader pass
        tl_bm = _fused_render.new_brick_words(B, 128, dev)
        cam_bp_lib.back_projection_forward_const(d, 2.2, 418.3, proj_bm, cnt_bm, shifted=True,   # ... as the layer calls it in the step
                                                 tile_live=tl_bm, sparse_cnt=True)
for vol_std, vol_bm in ((proj_std, proj_bm), (None, None), (soft_std, soft_bm)):
    if vol_std is None:                      # @dense: the segment forward on GenRe's volume without the occupancy words ...
        for _ in range(ITERS):
            seg_fwd(proj_std, False, save=False)
        if TB is not None:                   # ... @nosave: the image-minor forward as the step runs it on GenRe's volume (the clamp provably
            for _ in range(ITERS):           # blocks every voxel: nothing saved; toolbox/_fused_render.py: provably_blocked)
                _fused_render.attach_hint(proj_bm, tl_bm, 128)
                words, ps_empty = _fused_render.occupancy_hint(proj_bm, TB, 50.0, lib, with_grad=False)
                lib.render_bm_forward(proj_bm, out_p, TB["segs"], TB["rec_f"], TB["fwd_rows"], TB["ray_ptr"], TB["ray_seg"],
                                      TB["ray_pre"], ps, None, None, 50.0, words, ps_empty)
        continue
    for _ in range(ITERS):
        seg_fwd(vol_std, vol_std is proj_std)
        lib.render_seg_backward(vol_std, dirs, mod.depth_weight, gout, gvox, S["bwd_rows"], S["segs"], S["ray_nseg"], S["ray_pre"],
                                S["line_w"], ps_std, tr_std, vseg, halo_std, 50.0, live)
        if TB is not None:
            if vol_bm is proj_bm:            # (the raw-ABI camera calls above dropped the layer's hint: same values, words of the last call)
                _fused_render.attach_hint(proj_bm, tl_bm, 128)
            words, ps_empty = (None, None) if vol_bm is soft_bm else _fused_render.occupancy_hint(proj_bm, TB, 50.0, lib, with_grad=True)
            lib.render_bm_forward(vol_bm, out_p, TB["segs"], TB["rec_f"], TB["fwd_rows"], TB["ray_ptr"], TB["ray_seg"],
                                  TB["ray_pre"], ps, stash, mask, 50.0, words, ps_empty)
            lib.render_bm_backward(gout_p, gvox_bm, TB["segs"], TB["ray_ptr"], TB["ray_seg"], TB["ray_pre"], TB["ent"],
                                   TB["rec_b"], TB["bwd_rows"], mod.depth_weight, ps, trs, stash, mask, 50.0, TB["pull_code"])
# Chamfer forward (VALU-bound: used with --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES ...)
from genre_shapehd_amd.toolbox.nndistance._ext import my_lib  # noqa: E402
xa = torch.rand((B, 2048, 3), device=dev)
xb = torch.rand((B, 2048, 3), device=dev)
d1 = torch.empty((B, 2048), device=dev); d2 = torch.empty_like(d1)
i1 = torch.empty((B, 2048), device=dev, dtype=torch.int32); i2 = torch.empty_like(i1)
for _ in range(3):
    my_lib.nnd_forward_cuda(xa, xb, d1, d2, i1, i2)
torch.cuda.synchronize()
