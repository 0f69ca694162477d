"""The GenRe forward of the bench headline (HIP-graph replay, batch 1 and 8) under network-level variations, one MI355X.
Round 5 record (profiles/r05d_m1_rewrites_experiment.txt): folding the 137 eval-mode BatchNorms into their convolutions
5.569 -> 5.553 ms at batch 1 (PyTorch-ROCm adds a convolution's bias with a separate kernel: one launch replaced by another),
22.30 -> 21.98 ms at batch 8; Unet_3D's stride-2 ConvTranspose3d layers as eight (k/2)^3 convolutions per layer (no column buffer,
no col2im): 6.23 ms with the 8^3-kernel layer alone, 8.36 ms with all five -- MIOpen's forward solvers for those shapes lose to its
GEMM + col2im.  Neither was kept.  This script now times the memory-format variation:
  channels_last on the two 2-D U-ResNets (MIOpen's igemm_*_nhwc solvers are wrapped in batched_transpose kernels in the trace).
usage (GPU box): GENRE_MIOPEN_DIR=gpurun_out/miopen python tools/m1_experiments.py"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import miopen_cache  # noqa: E402

miopen_cache.use(os.environ.get("GENRE_MIOPEN_DIR"), create=True)
import torch  # noqa: E402
import genre_shapehd_amd  # noqa: E402,F401
from genre_shapehd_amd.models import GenReNet, GenReInference  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
net = GenReNet().to(dev).eval()


def event_ms(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def variant(name):
    v = copy.deepcopy(net)
    if name in ("cl_net1", "cl_both"):
        v.depth_and_inpaint.net1.to(memory_format=torch.channels_last)
    if name in ("cl_net2", "cl_both"):
        v.depth_and_inpaint.net2.to(memory_format=torch.channels_last)
    return v


res = {}
for n in (1, 8):
    rgb = torch.rand(n, 3, 256, 256, device=dev)
    sil = torch.zeros(n, 1, 256, 256, device=dev)
    sil[:, :, 48:208, 48:208] = 100.0
    ref = None
    for name in ("as_is", "cl_net1", "cl_net2", "cl_both"):
        inf = GenReInference(variant(name), device=dev, graph=True)
        x = rgb.contiguous(memory_format=torch.channels_last) if name in ("cl_net1", "cl_both") else rgb
        out = inf.predict(x, sil)["pred_voxel"].clone()
        g = inf._captured[tuple(x.shape)][0]
        ms = min(event_ms(g.replay) for _ in range(2))
        row = {"ms_per_forward": round(ms, 4), "shapes_per_s": round(n / ms * 1e3, 1)}
        if ref is None:
            ref = out
        else:
            row["max_abs_diff_to_as_is"] = float((out - ref).abs().max())
            row["logit_scale"] = float(ref.abs().max())
        res["batch%d %s" % (n, name)] = row
        print("M1X batch %d %-10s %s" % (n, name, json.dumps(row)), flush=True)
        del inf, g
        torch.cuda.empty_cache()
print(json.dumps(res))
