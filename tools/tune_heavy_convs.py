"""The two 8^3-kernel 3-D layers of Unet_3D that carry 64 of the GenRe forward's 102 GFLOP -- ConvTranspose3d(80 -> 20, k8 s2 p3) at
32^3 -> 64^3 and Conv3d(2 -> 20, k8 s2 p3) at 128^3 -> 64^3 (networks/networks.py:147-190 of the reference) -- timed alone at batch
1 and 8 under the MIOpen databases of $GENRE_MIOPEN_DIR.  Run once as is, once with MIOPEN_FIND_ENFORCE=3 (MIOpen tunes the solvers
that have tuning parameters and stores the result in its user perf-db), once more as is: the third run shows what the tuning bought.
usage (GPU box): GENRE_MIOPEN_DIR=gpurun_out/miopen [MIOPEN_FIND_ENFORCE=3] python tools/tune_heavy_convs.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import miopen_cache  # noqa: E402

miopen_cache.use(os.environ.get("GENRE_MIOPEN_DIR"), create=True)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
tag = "TUNE enforce=%s" % os.environ.get("MIOPEN_FIND_ENFORCE", "-")


def event_us(fn, iters=30, warm=3):
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


torch.manual_seed(0)
with torch.no_grad():
    for n in (1, 8):
        x = torch.randn(n, 80, 32, 32, 32, device=dev)
        w = torch.randn(80, 20, 8, 8, 8, device=dev) * 0.01
        t0 = time.time()
        us = event_us(lambda: F.conv_transpose3d(x, w, None, 2, 3))
        print("%s batch %d ConvTranspose3d(80->20,k8,s2,p3) 32^3: %.1f us = %.1f TFLOP/s (first call + timing %.0f s)"
              % (tag, n, us, 2 * n * 32 ** 3 * 80 * 20 * 512 / us / 1e6, time.time() - t0), flush=True)
        x = torch.randn(n, 2, 128, 128, 128, device=dev)
        w = torch.randn(20, 2, 8, 8, 8, device=dev) * 0.01
        t0 = time.time()
        us = event_us(lambda: F.conv3d(x, w, None, 2, 3))
        print("%s batch %d Conv3d(2->20,k8,s2,p3) 128^3: %.1f us = %.1f TFLOP/s (first call + timing %.0f s)"
              % (tag, n, us, 2 * n * 64 ** 3 * 20 * 2 * 512 / us / 1e6, time.time() - t0), flush=True)
