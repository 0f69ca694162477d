"""the two PyTorch-side knobs of the train steps, one at a time, fp32 only --
  benchmark       torch.backends.cudnn.benchmark = True (MIOpen's exhaustive find instead of the find-db / immediate mode)
  channels_last   memory_format=torch.channels_last_3d on the 3-D networks (and channels_last on the 2-D ones)
against the default, for the ShapeHD step (batch 8) and the GenRe joint step (batch 4).  usage (GPU box):
  python tools/train_knobs.py [default|benchmark|channels_last]      (one mode per process; prints ms per step)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import miopen_cache  # noqa: E402

miopen_cache.use()
import torch  # noqa: E402
import genre_shapehd_amd  # noqa: E402,F401
from genre_shapehd_amd import train as T  # noqa: E402
from genre_shapehd_amd.models import shapehd as MS  # noqa: E402
from genre_shapehd_amd.models.genre import GenReNet, GenReOptions  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
dev = torch.device("cuda:0")
if mode == "benchmark":
    torch.backends.cudnn.benchmark = True
to = lambda ns: type(ns)(**{k: v.to(dev) for k, v in vars(ns).items()})       # noqa: E731


def fmt(net):
    if mode != "channels_last":
        return net
    for m in net.modules():
        if isinstance(m, (torch.nn.Conv3d, torch.nn.ConvTranspose3d)):
            m.to(memory_format=torch.channels_last_3d)
        elif isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            m.to(memory_format=torch.channels_last)
    return net


def timed(fn, steps=6):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


torch.manual_seed(1234)
res = {}
net = fmt(MS.ShapeHDNet().to(dev).train())
ins, vox = T.sketch_batch(8, "cpu", seed=500)
ins, vox = to(ins), vox.to(dev)
optim = torch.optim.Adam(net.marrnet2.parameters(), lr=1e-4, betas=(0.5, 0.9))
try:
    res["shapehd_b8_ms"] = round(timed(lambda: T.shapehd_train_step(net, optim, ins, vox, 1e-3)), 1)
except Exception as e:
    res["shapehd_b8_ms"] = repr(e)[:200]
del net, optim
torch.cuda.empty_cache()
gopt = GenReOptions(joint_train=True)
g = GenReNet(gopt).to(dev).train()
with torch.no_grad():
    head = g.depth_and_inpaint.net1.decoder_minmax[9]
    head.weight.zero_()
    head.bias.copy_(torch.tensor([1.9, 2.4]))
g = fmt(g)
optim = torch.optim.Adam(g.parameters(), lr=1e-6, betas=(0.5, 0.9))
gin, gt = T.genre_batch(4, "cpu", seed=600)
gin, gt = to(gin), to(gt)
try:
    res["genre_joint_b4_ms"] = round(timed(lambda: T.genre_train_step(g, optim, gin, gt, gopt, chamfer_weight=0.1)), 1)
except Exception as e:
    res["genre_joint_b4_ms"] = repr(e)[:200]
print("KNOB %-14s %s" % (mode, res), flush=True)
