"""Compiles (once) every MIOpen kernel that bench.py's M1 and `train` sections and the GPU tests of the networks need, into
the kernel cache / user database under $GENRE_MIOPEN_DIR (default: genre-shapehd_amd/.miopen, where bench.py and the tests look
for it).  On a fresh box MIOpen compiles a HIP kernel per convolution configuration -- ~300 of them for the three GenRe
networks, MarrNet-2 and the 3-D GAN, forward, data- and weight-gradient: ~8 minutes -- before the first train step runs; the
cache is a build artefact like libgenre_hip.so (git-ignored, travels with the working tree).
usage (GPU box): GENRE_MIOPEN_DIR=gpurun_out/miopen python tools/warm_miopen.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import miopen_cache  # noqa: E402

miopen_cache.use(os.environ.get("GENRE_MIOPEN_DIR"), create=True)
import torch  # noqa: E402
import genre_shapehd_amd  # noqa: E402,F401
from genre_shapehd_amd import train as T  # noqa: E402
from genre_shapehd_amd.models import shapehd as MS  # noqa: E402
from genre_shapehd_amd.models.genre import GenReNet, GenReOptions, GenReInference  # noqa: E402

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True           # MIOpen's measured find per shape: what the find-db should hold (train.py sets it too)
t0 = time.time()
to = lambda ns: type(ns)(**{k: v.to(dev) for k, v in vars(ns).items()})       # noqa: E731
torch.manual_seed(0)
net = GenReNet().to(dev).eval()
for n in (1, 8):                                                          # M1: forward, batch 1 and 8
    inf = GenReInference(net, device=dev, graph=False)
    inf.predict(torch.rand(n, 3, 256, 256), torch.full((n, 1, 256, 256), 100.0))
print("GenRe forward: %.0f s" % (time.time() - t0), flush=True)
del net
shd = MS.ShapeHDNet().to(dev).train()
ins, vox = T.sketch_batch(8, "cpu", seed=1)
T.shapehd_train_step(shd, torch.optim.Adam(shd.marrnet2.parameters(), lr=1e-4), to(ins), vox.to(dev), 1e-3)
ins2, vox2 = T.sketch_batch(2, "cpu", seed=1)
T.shapehd_train_step(shd, torch.optim.Adam(shd.marrnet2.parameters(), lr=1e-4), to(ins2), vox2.to(dev), 1e-3)
print("ShapeHD step: %.0f s" % (time.time() - t0), flush=True)
del shd
gopt = GenReOptions(joint_train=True)
g = GenReNet(gopt).to(dev).train()
for n in (4, 2):
    gin, gt = T.genre_batch(n, "cpu", seed=2)
    T.genre_train_step(g, torch.optim.Adam(g.parameters(), lr=1e-6), to(gin), to(gt), gopt, chamfer_weight=0.1)
print("GenRe joint step: %.0f s" % (time.time() - t0), flush=True)
del g
if "--no-gan" not in sys.argv:
    gan = MS.WGANGP(lr=1e-4)
    gan.net_g.to(dev), gan.net_d.to(dev)
    for n in (8, 2):
        gan.train_on_batch(0, (torch.rand(n, 1, 128, 128, 128, device=dev) > 0.7).float())
    print("WGAN-GP step: %.0f s" % (time.time() - t0), flush=True)
torch.cuda.synchronize()
print("done in %.0f s; cache at %s" % (time.time() - t0, miopen_cache.current()))
