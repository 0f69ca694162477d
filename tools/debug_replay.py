"""debug: which stage of the GenRe forward differs between two runs / two graph replays"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import genre_shapehd_amd as G
import networks_fill as NF
from genre_shapehd_amd.models import GenReNet, Inputs
dev = torch.device("cuda:0")
net = NF.fill_state(GenReNet(), seed=3).eval().to(dev)
rng = np.random.default_rng(9)
rgb = torch.from_numpy(rng.uniform(0, 1, (1, 3, 256, 256)).astype(np.float32)).to(dev)
ax = np.linspace(-1, 1, 256)
sil = torch.from_numpy((((ax[:, None] ** 2 + ax[None, :] ** 2) < 0.5).astype(np.float32)[None, None]) * 100).to(dev)
keys = ["depth", "depth_minmax", "proj_depth", "pred_sph_partial", "pred_sph_full", "pred_proj_sph_full", "pred_proj_depth", "pred_voxel"]
with torch.no_grad():
    o1 = {k: v.clone() for k, v in net(Inputs(rgb, sil)).items()}
    o2 = {k: v.clone() for k, v in net(Inputs(rgb, sil)).items()}
for k in keys:
    print("eager run-to-run %-20s max diff %.3e  (scale %.3e)" % (k, (o1[k] - o2[k]).abs().max().item(), o1[k].abs().max().item()))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    net(Inputs(rgb, sil))
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g):
    og = net(Inputs(rgb, sil))
g.replay(); torch.cuda.synchronize(); r1 = {k: og[k].clone() for k in keys}
g.replay(); torch.cuda.synchronize(); r2 = {k: og[k].clone() for k in keys}
for k in keys:
    print("replay-to-replay %-20s %.3e   replay vs eager %.3e" % (k, (r1[k] - r2[k]).abs().max().item(), (r1[k] - o1[k]).abs().max().item()))
