"""rocprofv3 target: the batch-1 chain of bench.py's hot_path_batch1 (one depth map -> cam_bp -> clamp -> render_spherical ->
sph_pad, forward + backward), eager launches, GenRe's volume (pre_scale 50) then the live-gradient variant (0.9).
Usage: rocprofv3 --kernel-trace --stats -d DIR -- python tools/prof_b1_chain.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch
import inputs
import genre_shapehd_amd as G
import bench
dev = torch.device("cuda:0")
d = torch.from_numpy(inputs.sphere_depth(noise_seed=2)).to(dev).requires_grad_(True)
gout = torch.randn((1, 1, 160, 160), device=dev)
net = bench.HotPath(G, True).to(dev)
which = [float(a) for a in sys.argv[1:]] or [50.0, 0.9]
for scale in which:
    for _ in range(8):
        d.grad = None
        out = net.render(net.cam(d), pre_scale=scale, pad=16)
        out.backward(gout)
    torch.cuda.synchronize()
