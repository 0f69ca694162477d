"""The GenRe full-model forward that bench.py's `value` times (models/genre_full_model.py:116-132 of the reference:
MarrNet-1, the geometric ops, the inpainting U-ResNet, the spherical back-projection, Unet_3D), eager launches, for
`rocprofv3 --kernel-trace --stats` (profiles/collect_pmc.sh).  Usage on the GPU box:
  cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d OUT -o m1 -- python profiles/m1_target.py <batch> <iters>
With `flops` as third argument it prints the FLOP count of one forward (torch.utils.flop_counter: convolutions and
matmuls of the three networks; the geometric ops are not FLOP-bound and are not counted) as one JSON line instead."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import miopen_cache  # noqa: E402

miopen_cache.use()
import torch  # noqa: E402
import genre_shapehd_amd  # noqa: E402,F401
from genre_shapehd_amd.models import GenReNet, GenReInference  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = GenReNet().to(dev).eval()
inf = GenReInference(net, device=dev, graph=False)
rgb = torch.rand(B, 3, 256, 256, device=dev)
sil = torch.zeros(B, 1, 256, 256, device=dev)
sil[:, :, 48:208, 48:208] = 100.0
for _ in range(3):
    inf.predict(rgb, sil)
torch.cuda.synchronize()
if len(sys.argv) > 3 and sys.argv[3] == "flops":
    from torch.utils.flop_counter import FlopCounterMode
    with FlopCounterMode(display=False) as fc:
        inf.predict(rgb, sil)
    print(json.dumps({"batch": B, "flops_per_forward": fc.get_total_flops(),
                      "by_op": {str(k): v for k, v in fc.get_flop_counts()["Global"].items()}}))
else:
    for _ in range(ITERS):
        inf.predict(rgb, sil)
torch.cuda.synchronize()
