"""Generates tests/golden/networks_keys.json and networks_golden.npz FROM THE REFERENCE'S OWN network classes
(/root/reference/networks/{uresnet,revresnet,networks}.py), imported in this container:

  * the state_dict keys and shapes of every network the GenRe / ShapeHD callers use, and
  * forward outputs (sub-sampled + checksums) for key-seeded weights (tests/networks_fill.py) in eval mode.

torchvision is not installed here; the reference only takes `resnet18` from it, so a stub module hands it the
repo's restatement (genre-shapehd_amd/networks/resnet.py) -- torchvision's ResNet-18 attribute names are what the
reference's checkpoints were saved with.  Run from the repo root:  python tests/golden/make_networks_golden.py"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import genre_shapehd_amd  # noqa: E402,F401
from genre_shapehd_amd.networks.resnet import resnet18  # noqa: E402
import networks_fill as NF  # noqa: E402

tv = types.ModuleType("torchvision")
tv.models = types.ModuleType("torchvision.models")
tv.models.resnet18 = resnet18
sys.modules["torchvision"] = tv
sys.modules["torchvision.models"] = tv.models
sys.path.insert(0, "/root/reference")
import networks.uresnet as ref_u  # noqa: E402
import networks.networks as ref_n  # noqa: E402

ns = types.SimpleNamespace(Net=ref_u.Net, Net_inpaint=ref_u.Net_inpaint, Unet_3D=ref_n.Unet_3D,
                           ImageEncoder=ref_n.ImageEncoder, VoxelDecoder=ref_n.VoxelDecoder,
                           VoxelGenerator=ref_n.VoxelGenerator, VoxelDiscriminator=ref_n.VoxelDiscriminator)
torch.set_num_threads(8)
keys, gold = {}, {}
for name, shape in NF.cases():
    net = NF.build(ns, name)
    keys[name] = {k: list(v.shape) for k, v in net.state_dict().items()}
    NF.fill_state(net).eval()
    with torch.no_grad():
        out = net(NF.make_input(shape))
    for k, (sub, s, a) in NF.digest(out).items():
        gold["%s/%s/sub" % (name, k)] = sub
        gold["%s/%s/sums" % (name, k)] = np.array([s, a])
    print(name, "ok", len(keys[name]), "tensors")
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "networks_keys.json"), "w") as f:
    json.dump(keys, f, indent=0, sort_keys=True)
np.savez_compressed(os.path.join(here, "networks_golden.npz"), **gold)
