"""Networks around the geometric hot path: stock torch.nn on PyTorch-ROCm (MIOpen
Conv2d / Conv3d), re-hosted so that the GenRe / ShapeHD callers of the native ops exist in this tree.  Module trees
and parameter names equal the reference's (networks/uresnet.py, networks/revresnet.py, networks/networks.py), so its
released state_dicts load unchanged (tests/test_networks.py checks every key and shape against a fixture generated
from the reference's own classes)."""
from .resnet import resnet18, ResNet18                                   # noqa: F401
from .uresnet import Net, Net_inpaint, revuresnet18, RevResNet          # noqa: F401
from .networks import (ImageEncoder, VoxelDecoder, VoxelGenerator, VoxelDiscriminator, Unet_3D,    # noqa: F401
                       ViewAsLinear)
