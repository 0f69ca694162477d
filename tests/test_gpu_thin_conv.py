"""networks/thin_conv.py on the GPU at the layer shapes it exists for: forward (MIOpen, stock), data gradient (the adjoint
convolution on MIOpen) and the blocked-GEMM weight gradient against the stock operator in float64 on the CPU, within the
fp32 dot-product bound of the reduction length (tests/test_gpu_z_train.py, statement 1)."""
import copy
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
U32 = 2.0 ** -24


@pytest.mark.parametrize("cls,cin,cout,k,s,p,size,n", [
    ("t", 32, 1, 4, 2, 1, 64, 2),          # MarrNet-2 / ShapeHD decoder output (marrnet2.py:88-111)
    ("t", 64, 1, 4, 2, 1, 64, 2),          # 3-D GAN generator output
    ("t", 40, 1, 4, 2, 1, 64, 2),          # Unet_3D dec6
    ("c", 2, 20, 8, 2, 3, 128, 2),         # Unet_3D enc1
    ("t", 64, 32, 4, 2, 1, 32, 2),         # MarrNet-2 decoder 32^3 -> 64^3
    ("c", 20, 40, 4, 2, 1, 64, 2),         # Unet_3D enc2
    ("t", 80, 20, 8, 2, 3, 32, 2),         # Unet_3D dec5
    ("t", 8, 1, 4, 2, 1, 10, 3), ("c", 2, 5, 8, 2, 3, 18, 3)])
def test_thin_convolutions_on_the_gpu_against_float64(cls, cin, cout, k, s, p, size, n, genre, dev):
    from genre_shapehd_amd.networks import thin_conv as TC
    torch.manual_seed(cin + cout + size)
    mod64 = (TC.ThinConvTranspose3d if cls == "t" else TC.ThinConv3d)(cin, cout, k, s, p).double()
    mod64.force_stock = True
    x64 = torch.randn((n, cin, size, size, size), dtype=torch.float64, requires_grad=True)
    y64 = mod64(x64)
    g64 = torch.randn_like(y64)
    y64.backward(g64)
    mod = copy.deepcopy(mod64).float().to(dev)
    mod.force_stock = False
    mod.zero_grad()
    x = x64.detach().float().to(dev).requires_grad_(True)
    mod.force_custom = size < 32                                    # (the two small cases: the GEMM path all the same)
    y = mod(x)
    y.backward(g64.float().to(dev))
    kk, out_sp = k ** 3, math.prod(y64.shape[2:])
    red = {"y": cin * kk, "dx": cout * kk, "dw": n * (size ** 3 if cls == "t" else out_sp), "db": n * out_sp}
    for name, a, b in (("y", y, y64), ("dx", x.grad, x64.grad), ("dw", mod.weight.grad, mod64.weight.grad),
                       ("db", mod.bias.grad, mod64.bias.grad)):
        err = ((a.detach().double().cpu() - b.detach()).abs().max() / b.detach().abs().max()).item()
        bar = 8.0 * math.sqrt(red[name]) * U32
        print("%s: %.2e (bound %.2e)" % (name, err, bar))
        assert err <= bar, (name, err, bar)
