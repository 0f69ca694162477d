"""3-D networks of GenRe / ShapeHD on stock torch.nn (MIOpen Conv3d / ConvTranspose3d on ROCm):

  Unet_3D             GenRe's voxel refiner, [N,2,128^3] -> [N,1,128^3] logits (reference networks/networks.py:147-190)
  ImageEncoder        ResNet-18 on 2.5-D maps -> 200-d shape code                       (:6-23)
  VoxelDecoder        200-d code -> 128^3 voxel logits (MarrNet-2 / ShapeHD decoder)    (:26-64)
  VoxelGenerator      3-D WGAN-GP generator, 200-d noise -> 64^3 / 128^3               (:67-108)
  VoxelDiscriminator  3-D WGAN-GP critic                                               (:111-144)

Layer lists are written as tables; `nn.Sequential` indices and attribute names equal the reference's so that its
state_dicts load (including the two empty placeholders inside VoxelDecoder.main that keep old checkpoints aligned)."""
import torch
from torch import nn

from .resnet import resnet18
from .thin_conv import ThinConv3d, ThinConvTranspose3d


def _up3(cin, cout, bias):                 # x2
    # at >= 64^3 MIOpen's weight gradient is its naive reference kernel (283 ms for the 1-channel output layer at batch 8) or a
    # 40 ms CK kernel: thin_conv.py computes it as a blocked GEMM there and is nn.ConvTranspose3d everywhere else
    return ThinConvTranspose3d(cin, cout, 4, 2, 1, bias=bias)


def _grow3(cin, cout, bias):               # 1^3 -> 4^3
    return nn.ConvTranspose3d(cin, cout, 4, 1, 0, bias=bias)


def _down3(cin, cout, bias):               # /2
    return ThinConv3d(cin, cout, 4, 2, 1, bias=bias)


def _bn_relu3(c):
    return [nn.BatchNorm3d(c), nn.ReLU(inplace=True)]


class ViewAsLinear(nn.Module):
    @staticmethod
    def forward(x):
        return x.reshape(x.shape[0], -1)


class ImageEncoder(nn.Module):
    def __init__(self, input_nc, encode_dims=200):
        super().__init__()
        r = resnet18(pretrained=True)
        r.conv1 = nn.Conv2d(input_nc, 64, 7, 2, 3, bias=False)
        r.avgpool = nn.AdaptiveAvgPool2d(1)
        r.fc = nn.Linear(512, encode_dims)
        self.main = nn.Sequential(r)

    def forward(self, x):
        return self.main(x)


class VoxelDecoder(nn.Module):
    def __init__(self, n_dims=200, nf=512):
        super().__init__()
        layers = [_grow3(n_dims, nf, True), *_bn_relu3(nf),
                  _up3(nf, nf // 2, True), *_bn_relu3(nf // 2),
                  nn.Sequential(), nn.Sequential()]               # placeholders: indices of released checkpoints
        c = nf // 2
        for _ in range(3):
            layers += [_up3(c, c // 2, True), *_bn_relu3(c // 2)]
            c //= 2
        layers.append(_up3(c, 1, True))
        self.main = nn.Sequential(*layers)

    def forward(self, x):
        return self.main(x.reshape(x.size(0), -1, 1, 1, 1))


class VoxelGenerator(nn.Module):
    def __init__(self, nz=200, nf=64, bias=False, res=128):
        super().__init__()
        if res not in (64, 128):
            raise NotImplementedError(res)
        layers = [_grow3(nz, nf * 8, bias), *_bn_relu3(nf * 8)]
        for cin, cout in ((8, 4), (4, 2), (2, 1)):                # 4^3 -> 32^3
            layers += [_up3(nf * cin, nf * cout, bias), *_bn_relu3(nf * cout)]
        if res == 128:
            layers += [_up3(nf, nf, bias), *_bn_relu3(nf)]        # -> 64^3
        layers += [_up3(nf, 1, bias), nn.Sigmoid()]
        self.main = nn.Sequential(*layers)

    def forward(self, x):
        return self.main(x)


class VoxelDiscriminator(nn.Module):
    def __init__(self, nf=64, bias=False, res=128):
        super().__init__()
        if res not in (64, 128):
            raise NotImplementedError(res)
        widths = [1, nf] + ([nf] if res == 128 else []) + [nf * 2, nf * 4, nf * 8]
        layers = []
        for cin, cout in zip(widths[:-1], widths[1:]):
            layers += [_down3(cin, cout, bias), nn.LeakyReLU(0.2, inplace=True)]
        layers.append(nn.Conv3d(nf * 8, 1, 4, 1, 0, bias=bias))    # 4^3 -> 1
        self.main = nn.Sequential(*layers)

    def forward(self, x):
        return self.main(x).reshape(-1, 1).squeeze(1)


class Conv3d_block(nn.Module):
    def __init__(self, ncin, ncout, kernel_size, stride, pad, dropout=False):
        super().__init__()
        self.net = nn.Sequential(ThinConv3d(ncin, ncout, kernel_size, stride, pad), nn.BatchNorm3d(ncout), nn.LeakyReLU())

    def forward(self, x):
        return self.net(x)


class Deconv3d_skip(nn.Module):
    def __init__(self, ncin, ncout, kernel_size, stride, pad, extra=0, is_activate=True):
        super().__init__()
        up = ThinConvTranspose3d(ncin, ncout, kernel_size, stride, pad, extra)
        self.net = nn.Sequential(up, nn.BatchNorm3d(ncout), nn.LeakyReLU()) if is_activate else up

    def forward(self, x, skip_in):
        return self.net(torch.cat((x, skip_in), 1))


class Unet_3D(nn.Module):
    # (kernel, stride, pad): six encoder stages 128 -> 64 -> 32 -> 16 -> 8 -> 4 -> 1 and the six decoder stages back up
    # (the 8^3 kernel of the decoder sits on the 32 -> 64 stage, not on the last one: networks.py:163-165)
    ENC = ((8, 2, 3), (4, 2, 1), (4, 2, 1), (4, 2, 1), (4, 2, 1), (4, 1, 0))
    DEC = ((4, 1, 0), (4, 2, 1), (4, 2, 1), (4, 2, 1), (8, 2, 3), (4, 2, 1))

    def __init__(self, nf=20, in_channel=2, no_linear=False):
        super().__init__()
        self.nf = nf
        self.no_linear = no_linear
        w = [in_channel] + [nf << i for i in range(6)]            # 2, nf, 2nf, ... 32nf
        for i, (k, s, p) in enumerate(self.ENC):
            setattr(self, "enc%d" % (i + 1), Conv3d_block(w[i], w[i + 1], k, s, p))
        self.full_conv_block = nn.Sequential(nn.Linear(32 * nf, 32 * nf), nn.LeakyReLU())
        for i in range(6):                                          # dec(i+1) undoes enc(6-i); input = previous + skip
            k, s, p = self.DEC[i]
            cout = w[5 - i] if i < 5 else 1
            setattr(self, "dec%d" % (i + 1), Deconv3d_skip(2 * w[6 - i], cout, k, s, p, 0, is_activate=i < 5))

    def forward(self, x):
        enc = []
        for i in range(1, 7):
            x = getattr(self, "enc%d" % i)(x)
            enc.append(x)
        y = enc[5]
        if not self.no_linear:
            y = self.full_conv_block(y.reshape(y.size(0), self.nf * 32)).reshape(y.size(0), self.nf * 32, 1, 1, 1)
        for i in range(1, 7):
            y = getattr(self, "dec%d" % i)(y, enc[6 - i])
        return y
