"""U-Net on ResNet-18 blocks: the 2-D networks of GenRe / MarrNet.

`Net` maps an image to one map per named decoder (networks/uresnet.py:6-71 of the reference: MarrNet-1, RGB -> normal /
depth / silhouette); `Net_inpaint` is the spherical-map inpainting network (:74-145).  The decoder mirrors the
encoder with transposed convolutions (networks/revresnet.py:9-166) and concatenates the encoder feature map of the
same resolution after every stage.  Attribute names follow the reference so that state_dict keys are identical."""
import torch
from torch import nn

from .resnet import resnet18


def _deconv3(cin, cout, stride=1):
    return nn.ConvTranspose2d(cin, cout, 3, stride, 1, output_padding=stride - 1, bias=False)


class RevBasicBlock(nn.Module):
    """mirror image of a ResNet basic block: the (up-)stride sits on the second transposed convolution"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, upsample=None):
        super().__init__()
        self.deconv1 = _deconv3(inplanes, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.deconv2 = _deconv3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.upsample = upsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.deconv1(x)))
        y = self.bn2(self.deconv2(y))
        y = y + (x if self.upsample is None else self.upsample(x))
        return self.relu(y)


class RevResNet(nn.Module):
    """four stages of `block`s, then deconv1 (x2) + bn + relu + deconv2 (7x7, x2) (revresnet.py:102-166).
    planes[i]: output width of stage i; inplanes[i]: width arriving at stage i (after the U-Net concatenation)"""

    def __init__(self, block, layers, planes, inplanes=None, out_planes=5):
        super().__init__()
        inplanes = list(inplanes or [512])
        tail_in = inplanes[4] if len(inplanes) > 4 else planes[3]
        self.deconv1 = nn.ConvTranspose2d(tail_in, planes[3], 3, 2, 1, output_padding=1)
        self.deconv2 = nn.ConvTranspose2d(planes[3], out_planes, 7, 2, 3, output_padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes[3])
        self.relu = nn.ReLU(inplace=True)
        width = inplanes[0]
        for i in range(4):
            if 0 < i < len(inplanes):
                width = inplanes[i]
            stride = 2 if i < 3 else 1
            setattr(self, "layer%d" % (i + 1), self._stage(block, width, planes[i], layers[i], stride))
            width = planes[i]

    @staticmethod
    def _stage(block, cin, cout, n, stride):
        up = None
        if stride != 1 or cin != cout:
            up = nn.Sequential(nn.ConvTranspose2d(cin, cout, 1, stride, output_padding=stride - 1, bias=False),
                               nn.BatchNorm2d(cout))
        return nn.Sequential(block(cin, cout, stride, up), *[block(cout, cout) for _ in range(n - 1)])

    def forward(self, x):
        for i in range(1, 5):
            x = getattr(self, "layer%d" % i)(x)
        return self.deconv2(self.relu(self.bn1(self.deconv1(x))))


def revuresnet18(**kw):
    """decoder half of the U-ResNet-18: every stage receives its own output width plus the encoder's skip"""
    return RevResNet(RevBasicBlock, [2, 2, 2, 2], [256, 128, 64, 64], inplanes=[512, 512, 256, 128, 128], **kw)


def _encoder(input_planes):
    r = resnet18(pretrained=True)
    stem = r.conv1 if input_planes == 3 else nn.Conv2d(input_planes, 64, 7, 2, 3, bias=False)
    return nn.ModuleList([nn.Sequential(stem, r.bn1, r.relu, r.maxpool), r.layer1, r.layer2, r.layer3, r.layer4])


def _decoder(out_plane, last=None):
    r = revuresnet18(out_planes=out_plane)
    return nn.ModuleList([r.layer1, r.layer2, r.layer3, r.layer4,
                          nn.Sequential(r.deconv1, r.bn1, r.relu, r.deconv2 if last is None else last)])


class Net(nn.Module):
    """image -> {layer_name: map with out_planes[i] channels}"""

    def __init__(self, out_planes, layer_names, input_planes=3):
        super().__init__()
        self.encoder = _encoder(input_planes)
        self.encoder_out = None
        self.decoders = {}
        for planes, name in zip(out_planes, layer_names):
            self._add_decoder(name, _decoder(planes))

    def _add_decoder(self, name, modules):
        setattr(self, "decoder_" + name, modules)
        self.decoders[name] = modules

    def forward(self, im):
        feats = []
        x = im
        for stage in self.encoder:
            x = stage(x)
            feats.append(x)
        self.encoder_out = feats[-1]
        out = {}
        for name, dec in self.decoders.items():
            y = feats[-1]
            for i, stage in enumerate(dec):
                y = stage(y)
                if i + 1 < len(dec):
                    skip = feats[-(i + 2)]
                    assert skip.shape[2:] == y.shape[2:]
                    y = torch.cat((y, skip), 1)
            out[name] = y
        return out


class Net_inpaint(Net):
    """the inpainting U-ResNet: one 8x8 stride-2 transposed convolution `deconv2` (registered on the module itself
    AND as the last layer of every decoder, as in the reference -- both names appear in the state_dict) replaces
    the 7x7 one, so a 160x160 map comes back as 160x160"""

    def __init__(self, out_planes, layer_names, input_planes=3):
        nn.Module.__init__(self)
        self.encoder = _encoder(input_planes)
        self.encoder_out = None
        self.deconv2 = nn.ConvTranspose2d(64, 1, 8, 2, 3, bias=False)
        self.decoders = {}
        for planes, name in zip(out_planes, layer_names):
            self._add_decoder(name, _decoder(planes, last=self.deconv2))
