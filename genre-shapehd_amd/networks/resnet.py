"""ResNet-18 (He et al. 2015) with torchvision's attribute names -- the reference takes it from
torchvision.models.resnet18(pretrained=True) (networks/revresnet.py:6, uresnet.py:16, networks.py:13); torchvision is
not part of this image and there is no network, so the architecture is restated here.  `pretrained` is accepted and
ignored: weights come from a checkpoint (models/checkpoint.py) or stay at their seeded initialisation."""
import torch
from torch import nn


class _Block(nn.Module):
    """two 3x3 convolutions with identity (or 1x1-projected) shortcut; stride on the first one"""
    expansion = 1

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


class ResNet18(nn.Module):
    STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))          # (width, stride of the first block), two blocks each

    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (width, stride) in enumerate(self.STAGES, 1):
            setattr(self, "layer%d" % i, nn.Sequential(_Block(cin, width, stride), _Block(width, width, 1)))
            cin = width
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():                              # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        for i in range(1, 5):
            x = getattr(self, "layer%d" % i)(x)
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet18(pretrained=False, **kw):
    return ResNet18(**kw)
