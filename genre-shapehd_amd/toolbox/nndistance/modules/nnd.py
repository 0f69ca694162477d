"""NNDModule -- drop-in for toolbox/nndistance/modules/nnd.py:5-7."""
from torch.nn import Module

from ..functions.nnd import nndistance


class NNDModule(Module):
    def forward(self, input1, input2):
        return nndistance(input1, input2)
