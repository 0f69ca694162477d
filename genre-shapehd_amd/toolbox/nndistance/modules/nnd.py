"""NNDModule -- drop-in for toolbox/nndistance/modules/nnd.py:5-7: the Chamfer pair (dist1, dist2) as an nn.Module,
so that it can sit in a model definition; no parameters, no state."""
import torch.nn as nn

from ..functions import nnd as _F


class NNDModule(nn.Module):
    def forward(self, input1, input2):
        """input1 [B,n,3], input2 [B,m,3] (GPU, fp32) -> (dist1 [B,n], dist2 [B,m]) squared nearest-neighbour distances"""
        dist1, dist2 = _F.nndistance(input1, input2)
        return dist1, dist2
