"""STNv2b / STNv2c: predict one translation per image layer and re-place the layers.

Same classes, constructor arguments and ``state_dict`` keys (``localization.*``, ``fc_loc.*``) as the
reference (fukuwarai/networks.py:148-258).  The localisation CNN is plain conv/pool/relu (MIOpen);
what changes is the sampling step: upstream builds B*L 2x3 matrices in a python double loop, a
[B*L,H,W,2] grid with ``affine_grid`` and gathers with ``grid_sample``; here the predicted
translations go straight into ``translate_sample`` (csrc/translate.hip), which also folds in the
+1/-1 workaround of STNv2c.
"""
import torch
import torch.nn as nn

from ..custom_utils.image_utils import convert_translate_to_2x3
from ..torch_utils.ops.translate_sample import translate_sample


class STNv2b(nn.Module):
    """Translation-only STN for inputs in [0,1] (zero padding reads as 0)."""
    shift = 0.0

    def __init__(self, img_resolution, img_channels, img_layers, nf1=64, nf2=64):
        super().__init__()
        self.img_resolution, self.img_channels, self.img_layers = img_resolution, img_channels, img_layers
        chans = [img_channels * img_layers, nf1, nf1 * 2, nf1 * 4, nf1 * 6, nf1 * 8]
        kernels = [7, 5, 3, 3, 3]
        layers = []
        for cin, cout, k in zip(chans[:-1], chans[1:], kernels):
            layers += [nn.Conv2d(cin, cout, kernel_size=k), nn.MaxPool2d(2, stride=2), nn.ReLU(True)]
        self.localization = nn.Sequential(*layers)
        with torch.no_grad():
            self.len_loc = self.localization(torch.zeros(1, chans[0], img_resolution, img_resolution)).numel()
        self.fc_loc = nn.Sequential(nn.Linear(self.len_loc, nf2), nn.ReLU(True), nn.Linear(nf2, img_layers * 2))
        # start from the identity placement (reference :201-203)
        self.fc_loc[2].weight.data.zero_()
        self.fc_loc[2].bias.data.zero_()

    def forward(self, x):
        """[B,L,C,H,W] -> (re-placed [B,L,C,H,W], theta [B,L,2,3])."""
        b, l, c, h, w = x.shape
        feat = self.localization(x.reshape(b, l * c, h, w)).reshape(-1, self.len_loc)
        translation = self.fc_loc(feat).reshape(b, l, 2)
        theta = convert_translate_to_2x3(translation)
        y = translate_sample(x.reshape(b * l, c, h, w), translation.reshape(b * l, 2), shift=self.shift)
        return y.reshape(b, l, c, h, w), theta


class STNv2c(STNv2b):
    """Same for inputs in [-1,1]: sample x+1 and subtract 1, so out-of-frame pixels read -1
    (reference :226-258)."""
    shift = 1.0
