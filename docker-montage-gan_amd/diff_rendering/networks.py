"""Renderer CNNs: [B,L,C,H,W] layer stack -> blended [B,C,H,W].

Same classes and ``state_dict`` keys as the reference (diff_rendering/networks.py:6-134), including
the quirk that ``block`` is ONE module applied several times and therefore appears under both
``block.*`` and ``cnn.{2,3,4}.*``, so the shipped checkpoints
(pretrained/diff_rendering/*/renderer032000.pth.tar) load unchanged.
Plain convolutions: this is MIOpen work, there is nothing to hand-write here.  The analytic
alternative used when the renderer is bypassed is ``alpha_composite`` (csrc/composite.hip).
"""
import math

import torch.nn as nn


def _shared_block(nf):
    return nn.Sequential(nn.Conv2d(nf, nf, kernel_size=3, padding=1), nn.ReLU(True))


class _RendererBase(nn.Module):
    def __init__(self, img_resolution, img_channels, img_layers):
        super().__init__()
        self.img_resolution, self.img_channels, self.img_layers = img_resolution, img_channels, img_layers

    def forward(self, x):
        b, l, c, h, w = x.shape
        return self.cnn(x.reshape(b, l * c, h, w))


class Renderer(_RendererBase):
    """conv-relu, 3 x shared(conv-relu), conv, sigmoid: output in [0,1]."""
    _head = nn.Sigmoid

    def __init__(self, img_resolution, img_channels, img_layers, nf=64):
        super().__init__(img_resolution, img_channels, img_layers)
        self.block = _shared_block(nf)
        self.cnn = nn.Sequential(
            nn.Conv2d(img_channels * img_layers, nf, kernel_size=3, padding=1), nn.ReLU(True),
            self.block, self.block, self.block,
            nn.Conv2d(nf, img_channels, kernel_size=3, padding=1), self._head())


class RendererTanh(Renderer):
    """Same body with a tanh head: input and output in [-1,1]."""
    _head = nn.Tanh


class RendererSubPixelConv(_RendererBase):
    """PixelShuffle(6) front end; hard-wired upstream to 9 RGBA layers (36 = 6*6 channels)."""

    def __init__(self, img_resolution, img_channels, img_layers, nf1=8, nf2=64):
        super().__init__(img_resolution, img_channels, img_layers)
        r = math.isqrt(img_channels * img_layers)
        assert r ** 2 == img_channels * img_layers and r == 6, 'only 9 layers of RGBA are supported (as upstream)'
        self.block = _shared_block(nf2)
        self.cnn = nn.Sequential(
            nn.PixelShuffle(r),
            nn.Conv2d(1, nf1, kernel_size=3, stride=2, padding=1), nn.ReLU(True),
            nn.Conv2d(nf1, nf2, kernel_size=3, stride=3), nn.ReLU(True),
            self.block, self.block,
            nn.Conv2d(nf2, 4, kernel_size=3, padding=1), nn.Tanh())
