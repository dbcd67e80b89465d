"""State_dict compatibility: the renderer checkpoints the reference ships load STRICTLY into this package's
RendererTanh / RendererSubPixelConv and reproduce the reference's outputs (tests/golden/renderer.*, generated from the
reference's own diff_rendering/networks.py).  The checkpoint files live only in the reference tree: where it is absent (the
GPU box) the key names and shapes are still checked against the committed manifest."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_nets_golden import T, mod

REF_ROOT = '/root/reference'


@pytest.mark.parametrize('i', [0, 1])
def test_shipped_renderer_checkpoints_load_strictly(i):
    RN = mod('diff_rendering.networks')
    cases, arr = load_golden('renderer')
    c = cases[i]
    net = getattr(RN, c['cls'])(img_resolution=256, img_channels=4, img_layers=9)
    own = net.state_dict()
    assert list(own.keys()) == c['keys']                                    # incl. the block.* / cnn.{2,3,4}.* aliases (App. B-4)
    assert [list(v.shape) for v in own.values()] == c['shapes']
    path = os.path.join(REF_ROOT, c['path'])
    if not os.path.exists(path):
        pytest.skip('reference tree absent: manifest check only')
    saved = torch.load(path, map_location='cpu', weights_only=True)        # never unpickle code from a shipped file
    assert int(saved['global_step']) == c['global_step']
    missing, unexpected = net.load_state_dict(saved['renderer'], strict=True)
    assert not missing and not unexpected
    np.testing.assert_allclose([float(v.double().sum()) for v in net.state_dict().values()], c['sums'], rtol=1e-12)
    net.eval()
    x = T(arr['x16']).repeat_interleave(4, 3).repeat_interleave(4, 4).contiguous()
    with torch.no_grad():
        y = net(x)
    np.testing.assert_allclose(y.numpy(), arr[f'r{i}_y'], rtol=1e-5, atol=1e-6)
