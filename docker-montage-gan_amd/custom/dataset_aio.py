"""Datasets for the aio loop: the interface ``training_loop_aio`` reads from its dataset object
(reference custom/dataset_aio.py:143-254; consumers custom/training_loop_aio.py:251-254,262-263,282):

    name, layer_names, num_layers, num_channels, resolution, res_log2, init_res,
    init_res_layer(n), resolution_layer(n), target_res_layer(n), __len__, __getitem__ -> float [L,C,H,W] in [0,1]

``SyntheticLayerDataset`` generates RGBA layer stacks with blob-shaped alpha masks on the fly
(there is no dataset in the offline image) -- the benchmark's input.
``DatasetAIO`` reads the reference's on-disk format: one directory per sample holding one RGBA PNG
per layer, sorted by file name.
"""
import os

import numpy as np
import torch

from ..custom_utils.calc_res import calc_init_res, calc_res


class _LayerDatasetBase(torch.utils.data.Dataset):
    name = 'layers'

    def _init_meta(self, num_layers, num_channels, height, width, layer_names=None, layer_sizes=None, conv_config_index=2):
        self.conv_config_index = conv_config_index
        self.layer_names = list(layer_names) if layer_names is not None else [f'layer{i}' for i in range(num_layers)]
        assert len(self.layer_names) == num_layers
        sizes = layer_sizes if layer_sizes is not None else [[height, width]] * num_layers
        self.layer_stats = {n: dict(target_height=int(s[0]), target_width=int(s[1])) for n, s in zip(self.layer_names, sizes)}
        self._raw_shape = [len(self), num_layers, num_channels, height, width]

    layer_shape = property(lambda self: self._raw_shape[1:])
    image_shape = property(lambda self: self._raw_shape[2:])
    num_layers = property(lambda self: self._raw_shape[1])
    num_channels = property(lambda self: self._raw_shape[2])
    resolution = property(lambda self: calc_res(self.image_shape[1:]))
    res_log2 = property(lambda self: int(np.ceil(np.log2(self.resolution))))
    init_res = property(lambda self: calc_init_res(self.image_shape[1:], conv_config_index=self.conv_config_index)[0])

    def _layer_size(self, layer_name):
        st = self.layer_stats[layer_name]
        return [st['target_height'], st['target_width']]

    def res_log2_layer(self, layer_name):
        return calc_init_res(self._layer_size(layer_name), conv_config_index=self.conv_config_index)[2]

    def init_res_layer(self, layer_name):
        return calc_init_res(self._layer_size(layer_name), conv_config_index=self.conv_config_index)[0]

    def resolution_layer(self, layer_name):
        return calc_res(self._layer_size(layer_name))

    def target_res_layer(self, layer_name):
        return self._layer_size(layer_name)


class SyntheticLayerDataset(_LayerDatasetBase):
    """Deterministic synthetic RGBA layer stacks: sample i is a function of (seed, i) only."""
    name = 'synthetic-layers'

    def __init__(self, path=None, num_samples=1024, num_layers=8, num_channels=4, resolution=256, layer_sizes=None,
                 conv_config_index=2, seed=0, xflip=False, **_ignored):
        self._n = int(num_samples)
        self._seed = seed
        self._res = resolution
        self._init_meta(num_layers, num_channels, resolution, resolution, layer_sizes=layer_sizes,
                        conv_config_index=conv_config_index)

    def __len__(self):
        return self._n

    def __getitem__(self, index):
        return synthetic_layers(1, self.num_layers, self._res, seed=self._seed * 1000003 + int(index))[0]


def synthetic_layers(batch, num_layers, resolution, seed=0, device='cpu'):
    """[B,L,4,H,W] float32 in [0,1]: random colour fields behind blob-shaped alpha masks
    (alpha = lowpass(rand) > 0.5, smoothed at the rim), so that crop_to_content has real work."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    b, l, r = batch, num_layers, resolution
    coarse = max(r // 32, 2)
    field = torch.rand([b * l, 1, coarse, coarse], generator=g)
    field = torch.nn.functional.interpolate(field, size=(r, r), mode='bicubic', align_corners=False)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, r), torch.linspace(-1, 1, r), indexing='ij')
    window = (1 - (xx ** 2 + yy ** 2)).clamp(0, 1).reshape(1, 1, r, r)          # keep the blobs off the border
    alpha = ((field * window - 0.35) * 8).clamp(0, 1)
    rgb = torch.rand([b * l, 3, coarse, coarse], generator=g)
    rgb = torch.nn.functional.interpolate(rgb, size=(r, r), mode='bilinear', align_corners=False).clamp(0, 1)
    out = torch.cat([rgb, alpha], dim=1).reshape(b, l, 4, r, r)
    return out.to(device)


class DatasetAIO(_LayerDatasetBase):
    """<path>/<sample>/<layer>.png, RGBA, all the same size (reference :143-231).  Layer statistics
    (the per-layer target sizes) default to the full canvas; pass ``layer_sizes`` to crop."""

    def __init__(self, path, xflip=True, conv_config_index=2, layer_sizes=None, name=None, **_ignored):
        from PIL import Image
        self._path = path
        self._dirs = sorted(d for d in os.listdir(path) if os.path.isdir(os.path.join(path, d)))
        assert self._dirs, f'no samples under {path}'
        self.raw_idx = np.arange(len(self._dirs), dtype=np.int64)
        self.is_xflip = np.zeros(self.raw_idx.size, dtype=np.uint8)
        if xflip:
            self.raw_idx = np.tile(self.raw_idx, 2)
            self.is_xflip = np.concatenate([self.is_xflip, np.ones_like(self.is_xflip)])
        first = sorted(os.listdir(os.path.join(path, self._dirs[0])))
        probe = np.asarray(Image.open(os.path.join(path, self._dirs[0], first[0])).convert('RGBA'))
        self.name = name or os.path.basename(os.path.normpath(path))
        self._init_meta(len(first), 4, probe.shape[0], probe.shape[1], layer_names=[f[:-4] for f in first],
                        layer_sizes=layer_sizes, conv_config_index=conv_config_index)

    def __len__(self):
        return len(self.raw_idx)

    def __getitem__(self, index):
        from PIL import Image
        d = os.path.join(self._path, self._dirs[self.raw_idx[index]])
        layers = [torch.from_numpy(np.asarray(Image.open(os.path.join(d, f)).convert('RGBA'), dtype=np.uint8).copy())
                  for f in sorted(os.listdir(d))]
        out = torch.stack(layers).permute(0, 3, 1, 2).to(torch.float32) / 255.0
        return torch.flip(out, [-1]) if self.is_xflip[index] else out
