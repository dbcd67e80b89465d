"""Image datasets with the interface ``training_loop`` reads (reference training/dataset.py:27-160):
``resolution``, ``num_channels``, ``image_shape``, ``label_dim``, ``label_shape``, ``has_labels``,
``get_label(idx)``, ``__getitem__ -> (uint8 [C,H,W], label)``.  On-disk formats (zip / folder / pyspng)
are I/O and out of scope; the benchmark and the tests use the synthetic set."""
import numpy as np
import torch


class SyntheticImageDataset(torch.utils.data.Dataset):
    """Deterministic random uint8 images; sample i depends on (seed, i) only."""
    name = 'synthetic-images'

    def __init__(self, path=None, resolution=256, num_channels=4, num_samples=1024, label_dim=0, seed=0, xflip=False,
                 use_labels=False, max_size=None, **_ignored):
        self._n, self._res, self._ch, self._label_dim, self._seed = int(num_samples), resolution, num_channels, label_dim, seed

    resolution = property(lambda self: self._res)
    num_channels = property(lambda self: self._ch)
    image_shape = property(lambda self: [self._ch, self._res, self._res])
    label_dim = property(lambda self: self._label_dim)
    label_shape = property(lambda self: [self._label_dim])
    has_labels = property(lambda self: self._label_dim > 0)

    def __len__(self):
        return self._n

    def get_label(self, idx):
        lab = np.zeros([self._label_dim], dtype=np.float32)
        if self._label_dim:
            lab[int(idx) % self._label_dim] = 1
        return lab

    def __getitem__(self, idx):
        rs = np.random.RandomState((self._seed * 1000003 + int(idx)) % (2 ** 31))
        coarse = rs.randint(0, 256, size=[self._ch, max(self._res // 16, 1), max(self._res // 16, 1)]).astype(np.uint8)
        img = np.kron(coarse, np.ones([1, 16, 16], dtype=np.uint8))[:, :self._res, :self._res]
        return img.copy(), self.get_label(idx)
