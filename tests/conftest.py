"""pytest configuration: markers, import path for the product package and the oracle."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_pkg():
    """The product package; its directory name has a hyphen, so importlib is the way in."""
    return importlib.import_module('docker-montage-gan_amd')


def load_golden(name):
    with open(os.path.join(GOLDEN, name + '.json')) as fh:
        manifest = json.load(fh)
    arrays = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return manifest['cases'], arrays


@pytest.fixture(scope='session')
def pkg():
    return load_pkg()


@pytest.fixture(scope='session')
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
