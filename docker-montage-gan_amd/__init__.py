"""MI355X-native hot path of MontageGAN (uchidalab/docker-montage-gan).

The directory mirrors the reference's ``montage_gan/`` source root for the one
path this repo rebuilds: ``torch_utils/ops`` (the plugin API), the callers that
sit directly on it, and ``csrc/`` -- the hand-written HIP kernels for gfx950
behind the C ABI of ``include/montage_hip.h``.

Import with ``importlib.import_module('docker-montage-gan_amd')`` (the hyphen in
the directory name rules out a plain ``import``); everything inside uses
relative imports, so ``torch_utils/`` can also be dropped over the reference's
``montage_gan/torch_utils`` as is.
"""
from .torch_utils import native  # noqa: F401
