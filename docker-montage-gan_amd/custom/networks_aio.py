"""The aio network variants (reference custom/networks_aio.py).  Upstream this file is a copy of
training/networks.py with ``conv_config_index`` threaded through and ``GlobalMappingNetwork`` added;
here both live in one implementation and this module re-exports it under the reference's names."""
from ..training.networks import (Conv2dLayer, Discriminator, DiscriminatorBlock, DiscriminatorEpilogue,  # noqa: F401
                                 FullyConnectedLayer, Generator, GlobalMappingNetwork, MappingNetwork,
                                 MinibatchStdLayer, SynthesisBlock, SynthesisLayer, SynthesisNetwork, ToRGBLayer,
                                 modulated_conv2d, normalize_2nd_moment)
