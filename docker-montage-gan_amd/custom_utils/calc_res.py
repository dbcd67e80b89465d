"""Resolution arithmetic for non-square / non-power-of-two layer sizes (reference
custom_utils/calc_res.py:8-41): pure integer math, restated."""
import numpy as np


def calc_res(shape):
    """Power-of-two "resolution" label for an [H, W] image; doubled when the image is not exactly
    that power of two (or is >= 2:1) and both sides still divide evenly."""
    base = min(2 ** int(np.log2(shape[0])), 2 ** int(np.log2(shape[1])))
    min_res = min(shape[0], shape[1])
    if min_res != base or max(*shape) / min(*shape) >= 2:
        nxt = base * 2
        divides = all((s * 2 ** (2 - int(np.log2(nxt)))) % 1 == 0 for s in shape)
        if np.log2(base) < 10 and divides:
            base = nxt
    return base


def calc_init_res(shape, resolution=None, conv_config_index=2):
    """-> (init_res [h0, w0], resolution, log2(resolution)): the coarsest feature-map size such that
    init_res * 2**(res_log2 - conv_config_index) == image size."""
    if len(shape) == 1:
        shape = [shape[0], shape[0], 1]
    elif len(shape) == 2:
        shape = [*shape, 1]
    size = shape[:2] if shape[2] < min(*shape[:2]) else shape[1:]
    if resolution is None:
        resolution = calc_res(size)
    res_log2 = int(np.log2(resolution))
    return [int(s * 2 ** (conv_config_index - res_log2)) for s in size], resolution, res_log2
