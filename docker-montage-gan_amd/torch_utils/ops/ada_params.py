"""Transform parameters of the ADA pipe in one native launch.

``AugmentPipe`` (reference training/augment.py:236-330, 380-420) builds, per call, a per-sample 3x3 geometric matrix and a
4x4 colour matrix out of ~13 random decisions, each a handful of tiny aten kernels on [N]-sized tensors plus a stack and
a batched 3x3 / 4x4 matmul: ~200 launches of 3-4 us before a single pixel moves, 27 times per training step.
``ada_params`` draws nothing itself: it takes two tables of pre-drawn numbers (uniform U [N,nu], normal Z [N,nz], consumed
column by column in the order the pipe's python code consumes its draws) and the device scalar ``p`` and returns the same
matrices from ONE kernel (csrc/augment.hip ``ada_params_kernel``, one lane per sample).  ``training/augment.py`` can run
its python formulation from the same tables (``AugmentPipe.params_from_tables``), which is what the parity test compares.
"""
import ctypes

import numpy as np
import torch

from .. import native

NU, NZ = 21, 7          # columns the full 'bgc' specification consumes (fewer are fine when transforms are disabled)
_cfg_cache = dict()


def _cfg(pipe, width, height, channels, left, right):
    key = (id(pipe), width, height, channels, None if left is None else left.tobytes(), None if right is None else right.tobytes())
    hit = _cfg_cache.get(key)
    if hit is None:
        v = [pipe.xflip, pipe.rotate90, pipe.xint, pipe.xint_max, pipe.scale, pipe.scale_std, pipe.rotate, pipe.rotate_max,
             pipe.aniso, pipe.aniso_std, pipe.xfrac, pipe.xfrac_std, pipe.brightness, pipe.brightness_std, pipe.contrast,
             pipe.contrast_std, pipe.lumaflip, pipe.hue, pipe.hue_max, pipe.saturation, pipe.saturation_std,
             float(width), float(height), float(pipe.Hz_geom.shape[0] // 4), float(channels), 0.0 if left is None else 1.0]
        v += list(np.asarray(left if left is not None else np.eye(3), dtype=np.float32).reshape(-1))
        v += list(np.asarray(right if right is not None else np.eye(3), dtype=np.float32).reshape(-1))
        hit = (ctypes.c_float * len(v))(*[float(x) for x in v])
        if len(_cfg_cache) < 1024:
            _cfg_cache[key] = hit
    return hit


def has_geometric(pipe):
    return any(v > 0 for v in (pipe.xflip, pipe.rotate90, pipe.xint, pipe.scale, pipe.rotate, pipe.aniso, pipe.xfrac))


def has_color(pipe, channels):
    return any(v > 0 for v in (pipe.brightness, pipe.contrast, pipe.lumaflip)) or (channels > 1 and (pipe.hue > 0 or pipe.saturation > 0))


def ada_params(pipe, U, Z, width, height, channels, left=None, right=None):
    """-> dict(G_inv [N,3,3], margin [4] float, theta [N,2,3] (only with left/right), C [N,4,4], A4 [N,4,4], t4 [N,4]);
    entries of a disabled stage are None.  U, Z: float32 CUDA tensors [N, >= columns needed]."""
    assert U.is_cuda and U.dtype == torch.float32 and Z.dtype == torch.float32 and U.is_contiguous() and Z.is_contiguous()
    n, dev = U.shape[0], U.device
    geom, color = has_geometric(pipe), has_color(pipe, channels)
    f32 = dict(dtype=torch.float32, device=dev)
    out = dict(G_inv=None, margin=None, theta=None, C=None, A4=None, t4=None)
    if geom:
        out['G_inv'], out['margin'] = torch.empty([n, 3, 3], **f32), torch.empty([4], **f32)
        if left is not None:
            out['theta'] = torch.empty([n, 2, 3], **f32)
    if color:
        out['C'], out['A4'], out['t4'] = torch.empty([n, 4, 4], **f32), torch.empty([n, 4, 4], **f32), torch.empty([n, 4], **f32)
    if not (geom or color) or n == 0:
        return out
    cfg = _cfg(pipe, width, height, channels, left, right)
    with native.guard(dev):
        native.check(native.lib().mg_ada_params(
            U.data_ptr(), Z.data_ptr(), pipe.p.data_ptr(), n, U.shape[1], Z.shape[1], cfg, len(cfg),
            native.ptr(out['G_inv']), native.ptr(out['margin']), native.ptr(out['theta']), native.ptr(out['C']),
            native.ptr(out['A4']), native.ptr(out['t4']), native.stream_ptr(dev)))
    return out
