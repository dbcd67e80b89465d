"""filtered_lrelu: upsample-FIR -> bias + leaky ReLU (+clamp) -> downsample-FIR.

Named by the north star; the reference tree does not contain it (it is a StyleGAN3 op;
SURVEY D1), so its definition here is the composition of the reference's own ops:

    t = upfirdn2d(x + b[C], fu, up=up, padding=padding, gain=up**2, flip_filter=flip)
    u = bias_act(t, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    y = upfirdn2d(u, fd, down=down, flip_filter=flip)

``impl='cuda'`` keeps the up-sampled intermediate in LDS (csrc/fused.hip: filtered_lrelu_kernel).
Status: a FORWARD-ONLY inference helper.  Nothing on the training step calls it (StyleGAN2's layers have no
up-act-down sandwich), it is not part of the step-path kernel list or of the benchmark, and a tensor that requires grad takes the
composition (``impl='ref'``), which on GPU tensors itself runs the native upfirdn2d / bias_act kernels.
Parity: against that composition ("parity unpinned" upstream: there is no reference implementation to pin it to).
"""
import numpy as np
import torch

from .. import native
from . import bias_act as _ba
from . import upfirdn2d as _uf


def _as_2d(f):
    if f is None:
        return None
    return torch.outer(f, f) if f.ndim == 1 else f


def _composition(x, fu, fd, b, up, down, padding, gain, slope, clamp, flip_filter):
    if b is not None:
        x = x + b.to(x.dtype).reshape(1, -1, 1, 1)
    t = _uf.upfirdn2d(x, fu, up=up, padding=padding, gain=up ** 2, flip_filter=flip_filter)
    u = _ba.bias_act(t, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    return _uf.upfirdn2d(u, fd, down=down, flip_filter=flip_filter)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='cuda'):
    """x [N,C,H,W]; fu/fd float32 FIR filters ([taps] or [fh,fw]) or None; padding as in upfirdn2d
    (applies to the up-sampling stage)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert impl in ['ref', 'cuda']
    use_native = (impl == 'cuda' and x.device.type == 'cuda' and x.dtype != torch.float64 and up in (1, 2)
                  and down in (1, 2) and not (torch.is_grad_enabled() and (x.requires_grad or (b is not None and b.requires_grad))))
    if not use_native:
        return _composition(x, fu, fd, b, up, down, padding, gain, slope, clamp, flip_filter)
    fu2 = _as_2d(fu) if fu is not None else torch.ones([1, 1], dtype=torch.float32, device=x.device)
    fd2 = _as_2d(fd) if fd is not None else torch.ones([1, 1], dtype=torch.float32, device=x.device)
    fu2, fd2 = fu2.contiguous(), fd2.contiguous()
    px0, px1, py0, py1 = _uf._parse_padding(padding)
    n, c, ih, iw = x.shape
    tw = iw * up + px0 + px1 - (fu2.shape[1] - 1)
    th = ih * up + py0 + py1 - (fu2.shape[0] - 1)
    ow = (tw - fd2.shape[1] + down) // down
    oh = (th - fd2.shape[0] + down) // down
    assert ow >= 1 and oh >= 1
    x = x.contiguous()
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device)
    bb = b.to(x.dtype).contiguous() if b is not None else None
    with native.guard(x.device):
        native.check(native.lib().mg_filtered_lrelu(
            native.ptr(x), native.ptr(bb), native.ptr(fu2), native.ptr(fd2), native.ptr(y), native.dtype_code(x.dtype),
            n, c, ih, iw, oh, ow, fu2.shape[0], fu2.shape[1], fd2.shape[0], fd2.shape[1], up, down, px0, py0,
            float(slope), float(gain), float(-1 if clamp is None else clamp), 1 if flip_filter else 0,
            native.stream_ptr(x.device)))
    return y
