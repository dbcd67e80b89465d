"""Pad -> zero-upsample -> FIR -> decimate for batches of 2-D images.

Public interface identical to the reference (montage_gan/torch_utils/ops/upfirdn2d.py):
``setup_filter`` (:83), ``upfirdn2d`` (:132), ``filter2d`` (:290), ``upsample2d`` (:327),
``downsample2d`` (:367), with ``impl='cuda'|'ref'``.

* ``impl='cuda'`` on a GPU tensor runs the hand-written HIP kernels of
  ``csrc/upfirdn2d.hip`` through ``custom_ops.get_plugin('upfirdn2d_plugin')``.  If the
  native library cannot be loaded this RAISES -- the reference's warn-and-fall-back
  (:39-41) is exactly the silent slow path this build must not have.
* ``impl='ref'`` (or a CPU tensor) is the aten composition the API defines as its
  reference semantics; it is what the CPU configuration of BASELINE.json runs.

Gradients of any order: the backward of an upfirdn2d is an upfirdn2d with up and down
swapped, the filter flipped and the padding recomputed (:263-278).
"""
import numpy as np
import torch

from .. import custom_ops
from .. import native
from .. import misc
from . import conv2d_gradfix

_plugin = None


def _init():
    """Load the native plugin.  Returns True or raises; never reports a silent failure."""
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin('upfirdn2d_plugin', sources=['upfirdn2d.hip'])
    return True


def _parse_scaling(scaling):
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    assert isinstance(scaling, (list, tuple)) and len(scaling) == 2
    assert all(isinstance(v, int) and v >= 1 for v in scaling)
    return scaling[0], scaling[1]          # (x, y)


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple)) and len(padding) in (2, 4)
    assert all(isinstance(v, int) for v in padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    return tuple(padding)                  # (x0, x1, y0, y1)


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    fw, fh = int(f.shape[-1]), int(f.shape[0])
    assert fw >= 1 and fh >= 1
    return fw, fh                          # (width, height)


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Float32 FIR filter for ``upfirdn2d()``: [fh, fw] (non-separable) or [taps] (separable).
    A 1-D input with fewer than 8 taps becomes its outer product unless ``separable`` says
    otherwise; ``gain`` is split as gain**(ndim/2) per axis (reference :83-127)."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f.reshape(1)
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """x [N,C,H,W] (float16/bfloat16/float32/float64); f float32 [fh,fw], [taps] or None;
    up/down int or [x, y]; padding int, [x, y] or [x0, x1, y0, y1] w.r.t. the up-sampled
    image, negative = crop; flip_filter False = convolution.  Returns [N,C,H',W']."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda':
        _init()
        return _upfirdn2d_cuda(up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain).apply(x, f)
    return _upfirdn2d_ref(x, f, up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)


@misc.profiled_function
def _upfirdn2d_ref(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """aten composition with the API's reference semantics (reference :181-221)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    assert f.dtype == torch.float32 and not f.requires_grad
    n, c, ih, iw = x.shape
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)

    # zero insertion: sample (iy, ix) lands on (iy*upy, ix*upx)
    if upx > 1 or upy > 1:
        z = x.new_zeros([n, c, ih * upy, iw * upx])
        z[:, :, ::upy, ::upx] = x
        x = z
    # constant pad; negative amounts crop
    if px0 or px1 or py0 or py1:
        x = torch.nn.functional.pad(x, [px0, px1, py0, py1])

    # depthwise correlation with the (flipped unless flip_filter) filter, gain folded into it
    k = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if not flip_filter:
        k = k.flip(list(range(k.ndim)))
    if k.ndim == 2:
        x = conv2d_gradfix.conv2d(x, k.expand(c, 1, *k.shape).contiguous(), groups=c)
    else:
        x = conv2d_gradfix.conv2d(x, k.reshape(1, 1, 1, -1).expand(c, 1, 1, -1).contiguous(), groups=c)
        x = conv2d_gradfix.conv2d(x, k.reshape(1, 1, -1, 1).expand(c, 1, -1, 1).contiguous(), groups=c)
    # decimation
    return x[:, :, ::downy, ::downx]


_upfirdn2d_cuda_cache = dict()


class _NoCtx:
    """Stand-in for the autograd context when a Function's forward body is run outside autograd."""
    x_shape = None

    @staticmethod
    def save_for_backward(*tensors):
        pass


sep12_enabled = True        # the one-pass 12-tap x2 resampler (csrc/upfirdn2d_sep.hip); False: always two mg_upfirdn2d passes


def _sep12_applicable(x, f, upx, upy, downx, downy):
    """A 12-tap 1-D filter by a factor of 2 in both axes on dense float32 NCHW: the ADA geometric stage (augment.py:317,331) and
    its adjoint.  Everything else takes the two-pass form."""
    return (sep12_enabled and (upx, upy, downx, downy) in ((2, 2, 1, 1), (1, 1, 2, 2)) and f.numel() == 12 and f.dtype == torch.float32
            and x.dtype == torch.float32 and x.is_cuda and f.is_cuda and x.is_contiguous() and 0 < x.shape[0] * x.shape[1] <= 65535
            and x.shape[2] >= 1 and x.shape[3] >= 1)


def _sep12(x, f, up, px0, px1, py0, py1, flip_filter, gain_axis):
    from .. import native
    n, c, ih, iw = x.shape
    if up:
        oh, ow = ih * 2 + py0 + py1 - 11, iw * 2 + px0 + px1 - 11
    else:
        oh, ow = (ih + py0 + py1 - 10) // 2, (iw + px0 + px1 - 10) // 2
    if oh < 1 or ow < 1:    # same message as the plugin (upfirdn2d.cpp:34)
        raise RuntimeError('output must be at least 1x1')
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device)
    f = f.contiguous()
    with native.guard(x.device):
        native.check(native.lib().mg_upfirdn2d_sep12(native.ptr(x), native.ptr(f), native.ptr(y), n * c, ih, iw, oh, ow, int(up), px0, py0,
                                                     int(flip_filter), gain_axis, native.stream_ptr(x.device)))
    return y


class _ReflectPadUpsample12(torch.autograd.Function):
    """upsample2d(pad(x, margins, mode='reflect'), f12, up=2) in one launch; the gradient is the composition's: the adjoint of
    the upsampling (the one-pass kernel again, down by 2) followed by aten's reflection_pad2d_backward -- both differentiable
    again, which R1 needs (it differentiates the discriminator input twice through the augmentation pipe)."""

    @staticmethod
    def forward(ctx, x, f, mx0, mx1, my0, my1, window=None):
        from .. import native
        n, c, sh, sw = x.shape
        y = torch.empty([n, c, 2 * (sh + my0 + my1), 2 * (sw + mx0 + mx1)], dtype=x.dtype, device=x.device)
        with native.guard(x.device):
            # window: device float32[4] (mx0, my0, mx1, my1) of the margins the consumer really needs -- tiles outside stay unwritten
            native.check(native.lib().mg_reflect_pad_upsample12_window(native.ptr(x), native.ptr(f), native.ptr(y), n * c, sh, sw, mx0, mx1, my0, my1,
                                                                       2.0, native.ptr(window), native.stream_ptr(x.device)))
        ctx.save_for_backward(x, f)
        ctx.margins = (mx0, mx1, my0, my1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, f = ctx.saved_tensors
        dx = None
        if ctx.needs_input_grad[0]:
            grad_cls = _upfirdn2d_cuda(up=1, down=2, padding=[5, 5, 5, 5], flip_filter=True, gain=4)     # Upfirdn2dCuda.backward's arithmetic for up=2, pad (6, 5)
            gp = grad_cls.apply(dy, f) if torch.is_grad_enabled() else grad_cls.forward(_NoCtx, dy, f)
            dx = torch.ops.aten.reflection_pad2d_backward(gp, x, list(ctx.margins))
        return dx, None, None, None, None, None, None


def reflect_pad_upsample2d(x, f, margins, window=None):
    """``upsample2d(torch.nn.functional.pad(x, margins, mode='reflect'), f, up=2)`` (training/augment.py:315-317); margins =
    [left, right, top, bottom].  ``window`` (device float32[4]: left, top, right, bottom margins as computed on the device, each <= the
    padding): only the part of the result a consumer with those margins reads is computed, the rest is left uninitialised -- for callers that
    must pad by a host-known bound (a recorded HIP graph cannot read the data-dependent margin back).  One launch for the 12-tap filter on dense float32 device tensors, the two ops otherwise."""
    mx0, mx1, my0, my1 = (int(m) for m in margins)
    if (sep12_enabled and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and f.ndim == 1 and f.numel() == 12 and f.is_cuda
            and f.dtype == torch.float32 and 0 < x.shape[0] * x.shape[1] <= 65535 and min(mx0, mx1, my0, my1) >= 0
            and max(mx0, mx1) < x.shape[3] and max(my0, my1) < x.shape[2] and _init()):
        if window is not None:
            window = window.detach().to(torch.float32).contiguous()
            assert window.numel() == 4 and window.device == x.device
        return _ReflectPadUpsample12.apply(x, f.contiguous(), mx0, mx1, my0, my1, window)
    x = torch.nn.functional.pad(input=x, pad=[mx0, mx1, my0, my1], mode='reflect')
    return upsample2d(x=x, f=f, up=2)


def _upfirdn2d_cuda(up=1, down=1, padding=0, flip_filter=False, gain=1):
    """autograd.Function class for one scalar configuration (cached, reference :226-285)."""
    upx, upy = _parse_scaling(up)
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    key = (upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain)
    cls = _upfirdn2d_cuda_cache.get(key)
    if cls is not None:
        return cls

    class Upfirdn2dCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, f):
            assert isinstance(x, torch.Tensor) and x.ndim == 4
            if f is None:
                f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
            if f.ndim == 2:
                y = _plugin.upfirdn2d(x, f, upx, upy, downx, downy, px0, px1, py0, py1, flip_filter, gain)
            elif _sep12_applicable(x, f, upx, upy, downx, downy):
                y = _sep12(x, f, upx == 2, px0, px1, py0, py1, flip_filter, float(np.sqrt(gain)))
            else:   # separable: a row pass then a column pass, sqrt(gain) each (:253-257)
                g = float(np.sqrt(gain))
                y = _plugin.upfirdn2d(x, f.unsqueeze(0), upx, 1, downx, 1, px0, px1, 0, 0, flip_filter, g)
                y = _plugin.upfirdn2d(y, f.unsqueeze(1), 1, upy, 1, downy, 0, 0, py0, py1, flip_filter, g)
            ctx.save_for_backward(f)
            ctx.x_shape = x.shape
            return y

        @staticmethod
        def backward(ctx, dy):
            f, = ctx.saved_tensors
            _, _, ih, iw = ctx.x_shape
            _, _, oh, ow = dy.shape
            fw, fh = _get_filter_size(f)
            dx = None
            if ctx.needs_input_grad[0]:
                pad = [fw - px0 - 1, iw * upx - ow * downx + px0 - upx + 1,
                       fh - py0 - 1, ih * upy - oh * downy + py0 - upy + 1]
                grad_cls = _upfirdn2d_cuda(up=[downx, downy], down=[upx, upy], padding=pad, flip_filter=(not flip_filter), gain=gain)
                # plain backward: run the op's forward body directly; only create_graph=True needs another autograd node
                dx = grad_cls.apply(dy, f) if torch.is_grad_enabled() else grad_cls.forward(_NoCtx, dy, f)
            assert not ctx.needs_input_grad[1], 'the FIR filter is a constant'
            return dx, None

    _upfirdn2d_cuda_cache[key] = Upfirdn2dCuda
    return Upfirdn2dCuda


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """FIR filtering that keeps the image size (user padding on top; reference :290-322)."""
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Output size = input size * up; the gain is multiplied by upx*upy so that DC is kept
    (reference :327-362)."""
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Output size = input size / down (reference :367-402)."""
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
