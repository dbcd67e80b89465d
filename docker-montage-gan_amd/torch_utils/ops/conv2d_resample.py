"""2-D convolution with optional FIR up/down-sampling.

Same interface as the reference (montage_gan/torch_utils/ops/conv2d_resample.py:62):
``conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True,
flip_filter=False)``.  It does no arithmetic of its own: it picks how to combine the
dense contraction (``conv2d_gradfix`` -> MIOpen, the MFMA work) with the HBM-bound FIR
resampler (``upfirdn2d`` -> csrc/upfirdn2d.hip) so that the contraction always runs at
the lower of the two resolutions:

    1x1 conv + down   : resample first, then convolve
    1x1 conv + up     : convolve first, then resample
    kxk conv + down   : blur at full resolution, then a stride-`down` convolution
    kxk conv + up     : stride-`up` transposed convolution, then blur (gain up^2)
    no resampling     : plain convolution when the padding is symmetric and >= 0
    anything else     : upfirdn2d(up) -> convolution -> upfirdn2d(down)
"""
import torch

from .. import misc
from . import conv2d_gradfix
from . import upfirdn2d
from .upfirdn2d import _get_filter_size
from .upfirdn2d import _parse_padding


nhwc_1x1_workaround = True      # see _conv2d_wrapper


def _get_weight_shape(w):
    shape = [int(sz) for sz in w.shape]
    misc.assert_shape(w, shape)
    return shape


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """conv2d / conv_transpose2d; flip_weight=False means true convolution (flipped taps)."""
    oc, icg, kh, kw = _get_weight_shape(w)
    if not flip_weight and (kh > 1 or kw > 1):
        w = w.flip([2, 3])
    # channels_last 1x1 convolutions with few channels: the reference routes them around a cuDNN 8.0.5 pitfall
    # (:40-52) as a batched matmul / an NCHW convolution.  `nhwc_1x1_workaround` keeps that routing (default);
    # switched off, the convolution library gets the NHWC tensor directly.
    if nhwc_1x1_workaround and kw == 1 and kh == 1 and stride == 1 and padding in [0, [0, 0], (0, 0)] and not transpose:
        if x.stride()[1] == 1 and min(oc, icg) < 64:
            if oc <= 4 and groups == 1:
                n, _, h, ww = x.shape
                y = w.reshape(oc, icg) @ x.reshape(n, icg, h * ww)
                y = y.reshape(n, oc, h, ww)
            else:
                y = conv2d_gradfix.conv2d(x.contiguous(), w.contiguous(), groups=groups)
            return y.to(memory_format=torch.channels_last)
    op = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    return op(x, w, stride=stride, padding=padding, groups=groups)


@misc.profiled_function
def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False, tail=None):
    """x [N, C_in, H, W]; w [C_out, C_in/groups, kh, kw]; f from ``upfirdn2d.setup_filter`` or None;
    padding is w.r.t. the up-sampled image and applied once, before everything else.
    ``tail`` (additive, not in the reference's signature): a callable ``tail(x, f, padding, gain, flip_filter)`` that replaces the
    blur ending the up-sampling path (up > 1, down == 1) -- SynthesisLayer passes the blur fused with its epilogue."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1
    assert isinstance(groups, int) and groups >= 1
    oc, icg, kh, kw = _get_weight_shape(w)
    fw, fh = _get_filter_size(f)
    px0, px1, py0, py1 = _parse_padding(padding)

    # the resampling filters bring their own padding so that sizes come out as in*up/down
    if up > 1:
        px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    if down > 1:
        px0 += (fw - down + 1) // 2; px1 += (fw - down) // 2
        py0 += (fh - down + 1) // 2; py1 += (fh - down) // 2

    pointwise = (kw == 1 and kh == 1)
    if pointwise and down > 1 and up == 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)

    if pointwise and up > 1 and down == 1:
        x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
        return upfirdn2d.upfirdn2d(x=x, f=f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)

    if down > 1 and up == 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0, px1, py0, py1], flip_filter=flip_filter)
        return _conv2d_wrapper(x=x, w=w, stride=down, groups=groups, flip_weight=flip_weight)

    if up > 1:
        # transposed convolution wants [C_in, C_out/groups, kh, kw]
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, oc // groups, icg, kh, kw).transpose(1, 2).reshape(groups * icg, oc // groups, kh, kw)
        px0 -= kw - 1; px1 -= kw - up
        py0 -= kh - 1; py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)      # part of the crop the transposed conv can do itself
        pyt = max(min(-py0, -py1), 0)
        x = _conv2d_wrapper(x=x, w=wt, stride=up, padding=[pyt, pxt], groups=groups, transpose=True,
                            flip_weight=(not flip_weight))
        if tail is not None and down == 1:
            return tail(x, f, [px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], up ** 2, flip_filter)
        x = upfirdn2d.upfirdn2d(x=x, f=f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2,
                                flip_filter=flip_filter)
        if down > 1:
            x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
        return x

    if up == 1 and down == 1 and px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:
        return _conv2d_wrapper(x=x, w=w, padding=[py0, px0], groups=groups, flip_weight=flip_weight)

    x = upfirdn2d.upfirdn2d(x=x, f=(f if up > 1 else None), up=up, padding=[px0, px1, py0, py1], gain=up ** 2,
                            flip_filter=flip_filter)
    x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight)
    if down > 1:
        x = upfirdn2d.upfirdn2d(x=x, f=f, down=down, flip_filter=flip_filter)
    return x
