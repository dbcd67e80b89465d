"""The discriminator's FromRGB layer as one pass over the wide tensor each way.

What the reference executes for `y = self.fromrgb(img)` (training/networks.py:589-594; Conv2dLayer.forward :161-186 with
kernel_size = 1, up = down = 1):

    yc = conv2d_resample(img, weight.to(x.dtype), padding=0)          # 1x1, img_channels (<= 4) -> 64..512 channels
    y  = bias_act(yc, bias.to(x.dtype), act, gain, clamp)

``fromrgb(x, w, b, act, gain, clamp)`` returns that ``y``.  On channels-last float16 / bfloat16 GPU tensors it is ONE kernel
(csrc/fromrgb.hip ``mg_fromrgb_nhwc_fwd``: the C-channel tensor is written once, rounded where the two launches round it), and its
plain backward is one kernel that reads dy and the saved y once and leaves dx, dW and db (``mg_fromrgb_nhwc_bwd``) instead of the
bias_act gradient + the weight, bias and data gradients of the convolution.  Under ``create_graph=True`` (the R1 penalty
differentiates the discriminator twice w.r.t. the real images) the backward is the composition's own, built out of this repo's
differentiable ops; CPU tensors, float32 and other shapes run the composition.  An additive op: same values either way.
"""
import os

import torch

from .. import native
from . import bias_act as _ba
from . import conv2d_resample as _cr

enabled = os.environ.get('MG_FUSED_FROMRGB', '1') != '0'        # module switch (tests compare both ways)


def _composition(x, w, b, act, alpha, gain, clamp):
    yc = _cr.conv2d_resample(x=x, w=w, f=None, up=1, down=1, padding=0, flip_weight=True)
    return _ba.bias_act(yc, b, act=act, alpha=alpha, gain=gain, clamp=clamp)


def applicable(x, w):
    return (enabled and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and x.ndim == 4 and w.ndim == 4
            and w.shape[2] == w.shape[3] == 1 and x.shape[1] == w.shape[1] and 1 <= x.shape[1] <= 4
            and w.shape[0] in (64, 128, 256, 512) and x.numel() > 0 and w.dtype == x.dtype)


_cache = dict()


def _fused(act, alpha, gain, clamp):
    key = (act, alpha, gain, clamp)
    if key in _cache:
        return _cache[key]
    idx = _ba.activation_funcs[act].cuda_idx
    assert idx in (1, 3), 'linear or lrelu'

    class FromRGB(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, b):
            n, i, h, wd = x.shape
            c = w.shape[0]
            # pixel-major views of the operands: [N*H*W, I] and [C, I]
            xc = x.contiguous(memory_format=torch.channels_last) if i > 1 else x.contiguous()
            w2 = w.detach().reshape(c, i).contiguous()
            bb = b.detach().to(x.dtype).contiguous() if b is not None else None
            y = torch.empty([n, c, h, wd], dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            with native.guard(x.device):
                native.check(native.lib().mg_fromrgb_nhwc_fwd(native.ptr(xc), native.ptr(w2), native.ptr(bb), native.ptr(y),
                                                              native.dtype_code(x.dtype), idx, alpha, gain, clamp, n * h * wd, c, i,
                                                              native.stream_ptr(x.device)))
            # (x itself next to its pixel-major copy: a re-laid-out copy has no autograd identity, create_graph needs the input)
            ctx.save_for_backward(xc, w, b if b is not None else _ba._null_tensor, y, x)
            ctx.has_bias = b is not None
            return y

        @staticmethod
        def backward(ctx, dy):
            xc, w, b, y, x0 = ctx.saved_tensors
            b = b if ctx.has_bias else None
            need = ctx.needs_input_grad
            n, i, h, wd = xc.shape
            c = w.shape[0]
            if torch.is_grad_enabled():
                # create_graph=True (the R1 penalty differentiates the discriminator twice): the composition recomputed from the ORIGINAL
                # tensors and differentiated by autograd -- the results are functions of dy and of x / w / b themselves.  (x is the
                # real-image leaf there and w / b are requested as inputs, so the engine stops at them.)
                with torch.enable_grad():
                    yc = _composition(x0, w, b, act, alpha, gain, None if clamp < 0 else clamp)
                want = [need[0], need[1], need[2] and b is not None]
                ins = [t for t, nd in zip((x0, w, b), want) if nd]
                outs = iter(torch.autograd.grad([yc], ins, [dy], create_graph=True, allow_unused=True) if ins else [])
                return tuple((next(outs) if nd else None) for nd in want)
            dyc = dy.contiguous(memory_format=torch.channels_last)
            w2 = w.reshape(c, i).contiguous()
            want_b = ctx.has_bias and need[2]
            acc = torch.zeros([c * i + (c if want_b else 0)], dtype=torch.float32, device=dy.device)
            gw32, gb32 = acc[:c * i], (acc[c * i:] if want_b else None)
            out = torch.empty([c * i + (c if want_b else 0)], dtype=dy.dtype, device=dy.device)
            gw, gb = out[:c * i], (out[c * i:] if want_b else None)
            dx = torch.empty_like(xc) if need[0] else None
            with native.guard(dy.device):
                native.check(native.lib().mg_fromrgb_nhwc_bwd(native.ptr(xc), native.ptr(w2), native.ptr(y), native.ptr(dyc), native.ptr(dx),
                                                              native.ptr(gw32), native.ptr(gb32), native.ptr(gw), native.ptr(gb),
                                                              native.dtype_code(dy.dtype), idx, alpha, gain, clamp, n * h * wd, c, i,
                                                              native.stream_ptr(dy.device)))
            return dx, (gw.view(w.shape) if need[1] else None), (gb.to(b.dtype) if want_b else None)

    _cache[key] = FromRGB
    return FromRGB


def fromrgb(x, w, b=None, act='lrelu', alpha=None, gain=None, clamp=None):
    """x [N,I,H,W] (I <= 4); w [C,I,1,1] in x.dtype (gain folded in); b [C] or None -> [N,C,H,W] channels-last."""
    spec = _ba.activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    if act in ('linear', 'lrelu') and applicable(x, w):
        _ba._init()
        return _fused(act, alpha, gain, float(-1 if clamp is None else clamp)).apply(x, w, b)
    return _composition(x, w, b, act, alpha, gain, clamp)
