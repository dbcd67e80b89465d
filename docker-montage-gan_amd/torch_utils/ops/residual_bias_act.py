"""The tail of the discriminator's residual block as one pass over the block's output each way.

The reference's DiscriminatorBlock (training/networks.py:596-600, architecture 'resnet') ends with

    y = self.skip(x, gain=np.sqrt(0.5))         # ... conv ..., then bias_act(y_raw, None, act='linear', gain=sqrt(0.5))
    x = self.conv0(x)
    x = self.conv1(x, gain=np.sqrt(0.5))        # ... conv ..., then bias_act(x_raw, b, act, gain, clamp)
    x = y.add_(x)

``residual_bias_act(x_raw, b, y_raw, ...)`` is the last three element-wise operations -- the two bias_act calls and the add -- on the raw
outputs of the two convolutions: seven passes over the tensor forward (2 + 2 + 3) become three, and the backward pass (two bias_act gradients
and a bias reduction) one kernel that recomputes the activation's output from x_raw and b instead of reading a saved copy
(csrc/residual.hip).  Same values: every intermediate is rounded to x.dtype where the three launches round it.  CPU tensors, float32
blocks and ``create_graph=True`` (R1) run the composition out of this repo's differentiable ops.
"""
import os

import torch

from .. import native
from . import bias_act as _ba

enabled = os.environ.get('MG_FUSED_RESIDUAL', '1') != '0'        # module switch (tests compare both ways)


def _composition(x_raw, b, y_raw, act, alpha, gain, clamp, r_gain):
    y = _ba.bias_act(y_raw, None, act='linear', gain=r_gain)
    x = _ba.bias_act(x_raw, b, act=act, alpha=alpha, gain=gain, clamp=clamp)
    return y + x


def _nhwc(t):
    return t.ndim == 4 and t.is_contiguous(memory_format=torch.channels_last)


def applicable(x_raw, y_raw, act):
    return (enabled and act in ('linear', 'lrelu') and x_raw.is_cuda and x_raw.dtype in (torch.float16, torch.bfloat16)
            and y_raw.dtype == x_raw.dtype and x_raw.shape == y_raw.shape and _nhwc(x_raw) and _nhwc(y_raw)
            and x_raw.shape[1] in (64, 128, 256, 512) and x_raw.numel() > 0)


_cache = dict()


def _fused(act, alpha, gain, clamp, r_gain):
    key = (act, alpha, gain, clamp, r_gain)
    if key in _cache:
        return _cache[key]
    idx = _ba.activation_funcs[act].cuda_idx
    assert idx in (1, 3), 'linear or lrelu'

    class ResidualBiasAct(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x_raw, b, y_raw):
            n, c, h, w = x_raw.shape
            bb = b.detach().to(x_raw.dtype).contiguous() if b is not None else None
            out = torch.empty_like(x_raw)                   # keeps the channels-last layout
            with native.guard(x_raw.device):
                native.check(native.lib().mg_residual_bias_act_fwd(native.ptr(x_raw), native.ptr(bb), native.ptr(y_raw), native.ptr(out),
                                                                   native.dtype_code(x_raw.dtype), idx, alpha, gain, clamp, r_gain,
                                                                   n * h * w, c, native.stream_ptr(x_raw.device)))
            ctx.save_for_backward(x_raw, b if b is not None else _ba._null_tensor)
            ctx.has_bias = b is not None
            return out

        @staticmethod
        def backward(ctx, dout):
            x_raw, b = ctx.saved_tensors
            b = b if ctx.has_bias else None
            need = ctx.needs_input_grad            # x_raw, b, y_raw
            n, c, h, w = x_raw.shape
            if torch.is_grad_enabled():
                # create_graph=True (R1 differentiates the discriminator twice): explicit differentiable ops, no nested autograd.grad --
                # asking the engine for gradients w.r.t. x_raw from inside this backward made it walk the whole discriminator graph
                # below x_raw on every call (+0.43 s per regularisation iteration at batch 32).  bias_act's own first-order gradient op
                # is differentiable in dout (its dependence on x vanishes for the piecewise-linear activations used here).
                dout = dout.contiguous(memory_format=torch.channels_last)            # (differentiable; the plugin wants dy laid out like y)
                gy = _ba.bias_act(dout, None, act='linear', gain=r_gain) if need[2] else None
                gx = gb = None
                if need[0] or (need[1] and b is not None):
                    with torch.no_grad():
                        a = _ba.bias_act(x_raw, b, act=act, alpha=alpha, gain=gain, clamp=None if clamp < 0 else clamp)
                    grad_fn = _ba._bias_act_cuda(dim=1, act=act, alpha=alpha, gain=gain, clamp=(None if clamp < 0 else clamp)).Grad
                    ident = (act == 'linear' and gain == 1 and clamp < 0)
                    gx = dout if ident else grad_fn.apply(dout, _ba._null_tensor, _ba._null_tensor, a)
                    if need[1] and b is not None:
                        gb = gx.sum(dim=[0, 2, 3]).to(b.dtype)
                    if not need[0]:
                        gx = None
                return gx, gb, gy
            dout = dout.contiguous(memory_format=torch.channels_last)
            bb = b.to(x_raw.dtype).contiguous() if b is not None else None
            want_b = ctx.has_bias and need[1]
            dx = torch.empty_like(x_raw)
            dr = torch.empty_like(x_raw) if need[2] else None
            gb32 = torch.zeros([c], dtype=torch.float32, device=dout.device) if want_b else None
            with native.guard(dout.device):
                native.check(native.lib().mg_residual_bias_act_bwd(native.ptr(dout), native.ptr(x_raw), native.ptr(bb), native.ptr(dx), native.ptr(dr),
                                                                   native.ptr(gb32), native.dtype_code(dout.dtype), idx, alpha, gain, clamp, r_gain,
                                                                   n * h * w, c, native.stream_ptr(dout.device)))
            return (dx if need[0] else None), (gb32.to(b.dtype) if want_b else None), dr

    _cache[key] = ResidualBiasAct
    return ResidualBiasAct


def residual_bias_act(x_raw, b, y_raw, act='lrelu', alpha=None, gain=None, clamp=None, r_gain=1.0):
    """T(T(bias_act(x_raw, b, act, alpha, gain, clamp)) + T(bias_act(y_raw, None, 'linear', gain=r_gain)))."""
    spec = _ba.activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    if applicable(x_raw, y_raw, act):
        _ba._init()
        return _fused(act, alpha, gain, float(-1 if clamp is None else clamp), float(r_gain)).apply(x_raw, b, y_raw)
    return _composition(x_raw, b, y_raw, act, alpha, gain, clamp, r_gain)
