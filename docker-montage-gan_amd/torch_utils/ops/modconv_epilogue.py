"""Fused epilogue of the (non-fused-weights) modulated convolution.

The reference's SynthesisLayer finishes every convolution with two separate ops
(training/networks.py:86-88 then :311-313):

    x = fma.fma(x, dcoefs.reshape(N,-1,1,1), noise)            # read x, write x
    x = bias_act.bias_act(x, b, act=act, gain=gain, clamp=c)  # read x, write x

``modconv_epilogue`` is the same arithmetic -- including the rounding of the intermediate to
x.dtype -- in one pass over the activation (csrc/fused.hip).  It is an additive op: callers that
keep calling fma + bias_act get identical values, at three HBM round trips instead of one.
The backward returns gradients for x, dcoefs, noise (already scaled) and b.
"""
import torch

from .. import native
from . import bias_act as _ba
from . import fma as _fma


def _epilogue_ref(x, dcoefs, noise, b, act, alpha, gain, clamp):
    n, c = x.shape[:2]
    if dcoefs is not None and noise is not None:
        x = _fma.fma(x, dcoefs.to(x.dtype).reshape(n, c, 1, 1), noise.to(x.dtype))
    elif dcoefs is not None:
        x = x * dcoefs.to(x.dtype).reshape(n, c, 1, 1)
    elif noise is not None:
        x = x + noise.to(x.dtype)
    return _ba.bias_act(x, b, act=act, alpha=alpha, gain=gain, clamp=clamp, impl='ref')


_cache = dict()


def _epilogue_hip(act, alpha, gain, clamp):
    key = (act, alpha, gain, clamp)
    if key in _cache:
        return _cache[key]
    spec = _ba.activation_funcs[act]
    assert 'x' not in spec.ref, f'{act} needs its pre-activation input and cannot be fused'
    idx = spec.cuda_idx

    class ModconvEpilogue(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, dcoefs, noise, b):
            n, c, h, w = x.shape
            x = x.contiguous()
            d32 = dcoefs.detach().to(torch.float32).contiguous() if dcoefs is not None else None
            nz32 = noise.detach().to(torch.float32).contiguous() if noise is not None else None
            bb = b.detach().to(x.dtype).contiguous() if b is not None else None
            y = torch.empty_like(x)
            with torch.cuda.device(x.device):
                native.check(native.lib().mg_fma_bias_act(
                    native.ptr(x), native.ptr(d32), native.ptr(nz32), native.ptr(bb), native.ptr(y),
                    native.dtype_code(x.dtype), idx, alpha, gain, clamp, 1.0, n, c, h * w, native.stream_ptr(x.device)))
            ctx.save_for_backward(x, d32 if d32 is not None else torch.empty(0), y)
            ctx.has = (dcoefs is not None, noise is not None, b is not None)
            ctx.meta = (None if dcoefs is None else dcoefs.dtype, None if noise is None else (noise.dtype, noise.shape),
                        None if b is None else b.dtype)
            return y

        @staticmethod
        def backward(ctx, dy):
            x, d32, y = ctx.saved_tensors
            n, c, h, w = x.shape
            has_d, has_nz, has_b = ctx.has
            dy = dy.contiguous()
            need_pre = has_d and ctx.needs_input_grad[1] or has_nz and ctx.needs_input_grad[2] or has_b and ctx.needs_input_grad[3]
            dx = torch.empty_like(x)
            dpre = torch.empty_like(x) if (need_pre and has_d) else None
            with torch.cuda.device(x.device):
                native.check(native.lib().mg_fma_bias_act_bwd(
                    native.ptr(dy), native.ptr(y), native.ptr(d32 if has_d else None), native.ptr(dx), native.ptr(dpre),
                    native.dtype_code(x.dtype), idx, alpha, gain, clamp, n, c, h * w, native.stream_ptr(x.device)))
            if dpre is None:
                dpre = dx                      # without dcoefs, dx IS the pre-activation gradient
            gd = gn = gb = None
            if has_d and ctx.needs_input_grad[1]:
                gd = (dpre * x).sum(dim=[2, 3]).to(ctx.meta[0])
            if has_nz and ctx.needs_input_grad[2]:
                gn = dpre.sum(dim=[0, 1]).reshape(ctx.meta[1][1]).to(ctx.meta[1][0])
            if has_b and ctx.needs_input_grad[3]:
                gb = dpre.sum(dim=[0, 2, 3]).to(ctx.meta[2])
            return (dx if ctx.needs_input_grad[0] else None), gd, gn, gb

    _cache[key] = ModconvEpilogue
    return ModconvEpilogue


def modconv_epilogue(x, dcoefs=None, noise=None, b=None, act='lrelu', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(act(x * dcoefs[N,C,1,1] + noise[H,W] + b[C]) * gain).

    x [N,C,H,W]; dcoefs [N,C] or None; noise [H,W] (already multiplied by its strength) or None;
    b [C] or None.  First-order gradients on the native path (the generator is never
    differentiated twice through its activations: path-length regularisation differentiates
    w.r.t. ws only once with create_graph, which needs second order -- callers use
    ``impl='ref'`` semantics there; see training/networks.py)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    spec = _ba.activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    if impl == 'cuda' and x.device.type == 'cuda' and x.dtype != torch.float64:
        return _epilogue_hip(act, alpha, gain, clamp).apply(x, dcoefs, noise, b)
    return _epilogue_ref(x, dcoefs, noise, b, act, alpha, gain, None if clamp < 0 else clamp)
