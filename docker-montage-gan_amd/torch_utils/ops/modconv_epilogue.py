"""Fused epilogue of the (non-fused-weights) modulated convolution.

The reference's SynthesisLayer finishes every convolution with two separate ops
(training/networks.py:86-88 then :311-313):

    x = fma.fma(x, dcoefs.reshape(N,-1,1,1), noise)            # read x, write x
    x = bias_act.bias_act(x, b, act=act, gain=gain, clamp=c)  # read x, write x

``modconv_epilogue`` is the same arithmetic -- including the rounding of the intermediate to
x.dtype -- in one pass over the activation (csrc/fused.hip).  It is an additive op: callers that
keep calling fma + bias_act get identical values, at three HBM round trips instead of one.

Gradients: x, dcoefs, noise and b.  A plain ``backward()`` (grad mode off inside backward) takes
the fused native gradient kernel; under ``create_graph=True`` (path-length regularisation
differentiates the generator twice) the backward is assembled from ops that are themselves
differentiable (the native bias_act gradient Function + aten products/reductions), so gradients
of any order the underlying ops support are available.
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
    return _ba.bias_act(x, b.to(x.dtype) if b is not None else None, act=act, alpha=alpha, gain=gain, clamp=clamp)


_cache = dict()


def _epilogue_hip(act, alpha, gain, clamp):
    key = (act, alpha, gain, clamp)
    if key in _cache:
        return _cache[key]
    spec = _ba.activation_funcs[act]
    assert 'x' not in spec.ref and not spec.has_2nd_grad, f'{act}: only piecewise-linear activations are fused'
    idx = spec.cuda_idx
    _ba._init()
    grad_fn = _ba._bias_act_cuda(dim=1, act=act, alpha=alpha, gain=gain, clamp=(None if clamp < 0 else clamp)).Grad
    null = _ba._null_tensor

    class ModconvEpilogue(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, dcoefs, noise, b):
            n, c, h, w = x.shape
            x = x.contiguous()
            d32 = dcoefs.detach().to(torch.float32).contiguous() if dcoefs is not None else None
            nz32, nz_stride = None, 0
            if noise is not None:
                assert noise.numel() in (h * w, n * h * w), 'noise must be [H,W] or [N,1,H,W]'
                nz32 = noise.detach().to(torch.float32).contiguous()
                nz_stride = h * w if noise.numel() == n * h * w and n > 1 else 0
            bb = b.detach().to(x.dtype).contiguous() if b is not None else None
            y = torch.empty_like(x)
            if x.numel():
                with torch.cuda.device(x.device):
                    native.check(native.lib().mg_fma_bias_act(
                        native.ptr(x), native.ptr(d32), native.ptr(nz32), native.ptr(bb), native.ptr(y),
                        native.dtype_code(x.dtype), idx, alpha, gain, clamp, 1.0, nz_stride, n, c, h * w,
                        native.stream_ptr(x.device)))
            ctx.save_for_backward(x, dcoefs if dcoefs is not None else null, y)
            ctx.has = (dcoefs is not None, noise is not None, b is not None)
            ctx.noise_meta = None if noise is None else (noise.dtype, noise.shape)
            ctx.b_dtype = None if b is None else b.dtype
            return y

        @staticmethod
        def backward(ctx, dy):
            x, dcoefs, y = ctx.saved_tensors
            n, c, h, w = x.shape
            has_d, has_nz, has_b = ctx.has
            need = ctx.needs_input_grad
            dy = dy.contiguous()
            if torch.is_grad_enabled():
                # differentiable assembly (create_graph=True)
                dpre = grad_fn.apply(dy, null, null, y) if not (act == 'linear' and gain == 1 and clamp < 0) else dy
                dx = dpre * dcoefs.to(dpre.dtype).reshape(n, c, 1, 1) if has_d else dpre
            else:
                want_pre = has_d and (need[1] or (has_nz and need[2]) or (has_b and need[3]))
                dx = torch.empty_like(x)
                dpre = torch.empty_like(x) if want_pre else None
                d32 = dcoefs.to(torch.float32).contiguous() if has_d else None
                if x.numel():
                    with torch.cuda.device(x.device):
                        native.check(native.lib().mg_fma_bias_act_bwd(
                            native.ptr(dy), native.ptr(y), native.ptr(d32), native.ptr(dx), native.ptr(dpre),
                            native.dtype_code(x.dtype), idx, alpha, gain, clamp, n, c, h * w, native.stream_ptr(x.device)))
                if dpre is None:
                    dpre = dx                  # without dcoefs, dx IS the pre-activation gradient
            gd = gn = gb = None
            if has_d and need[1]:
                gd = (dpre * x).sum(dim=[2, 3]).to(dcoefs.dtype)
            if has_nz and need[2]:
                dt, shp = ctx.noise_meta
                per_sample = len(shp) == 4 and shp[0] == n and n > 1
                gn = (dpre.sum(dim=1, keepdim=True) if per_sample else dpre.sum(dim=[0, 1])).reshape(shp).to(dt)
            if has_b and need[3]:
                gb = dpre.sum(dim=[0, 2, 3]).to(ctx.b_dtype)
            return (dx if need[0] else None), gd, gn, gb

    _cache[key] = ModconvEpilogue
    return ModconvEpilogue


def modconv_epilogue(x, dcoefs=None, noise=None, b=None, act='lrelu', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(act(x * dcoefs[N,C,1,1] + noise + b[C]) * gain).

    x [N,C,H,W]; dcoefs [N,C] or None; noise [H,W] / [1,1,H,W] (shared) or [N,1,H,W] (per sample),
    already multiplied by its strength, or None; b [C] or None."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    spec = _ba.activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    fusable = 'x' not in spec.ref and not spec.has_2nd_grad
    if impl == 'cuda' and x.device.type == 'cuda' and x.dtype != torch.float64 and fusable and x.is_contiguous():
        return _epilogue_hip(act, alpha, gain, clamp).apply(x, dcoefs, noise, b)
    return _epilogue_ref(x, dcoefs, noise, b, act, alpha, gain, None if clamp < 0 else clamp)
