"""Fused bias + activation + gain + clamp.

Public interface identical to the reference (montage_gan/torch_utils/ops/bias_act.py):
``activation_funcs`` (:25-43) and ``bias_act(x, b, dim, act, alpha, gain, clamp, impl)`` (:71).

``impl='cuda'`` on a GPU tensor runs ``csrc/bias_act.hip`` through
``custom_ops.get_plugin('bias_act_plugin')`` -- forward and the explicit first and second
order gradient kernels -- and raises if the native library is unavailable (no silent
fallback).  ``impl='ref'`` / CPU tensors use the aten composition that defines the API.
"""
import numpy as np
import torch

from .. import custom_ops
from .. import misc
from .. import native


class _ActSpec(dict):
    """Attribute-style record, same fields as the reference's EasyDict entries."""
    __getattr__ = dict.__getitem__


def _spec(func, def_alpha, def_gain, cuda_idx, ref, has_2nd_grad):
    return _ActSpec(func=func, def_alpha=def_alpha, def_gain=def_gain, cuda_idx=cuda_idx, ref=ref,
                    has_2nd_grad=has_2nd_grad)


_F = torch.nn.functional
# name -> (aten function, default alpha, default gain, native kernel id, which tensor the gradient
# kernel needs ('x', 'y' or ''), whether a second-order kernel exists)
activation_funcs = {
    'linear':   _spec(lambda x, **_: x,                         0,   1,          1, '',  False),
    'relu':     _spec(lambda x, **_: _F.relu(x),                0,   np.sqrt(2), 2, 'y', False),
    'lrelu':    _spec(lambda x, alpha, **_: _F.leaky_relu(x, alpha), 0.2, np.sqrt(2), 3, 'y', False),
    'tanh':     _spec(lambda x, **_: torch.tanh(x),             0,   1,          4, 'y', True),
    'sigmoid':  _spec(lambda x, **_: torch.sigmoid(x),          0,   1,          5, 'y', True),
    'elu':      _spec(lambda x, **_: _F.elu(x),                 0,   1,          6, 'y', True),
    'selu':     _spec(lambda x, **_: _F.selu(x),                0,   1,          7, 'y', True),
    'softplus': _spec(lambda x, **_: _F.softplus(x),            0,   1,          8, 'y', True),
    'swish':    _spec(lambda x, **_: torch.sigmoid(x) * x,      0,   np.sqrt(2), 9, 'x', True),
}

_plugin = None
_null_tensor = torch.empty([0])     # "absent" marker across the plugin boundary (reference :49)


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin('bias_act_plugin', sources=['bias_act.hip'])
    return True


def _resolve(act, alpha, gain, clamp):
    spec = activation_funcs[act]
    assert clamp is None or clamp >= 0
    return (spec, float(spec.def_alpha if alpha is None else alpha), float(spec.def_gain if gain is None else gain),
            float(-1 if clamp is None else clamp))


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(act(x + b.reshape(broadcast along dim)) * gain, -clamp, +clamp).

    x any shape; b 1-D with b.shape[0] == x.shape[dim] or None; act a key of
    ``activation_funcs``; alpha/gain None = the activation's defaults; clamp None = off.
    First and second order gradients are supported (reference :71-105)."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda':
        _init()
        return _bias_act_cuda(dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp).apply(x, b)
    return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)


@misc.profiled_function
def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """aten composition (reference :110-140)."""
    assert isinstance(x, torch.Tensor)
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    x = spec.func(x, alpha=alpha)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


_bias_act_cuda_cache = dict()


def _bias_act_cuda(dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """autograd.Function class for one scalar configuration (cached, reference :145-230)."""
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    key = (dim, act, alpha, gain, clamp)
    cls = _bias_act_cuda_cache.get(key)
    if cls is not None:
        return cls

    idx = spec.cuda_idx
    keep_x = ('x' in spec.ref) or spec.has_2nd_grad
    keep_y = 'y' in spec.ref
    is_identity = (act == 'linear' and gain == 1 and clamp < 0)

    def layout_of(t):
        return torch.channels_last if t.ndim > 2 and t.stride()[1] == 1 else torch.contiguous_format

    class BiasActCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):
            ctx.memory_format = layout_of(x)
            x = x.contiguous(memory_format=ctx.memory_format)
            b = b.contiguous() if b is not None else _null_tensor
            y = x
            if not is_identity or b is not _null_tensor:
                y = _plugin.bias_act(x, b, _null_tensor, _null_tensor, _null_tensor, 0, dim, idx, alpha, gain, clamp)
            ctx.save_for_backward(x if keep_x else _null_tensor, b if keep_x else _null_tensor,
                                  y if keep_y else _null_tensor)
            return y

        @staticmethod
        def backward(ctx, dy):
            dy = dy.contiguous(memory_format=ctx.memory_format)
            x, b, y = ctx.saved_tensors
            dx = db = None
            # Plain first-order backward of the network layers (NCHW or NHWC, bias on dim 1, piecewise-linear activation):
            # ONE native pass writes dx and leaves sum(dx) per channel in a float32 accumulator, instead of the
            # gradient kernel followed by a separate reduction over dx (upstream :186-196).
            # (linear saves no output, so the kernel has nothing to evaluate the gain against: only gain == 1 qualifies)
            if (ctx.needs_input_grad[1] and not torch.is_grad_enabled() and not is_identity and (idx in (2, 3) or (idx == 1 and gain == 1)) and dim == 1
                    and dy.ndim == 4 and dy.dtype != torch.float64 and dy.numel() > 0):
                n, c, h, w = dy.shape
                vec = 16 // dy.element_size()
                nhwc = ctx.memory_format == torch.channels_last
                if not nhwc or (c % vec == 0 and c // vec <= 256 and 256 % (c // vec) == 0):
                    dx = torch.empty_like(dy)
                    # the channels-last kernel leaves per-sample partial sums [N, C]
                    acc = torch.zeros([n * c if nhwc else c], dtype=torch.float32, device=dy.device)
                    with native.guard(dy.device):
                        bwd = native.lib().mg_bwd_fused_nhwc if nhwc else native.lib().mg_bwd_fused
                        native.check(bwd(
                            native.ptr(dy), native.ptr(y if y.numel() else None), native.ptr(None), native.ptr(None), native.ptr(None),
                            native.ptr(dx), native.ptr(None), native.ptr(acc), native.ptr(None), native.dtype_code(dy.dtype), idx,
                            alpha, gain, clamp, 0, n, c, h * w, native.stream_ptr(dy.device)))
                    db = acc.reshape(n, c).sum(0) if nhwc else acc
                    return (dx if ctx.needs_input_grad[0] else None), db.to(dy.dtype)
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                if is_identity:
                    dx = dy
                elif torch.is_grad_enabled():
                    dx = BiasActCudaGrad.apply(dy, x, b, y)
                else:       # plain backward: the gradient kernel without another autograd node
                    dx = _plugin.bias_act(dy, b, x, y, _null_tensor, 1, dim, idx, alpha, gain, clamp)
            if ctx.needs_input_grad[1]:
                db = dx.sum([i for i in range(dx.ndim) if i != dim])
            return dx, db

    class BiasActCudaGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b, y):
            ctx.memory_format = layout_of(dy)
            dx = _plugin.bias_act(dy, b, x, y, _null_tensor, 1, dim, idx, alpha, gain, clamp)
            ctx.save_for_backward(dy if spec.has_2nd_grad else _null_tensor, x, b, y)
            return dx

        @staticmethod
        def backward(ctx, d_dx):
            d_dx = d_dx.contiguous(memory_format=ctx.memory_format)
            dy, x, b, y = ctx.saved_tensors
            d_dy = d_x = d_b = None
            if ctx.needs_input_grad[0]:
                d_dy = BiasActCudaGrad.apply(d_dx, x, b, y)
            if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                d_x = _plugin.bias_act(d_dx, b, x, y, dy, 2, dim, idx, alpha, gain, clamp)
            if spec.has_2nd_grad and ctx.needs_input_grad[2]:
                d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
            return d_dy, d_x, d_b, None

    BiasActCuda.Grad = BiasActCudaGrad      # first-order gradient op, itself differentiable (used by modconv_epilogue)
    _bias_act_cuda_cache[key] = BiasActCuda
    return BiasActCuda
