"""Plugin loader with the reference's interface (montage_gan/torch_utils/custom_ops.py:23,49).

``get_plugin(module_name, sources, **build_kwargs)`` returns an object exposing the same
functions as the reference's pybind modules:

    upfirdn2d_plugin.upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain)
    bias_act_plugin.bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)

They are the functions of a ``torch.utils.cpp_extension`` pybind module (csrc/torch_plugin.cpp: at::Tensor arguments,
TORCH_CHECK validation, output allocation with ``suggest_memory_format``, PyTorch's current HIP stream) over the C ABI of
libmontage_hip.so (include/montage_hip.h) -- the "C-ABI shim" of the reference's loader (custom_ops.py:49-129).  Both
are compiled ahead of time and in-tree (``native.build()``) instead of JIT-built per process, so there is no FileBaton / md5
build cache to keep; if the shim is missing, ``get_plugin`` builds it on the spot from ``sources`` (``.cpp`` files; default
csrc/torch_plugin.cpp) with ``build_kwargs`` (extra_cflags / extra_ldflags / extra_include_paths).  ``MONTAGE_PLUGIN=ctypes``
selects the pure-ctypes binding of the same C ABI instead (what tests/test_abi_cpu.py exercises without a GPU).
A load failure RAISES; callers never get a silent slow path.
"""
import os

import torch

from . import native

verbosity = 'brief'  # 'none' | 'brief' | 'full'   (custom_ops.py:23)

_cached_plugins = dict()
_null = None


def _is_null(t):
    return t is None or t.numel() == 0


def _suggest_memory_format(x):
    # at::Tensor::suggest_memory_format() for rank 4 (upfirdn2d.cpp:35)
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous():
        return torch.channels_last
    return torch.contiguous_format


def _has_same_layout(a, b):
    # bias_act.cpp:16-28
    if a.dim() != b.dim():
        return False
    for i in range(a.dim()):
        if a.size(i) != b.size(i):
            return False
        if a.size(i) >= 2 and a.stride(i) != b.stride(i):
            return False
    return True


def _is_non_overlapping_and_dense(t):
    """at::Tensor::is_non_overlapping_and_dense(): some permutation of the dims is contiguous."""
    if t.numel() <= 1 or t.is_contiguous():
        return True
    dims = sorted((d for d in range(t.dim()) if t.size(d) != 1), key=lambda d: t.stride(d))
    expect = 1
    for d in dims:
        if t.stride(d) != expect:
            return False
        expect *= t.size(d)
    return True


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


class _Upfirdn2dPlugin:
    """Stands where ``upfirdn2d_plugin`` (upfirdn2d.cpp:98-101) stood."""

    @staticmethod
    def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
        _require(x.is_cuda, 'x must reside on CUDA device')                                    # upfirdn2d.cpp:19
        _require(f.device == x.device, 'f must reside on the same device as x')                # :20
        _require(f.dtype == torch.float32, 'f must be float32')                                # :21
        _require(x.numel() <= 2 ** 31 - 1, 'x is too large')                                   # :22
        _require(f.numel() <= 2 ** 31 - 1, 'f is too large')                                   # :23
        _require(x.dim() == 4, 'x must be rank 4')                                             # :24
        _require(f.dim() == 2, 'f must be rank 2')                                             # :25
        _require(f.size(0) >= 1 and f.size(1) >= 1, 'f must be at least 1x1')                  # :26
        _require(upx >= 1 and upy >= 1, 'upsampling factor must be at least 1')                # :27
        _require(downx >= 1 and downy >= 1, 'downsampling factor must be at least 1')          # :28
        n, c, ih, iw = x.shape
        key = (n, c, ih, iw, x.stride(), f.shape[0], f.shape[1], f.stride(), upx, upy, downx, downy, padx0, padx1, pady0, pady1)
        plan = _uf_plans.get(key)
        if plan is None:
            def cdiv(a, b):  # C integer division truncates toward zero (upfirdn2d.cpp:32-33)
                return int(a / b) if a * b < 0 else a // b

            ow = cdiv(iw * upx + padx0 + padx1 - f.size(1) + downx, downx)
            oh = cdiv(ih * upy + pady0 + pady1 - f.size(0) + downy, downy)
            _require(ow >= 1 and oh >= 1, 'output must be at least 1x1')                       # :34
            _require(n * c * oh * ow <= 2 ** 31 - 1, 'output is too large')                    # :36
            fmt = _suggest_memory_format(x)
            ys = torch.empty([n, c, oh, ow], dtype=torch.uint8, device='meta', memory_format=fmt).stride()
            plan = (oh, ow, fmt, native.I32x4(n, c, ih, iw), native.I64x4(*x.stride()), native.I32x2(f.size(0), f.size(1)),
                    native.I64x2(*f.stride()), native.I32x4(n, c, oh, ow), native.I64x4(*ys),
                    int(upx), int(upy), int(downx), int(downy), int(padx0), int(pady0))
            if len(_uf_plans) < 4096:
                _uf_plans[key] = plan
        oh, ow, fmt = plan[0], plan[1], plan[2]
        y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device, memory_format=fmt)
        if y.numel() == 0:
            return y
        with native.guard(x.device):
            rc = native.lib().mg_upfirdn2d(
                x.data_ptr(), f.data_ptr(), y.data_ptr(), native.dtype_code(x.dtype), *plan[3:],
                1 if flip else 0, float(gain), native.stream_ptr(x.device))
        if rc:
            native.check(rc)
        return y


_uf_plans = dict()


class _BiasActPlugin:
    """Stands where ``bias_act_plugin`` (bias_act.cpp:94-97) stood."""

    @staticmethod
    def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
        has_b, has_xref, has_yref, has_dy = not _is_null(b), not _is_null(xref), not _is_null(yref), not _is_null(dy)
        _require(x.is_cuda, 'x must reside on CUDA device')                                    # bias_act.cpp:35
        _require(not has_b or (b.dtype == x.dtype and b.device == x.device),
                 'b must have the same dtype and device as x')                                 # :36
        if has_xref:
            _require(xref.shape == x.shape and xref.dtype == x.dtype and xref.device == x.device,
                     'xref must have the same shape, dtype, and device as x')                  # :37
        if has_yref:
            _require(yref.shape == x.shape and yref.dtype == x.dtype and yref.device == x.device,
                     'yref must have the same shape, dtype, and device as x')                  # :38
        if has_dy:
            _require(dy.shape == x.shape and dy.dtype == x.dtype and dy.device == x.device,
                     'dy must have the same dtype and device as x')                            # :39
        size_x = x.numel()
        _require(size_x <= 2 ** 31 - 1, 'x is too large')                                      # :40
        _require(b is None or b.dim() == 1, 'b must have rank 1')                              # :41
        if has_b:
            _require(0 <= dim < x.dim(), 'dim is out of bounds')                               # :42
            _require(b.numel() == x.size(dim), 'b has wrong number of elements')               # :43
            _require(b.is_contiguous(), 'b must be contiguous')                                # :48
        _require(grad >= 0, 'grad must be non-negative')                                       # :44
        _require(_is_non_overlapping_and_dense(x), 'x must be non-overlapping and dense')      # :47
        xs = x.stride()
        for t, name, has in ((xref, 'xref', has_xref), (yref, 'yref', has_yref), (dy, 'dy', has_dy)):
            # same shape (checked above) + same strides = same layout; the slow path handles size-1 dims (:16-28)
            _require(not has or t.stride() == xs or _has_same_layout(t, x), f'{name} must have the same layout as x')  # :49-51
        y = torch.empty_like(x)
        _require(y.stride() == xs or _has_same_layout(y, x), 'y must have the same layout as x')   # :56
        if size_x == 0:
            return y
        with native.guard(x.device):
            rc = native.lib().mg_bias_act(
                x.data_ptr(), b.data_ptr() if has_b else None, xref.data_ptr() if has_xref else None,
                yref.data_ptr() if has_yref else None, dy.data_ptr() if has_dy else None, y.data_ptr(), native.dtype_code(x.dtype),
                int(grad), int(act), float(alpha), float(gain), float(clamp),
                size_x, b.numel() if has_b else 0, x.stride(dim) if has_b else 1,
                native.stream_ptr(x.device))
        if rc:
            native.check(rc)
        return y


_PLUGINS = {'upfirdn2d_plugin': _Upfirdn2dPlugin, 'bias_act_plugin': _BiasActPlugin}


class _TorchPlugin:
    """One reference plugin name -> the functions of the shared pybind module."""

    def __init__(self, mod, names):
        for n in names:
            setattr(self, n, getattr(mod, n))


_EXPORTS = {'upfirdn2d_plugin': ('upfirdn2d',), 'bias_act_plugin': ('bias_act',)}


def get_plugin(module_name, sources=None, **build_kwargs):
    assert verbosity in ['none', 'brief', 'full']
    if module_name in _cached_plugins:
        return _cached_plugins[module_name]
    if module_name not in _PLUGINS:
        raise RuntimeError(f'unknown plugin "{module_name}"; libmontage_hip.so provides {sorted(_PLUGINS)}')
    if verbosity != 'none':
        print(f'Setting up PyTorch plugin "{module_name}"... ', end='', flush=True)
    try:
        native.lib()          # raises when the shared library is missing
        if os.environ.get('MONTAGE_PLUGIN', 'torch') == 'ctypes':
            plugin = _PLUGINS[module_name]
        else:
            if not os.path.exists(native.torch_plugin_path()):
                cpp = [s for s in (sources or []) if str(s).endswith('.cpp') and os.path.exists(s)]
                native.build_torch_plugin(sources=cpp or None, verbose=(verbosity == 'full'),
                                          **{k: v for k, v in build_kwargs.items() if k in ('extra_cflags', 'extra_ldflags', 'extra_include_paths')})
            plugin = _TorchPlugin(native.torch_plugin(), _EXPORTS[module_name])
    except Exception:
        if verbosity != 'none':
            print('Failed!')
        raise
    if verbosity != 'none':
        print('Done.')
    _cached_plugins[module_name] = plugin
    return plugin
