"""ctypes/numpy front end of the CPU oracle (oracle/montage_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, by
``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of bench.py --
never by the product package ``docker-montage-gan_amd``.  Parity status:
pinned by tests/golden (see montage_oracle.c header).

All functions take and return dense numpy arrays (float32 or float64, NCHW).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libmontage_oracle.so')
_lib = None

c_int, c_long, c_float, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p


def build(force=False):
    """Compile the oracle with gcc (a few hundred ms)."""
    src = [os.path.join(_HERE, n) for n in ('montage_oracle.c', 'montage_oracle_body.inc')]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.run(['make', '-C', _HERE, '-B'], check=True, stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_upfirdn2d_out_size.restype = c_int
    return _lib


def _sfx(dtype):
    if dtype == np.float32:
        return '_f32'
    if dtype == np.float64:
        return '_f64'
    raise TypeError(f'oracle handles float32/float64, got {dtype}')


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _dense(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def out_size(in_size, up, down, pad0, pad1, fsize):
    return int(lib().orc_upfirdn2d_out_size(in_size, up, down, pad0, pad1, fsize))


def index_table(out_sz, in_size, up, down, pad0, fsize, flip):
    """int32 [out_sz, 3] = (in0, count, filter0) per output coordinate."""
    t = np.zeros([out_sz, 3], dtype=np.int32)
    lib().orc_upfirdn2d_index_table(_p(t), c_int(out_sz), c_int(in_size), c_int(up), c_int(down),
                                    c_int(pad0), c_int(fsize), c_int(int(bool(flip))))
    return t


def upfirdn2d(x, f, up=(1, 1), down=(1, 1), padding=(0, 0, 0, 0), flip_filter=False, gain=1.0):
    """x [N,C,H,W]; f float32 [fh,fw] (2-D; None = identity); up/down = (x, y);
    padding = (x0, x1, y0, y1).  One fused 2-D pass, as upfirdn2d_plugin.upfirdn2d."""
    dt = x.dtype
    x = _dense(x, dt)
    if f is None:
        f = np.ones([1, 1], dtype=np.float32)
    f = _dense(f, np.float32)
    assert x.ndim == 4 and f.ndim == 2
    n, c, ih, iw = x.shape
    fh, fw = f.shape
    upx, upy = up
    downx, downy = down
    px0, px1, py0, py1 = padding
    ow = out_size(iw, upx, downx, px0, px1, fw)
    oh = out_size(ih, upy, downy, py0, py1, fh)
    assert ow >= 1 and oh >= 1
    y = np.empty([n, c, oh, ow], dtype=dt)
    getattr(lib(), 'orc_upfirdn2d' + _sfx(dt))(
        _p(x), _p(f), _p(y), c_int(n), c_int(c), c_int(ih), c_int(iw), c_int(fh), c_int(fw),
        c_int(oh), c_int(ow), c_int(upx), c_int(upy), c_int(downx), c_int(downy),
        c_int(px0), c_int(py0), c_int(int(bool(flip_filter))), c_float(gain))
    return y


def upfirdn2d_api(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """Same calling convention as torch_utils.ops.upfirdn2d.upfirdn2d (upfirdn2d.py:132):
    ints or [x, y] lists, 2- or 4-element padding, 1-D f = separable (two passes with
    sqrt(gain) each, upfirdn2d.py:253-257)."""
    up = (up, up) if isinstance(up, int) else tuple(up)
    down = (down, down) if isinstance(down, int) else tuple(down)
    if isinstance(padding, int):
        padding = [padding, padding]
    padding = list(padding)
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    if f is not None and np.asarray(f).ndim == 1:
        f = np.asarray(f, dtype=np.float32)
        g = float(np.sqrt(gain))
        y = upfirdn2d(x, f[None, :], (up[0], 1), (down[0], 1), (padding[0], padding[1], 0, 0), flip_filter, g)
        return upfirdn2d(y, f[:, None], (1, up[1]), (1, down[1]), (0, 0, padding[2], padding[3]), flip_filter, g)
    return upfirdn2d(x, f, up, down, tuple(padding), flip_filter, gain)


def bias_act_raw(x, b=None, xref=None, yref=None, dy=None, grad=0, dim=1, act=1, alpha=0.0, gain=1.0,
                 clamp=-1.0):
    """The plugin-level call bias_act_plugin.bias_act (bias_act.cpp:32) on dense C-order arrays."""
    dt = x.dtype
    x = _dense(x, dt)
    b, xref, yref, dy = (_dense(a, dt) for a in (b, xref, yref, dy))
    size_b = 0 if b is None else b.size
    step_b = 1
    if b is not None:
        step_b = int(np.prod(x.shape[dim + 1:], dtype=np.int64)) if x.ndim > dim + 1 else 1
    y = np.empty_like(x)
    getattr(lib(), 'orc_bias_act' + _sfx(dt))(
        _p(x), _p(b), _p(xref), _p(yref), _p(dy), _p(y), c_int(grad), c_int(act), c_float(alpha),
        c_float(gain), c_float(clamp), c_long(x.size), c_int(max(size_b, 1)), c_long(max(step_b, 1)))
    return y


ACTS = {  # name -> (cuda_idx, def_alpha, def_gain, ref, has_2nd_grad)   bias_act.py:25-43
    'linear': (1, 0.0, 1.0, '', False),
    'relu': (2, 0.0, float(np.sqrt(2)), 'y', False),
    'lrelu': (3, 0.2, float(np.sqrt(2)), 'y', False),
    'tanh': (4, 0.0, 1.0, 'y', True),
    'sigmoid': (5, 0.0, 1.0, 'y', True),
    'elu': (6, 0.0, 1.0, 'y', True),
    'selu': (7, 0.0, 1.0, 'y', True),
    'softplus': (8, 0.0, 1.0, 'y', True),
    'swish': (9, 0.0, float(np.sqrt(2)), 'x', True),
}


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    idx, da, dg, _, _ = ACTS[act]
    alpha = da if alpha is None else alpha
    gain = dg if gain is None else gain
    clamp = -1.0 if clamp is None else clamp
    return bias_act_raw(x, b, grad=0, dim=dim, act=idx, alpha=alpha, gain=gain, clamp=clamp)


def alpha_composite(x):
    """x [B,L,4,H,W] -> [B,4,H,W]  (image_utils.py:112-163, default branch)."""
    dt = x.dtype
    x = _dense(x, dt)
    b, l, c, h, w = x.shape
    assert c == 4 and 1 <= l <= 16
    y = np.empty([b, 4, h, w], dtype=dt)
    getattr(lib(), 'orc_alpha_composite_fwd' + _sfx(dt))(_p(x), _p(y), c_int(b), c_int(l), c_int(h), c_int(w))
    return y


def alpha_composite_bwd(x, dy, safe_grad=False):
    dt = x.dtype
    x, dy = _dense(x, dt), _dense(dy, dt)
    b, l, c, h, w = x.shape
    dx = np.empty_like(x)
    getattr(lib(), 'orc_alpha_composite_bwd' + _sfx(dt))(_p(x), _p(dy), _p(dx), c_int(b), c_int(l), c_int(h),
                                                         c_int(w), c_int(int(safe_grad)))
    return dx


def translate_sample(x, t, shift=0.0):
    """x [M,C,H,W], t float32 [M,2] -> y  (fukuwarai/networks.py:246-255)."""
    dt = x.dtype
    x = _dense(x, dt)
    t = _dense(t, np.float32)
    m, c, h, w = x.shape
    y = np.empty_like(x)
    getattr(lib(), 'orc_translate_sample_fwd' + _sfx(dt))(_p(x), _p(t), _p(y), c_int(m), c_int(c), c_int(h),
                                                          c_int(w), c_float(shift))
    return y


def translate_sample_bwd(x, dy, t, shift=0.0):
    """-> (dx like x, dt float64 [M,2])."""
    dt_ = x.dtype
    x, dy = _dense(x, dt_), _dense(dy, dt_)
    t = _dense(t, np.float32)
    m, c, h, w = x.shape
    dx = np.empty_like(x)
    gt = np.zeros([m, 2], dtype=np.float64)
    getattr(lib(), 'orc_translate_sample_bwd' + _sfx(dt_))(_p(x), _p(dy), _p(t), _p(dx), _p(gt), c_int(m), c_int(c),
                                                           c_int(h), c_int(w), c_float(shift))
    return dx, gt


def affine_warp(x, theta, out_h, out_w):
    """grid_sample(x, affine_grid(theta, [N,C,out_h,out_w])), bilinear / zeros / align_corners=False
    (training/augment.py:376-377).  x [N,C,H,W]; theta float32 [N,2,3]."""
    dt = x.dtype
    x = _dense(x, dt)
    theta = _dense(theta, np.float32)
    n, c, h, w = x.shape
    y = np.empty([n, c, out_h, out_w], dtype=dt)
    getattr(lib(), 'orc_affine_warp_fwd' + _sfx(dt))(_p(x), _p(theta), _p(y), c_int(n), c_int(c), c_int(h), c_int(w),
                                                     c_int(out_h), c_int(out_w))
    return y


def affine_warp_bwd(dy, theta, in_h, in_w):
    """gradient of affine_warp w.r.t. x for upstream gradient dy [N,C,out_h,out_w]."""
    dt = dy.dtype
    dy = _dense(dy, dt)
    theta = _dense(theta, np.float32)
    n, c, oh, ow = dy.shape
    dx = np.empty([n, c, in_h, in_w], dtype=dt)
    getattr(lib(), 'orc_affine_warp_bwd' + _sfx(dt))(_p(dy), _p(theta), _p(dx), c_int(n), c_int(c), c_int(in_h), c_int(in_w),
                                                     c_int(oh), c_int(ow))
    return dx


def channel_mix(x, A, t=None):
    """y[n,i,...] = sum_j A[n,i,j] x[n,j,...] + t[n,i]  (training/augment.py:403-406)."""
    dt = x.dtype
    x = _dense(x, dt)
    A = _dense(A, np.float32)
    t = _dense(t, np.float32) if t is not None else None
    n, c = x.shape[:2]
    hw = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    getattr(lib(), 'orc_channel_mix' + _sfx(dt))(_p(x), _p(A), _p(t) if t is not None else None, _p(y), c_int(n), c_int(c),
                                                 ctypes.c_long(hw))
    return y


def demod_coefs(w, s):
    """d[n,o] = rsqrt(sum_{i,kh,kw} (w[o,i,kh,kw] * s[n,i])^2 + 1e-8)  (training/networks.py:76-77 on the per-sample
    weights w[None] * s[:, None, :, None, None]; evaluated literally that way here, in float64)."""
    w = np.asarray(w, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64)
    per_sample = w[None] * s[:, None, :, None, None]                     # [N,O,I,kh,kw]
    return 1.0 / np.sqrt((per_sample ** 2).sum(axis=(2, 3, 4)) + 1e-8)


def demod_coefs_bwd(w, s, g):
    """Gradients of sum(d * g) w.r.t. w and s (float64, by the chain rule on the definition above)."""
    w = np.asarray(w, dtype=np.float64)
    s = np.asarray(s, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    d = demod_coefs(w, s)
    t = -0.5 * g * d ** 3                                                # d/d(sum of squares)
    wsq = (w ** 2).sum(axis=(2, 3))                                      # [O,I]
    ds = 2 * s * (t @ wsq)
    dw = 2 * w * (t.T @ (s ** 2))[:, :, None, None]
    return dw, ds
