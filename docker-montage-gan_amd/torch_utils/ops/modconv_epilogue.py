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

The first-order backward is ONE kernel too (``mg_bwd_fused``): it reads dy, y and the convolution
output x once, writes dx, and leaves sum(dpre*x) per (n,c), sum(dpre) per channel and
sum(dpre*noise_raw) in float32 accumulators -- the gradients of dcoefs, bias and noise_strength that
autograd would otherwise produce with five more passes over the activation.  ``channel_scale`` is the
same machinery for the modulation ``x * styles[N,C,1,1]`` that precedes the convolution.
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


def _is_nhwc(x):
    """Dense channels-last 4-D tensor the NHWC kernels accept (channel count a multiple of the 16-byte vector that
    divides 256 vectors, so the fused backward can keep one channel vector per lane)."""
    if x.ndim != 4 or x.is_contiguous() or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    vec = 16 // x.element_size()
    c = x.shape[1]
    return c % vec == 0 and c // vec <= 256 and 256 % (c // vec) == 0


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
        """inputs: x, dcoefs, noise (scaled, or raw when `strength` is given), strength (0-dim or None), b"""

        @staticmethod
        def forward(ctx, x, dcoefs, noise, strength, b):
            n, c, h, w = x.shape
            nhwc = _is_nhwc(x)
            if not nhwc:
                x = x.contiguous()
            # the reference multiplies by dcoefs.to(x.dtype): the kernel applies that rounding to the float32 values it is handed
            d32 = dcoefs.detach().to(torch.float32).contiguous() if dcoefs is not None else None       # rounded to x.dtype inside the kernel
            nz32, nz_stride, s32 = None, 0, None
            if noise is not None:
                assert noise.numel() in (h * w, n * h * w), 'noise must be [H,W] or [N,1,H,W]'
                nz32 = noise.detach().to(torch.float32).contiguous()
                nz_stride = h * w if noise.numel() == n * h * w and n > 1 else 0
                if strength is not None:        # stays on the device: the kernel reads the scalar itself
                    s32 = strength.detach().to(torch.float32).reshape(1).contiguous()
            bb = b.detach().to(x.dtype).contiguous() if b is not None else None
            y = torch.empty_like(x)         # keeps the (dense) layout of x
            if x.numel():
                with native.guard(x.device):
                    fwd = native.lib().mg_fma_bias_act_nhwc if nhwc else native.lib().mg_fma_bias_act
                    native.check(fwd(
                        native.ptr(x), native.ptr(d32), native.ptr(nz32), native.ptr(bb), native.ptr(y),
                        native.dtype_code(x.dtype), idx, alpha, gain, clamp, 1.0, native.ptr(s32), nz_stride, n, c, h * w,
                        native.stream_ptr(x.device)))
            ctx.save_for_backward(x, dcoefs if dcoefs is not None else null, y, nz32 if nz32 is not None else null,
                                  s32 if s32 is not None else null)
            ctx.has = (dcoefs is not None, noise is not None, b is not None, strength is not None)
            ctx.nhwc = nhwc
            ctx.noise_meta = None if noise is None else (noise.dtype, noise.shape, nz_stride)
            ctx.b_dtype = None if b is None else b.dtype
            ctx.s_dtype = None if strength is None else strength.dtype
            return y

        @staticmethod
        def backward(ctx, dy):
            x, dcoefs, y, nz32, s32 = ctx.saved_tensors
            n, c, h, w = x.shape
            has_d, has_nz, has_b, has_s = ctx.has
            nz_gain = s32.reshape([]) if has_s else 1.0
            need = ctx.needs_input_grad          # x, dcoefs, noise, strength, b
            nhwc = ctx.nhwc
            dy = dy.contiguous(memory_format=torch.channels_last) if nhwc else dy.contiguous()
            ident = (act == 'linear' and gain == 1 and clamp < 0)
            if torch.is_grad_enabled():
                # differentiable assembly (create_graph=True)
                dpre = dy if ident else grad_fn.apply(dy, null, null, y)
                dx = dpre * dcoefs.to(dpre.dtype).reshape(n, c, 1, 1) if has_d else dpre
                gd = (dpre * x).sum(dim=[2, 3]).to(dcoefs.dtype) if has_d and need[1] else None
                gn = gs = None
                if has_nz and (need[2] or (has_s and need[3])):
                    dt, shp, stride = ctx.noise_meta
                    per_sample = stride != 0
                    gnoise = (dpre.sum(dim=1, keepdim=True) if per_sample else dpre.sum(dim=[0, 1])).reshape(shp)
                    if need[2]:
                        gn = (gnoise * nz_gain).to(dt)
                    if has_s and need[3]:
                        gs = (gnoise.to(torch.float32) * nz32.reshape(shp)).sum().to(ctx.s_dtype)
                gb = dpre.sum(dim=[0, 2, 3]).to(ctx.b_dtype) if has_b and need[4] else None
                return (dx if need[0] else None), gd, gn, gs, gb
            # plain backward: one fused kernel with the reductions in float32 accumulators
            want_gd = has_d and need[1]
            want_gb = has_b and need[4]
            want_gs = has_nz and has_s and need[3]
            want_gn = has_nz and need[2]
            dx = torch.empty_like(x)
            nb = n * c if nhwc else c       # the NHWC kernel leaves per-sample partial bias sums
            acc = torch.zeros([n * c + nb + native.GS_SLOTS], dtype=torch.float32, device=x.device)
            gd_a, gb_a, gs_a = acc[:n * c], acc[n * c:n * c + nb], acc[n * c + nb:]     # gs: GS_SLOTS partial sums
            stride = ctx.noise_meta[2] if has_nz else 0
            d32 = dcoefs.to(torch.float32).contiguous() if has_d else None
            if x.numel():
                with native.guard(x.device):
                    bwd = native.lib().mg_bwd_fused_nhwc if nhwc else native.lib().mg_bwd_fused
                    native.check(bwd(
                        native.ptr(dy), native.ptr(None if ident else y), native.ptr(x if want_gd else None), native.ptr(d32),
                        native.ptr(nz32 if want_gs else None), native.ptr(dx), native.ptr(gd_a if want_gd else None),
                        native.ptr(gb_a if want_gb else None), native.ptr(gs_a if want_gs else None),
                        native.dtype_code(x.dtype), idx, alpha, gain, clamp, stride, n, c, h * w, native.stream_ptr(x.device)))
            gd = gd_a.reshape(n, c).to(dcoefs.dtype) if want_gd else None
            gb = (gb_a.reshape(n, c).sum(0) if nhwc else gb_a).to(ctx.b_dtype) if want_gb else None
            gs = gs_a.sum().to(ctx.s_dtype) if want_gs else None
            gn = None
            if want_gn:         # a caller-scaled noise tensor that wants its own gradient: not the hot path
                dt, shp, stride = ctx.noise_meta
                dpre = dx / dcoefs.to(dx.dtype).reshape(n, c, 1, 1) if has_d else dx
                gn = ((dpre.sum(dim=1, keepdim=True) if stride else dpre.sum(dim=[0, 1])).reshape(shp) * nz_gain).to(dt)
            return (dx if need[0] else None), gd, gn, gs, gb

    _cache[key] = ModconvEpilogue
    return ModconvEpilogue


def modconv_epilogue(x, dcoefs=None, noise=None, b=None, act='lrelu', alpha=None, gain=None, clamp=None, impl='cuda',
                     noise_strength=None):
    """y = clamp(act(x * dcoefs[N,C,1,1] + noise * noise_strength + b[C]) * gain).

    x [N,C,H,W]; dcoefs [N,C] or None; noise [H,W] / [1,1,H,W] (shared) or [N,1,H,W] (per sample) or None;
    noise_strength: 0-dim tensor multiplying `noise` (None: `noise` is already scaled); b [C] or None."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    spec = _ba.activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    fusable = 'x' not in spec.ref and not spec.has_2nd_grad
    if impl == 'cuda' and x.device.type == 'cuda' and x.dtype != torch.float64 and fusable and (x.is_contiguous() or _is_nhwc(x)):
        return _epilogue_hip(act, alpha, gain, clamp).apply(x, dcoefs, noise, noise_strength, b)
    if noise is not None and noise_strength is not None:
        noise = noise * noise_strength
    return _epilogue_ref(x, dcoefs, noise, b, act, alpha, gain, None if clamp < 0 else clamp)


# ---------------------------------------------------------------------------------------------------------------------
# The up-sampling layers (SynthesisLayer with up=2, reference training/networks.py:299-313 -> conv2d_resample.py:123-144) end in
#     x = upfirdn2d(conv_transpose2d(x, w, stride 2), f, padding, gain = 4)      # the 4x4 blur: read [r+1]^2, write r^2
#     x = fma(x, dcoefs, noise) ; x = bias_act(x, b, lrelu, gain, clamp)         # read r^2, write r^2
# ``blur_epilogue`` is both in ONE pass over the transposed convolution's output (mg_upfirdn2d_epilogue: the channels-last 1:1 FIR
# kernels of csrc/upfirdn2d.hip with the epilogue applied to each output vector on its way out -- same rounding points: the
# filter's result to x.dtype, dcoefs to x.dtype, the fma sum to x.dtype).  Backward (plain): the fused epilogue gradient with the
# filter's output REBUILT from the saved y (mg_bwd_fused_nhwc_y, recon: the filtered tensor is never written), then the filter's
# own gradient (upfirdn2d with the mirrored filter).  Under create_graph=True: the composition of the two differentiable ops.
blur_fusion = __import__('os').environ.get('MG_BLUR_EPILOGUE', '1') != '0'


def _blur_sizes(x, f, padding):
    px0, px1, py0, py1 = padding
    fh, fw = (int(f.shape[0]), int(f.shape[1])) if f.ndim == 2 else (int(f.shape[0]), int(f.shape[0]))
    return fh, fw, int(x.shape[2]) + py0 + py1 - fh + 1, int(x.shape[3]) + px0 + px1 - fw + 1


def blur_epilogue_applicable(x, f, padding, act):
    if not (blur_fusion and x.is_cuda and x.dtype in (torch.float16, torch.bfloat16) and _is_nhwc(x) and f is not None and f.ndim == 2
            and f.shape[0] <= 4 and f.shape[1] <= 4 and act in ('linear', 'lrelu')):
        return False
    fh, fw, oh, ow = _blur_sizes(x, f, padding)
    return oh >= 1 and ow >= 1 and x.numel() > 0


_blur_cache = dict()


def _blur_epilogue_hip(act, alpha, gain, clamp, padding, flip_filter, fgain):
    key = (act, alpha, gain, clamp, tuple(padding), flip_filter, fgain)
    if key in _blur_cache:
        return _blur_cache[key]
    idx = _ba.activation_funcs[act].cuda_idx
    from . import upfirdn2d as _up
    px0, px1, py0, py1 = padding

    class BlurEpilogue(torch.autograd.Function):
        """inputs: x (transposed-convolution output), f, dcoefs, noise (raw), strength (0-dim), b"""

        @staticmethod
        def forward(ctx, x, f, dcoefs, noise, strength, b):
            n, c, ih, iw = x.shape
            fh, fw, oh, ow = _blur_sizes(x, f, padding)
            d32 = dcoefs.detach().to(torch.float32).contiguous() if dcoefs is not None else None
            nz32, nz_stride, s32 = None, 0, None
            if noise is not None:
                assert noise.numel() in (oh * ow, n * oh * ow), 'noise must be [H,W] or [N,1,H,W]'
                nz32 = noise.detach().to(torch.float32).contiguous()
                nz_stride = oh * ow if noise.numel() == n * oh * ow and n > 1 else 0
                if strength is not None:
                    s32 = strength.detach().to(torch.float32).reshape(1).contiguous()
            bb = b.detach().to(x.dtype).contiguous() if b is not None else None
            f32 = f.detach().to(torch.float32)
            y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
            with native.guard(x.device):
                native.check(native.lib().mg_upfirdn2d_epilogue(
                    native.ptr(x), native.ptr(f32), native.ptr(y), native.dtype_code(x.dtype), native.I32x4(n, c, ih, iw), native.I32x2(fh, fw),
                    native.I64x2(f32.stride(0), f32.stride(1)), native.I32x4(n, c, oh, ow), px0, py0, int(flip_filter), fgain,
                    native.ptr(d32), native.ptr(nz32), native.ptr(s32), native.ptr(bb), nz_stride, idx, alpha, gain, clamp,
                    native.stream_ptr(x.device)))
            ctx.save_for_backward(x, f, dcoefs if dcoefs is not None else _ba._null_tensor, y, nz32 if nz32 is not None else _ba._null_tensor,
                                  s32 if s32 is not None else _ba._null_tensor, noise if noise is not None else _ba._null_tensor,
                                  strength if strength is not None else _ba._null_tensor, b if b is not None else _ba._null_tensor)
            ctx.has = (dcoefs is not None, noise is not None, strength is not None, b is not None)
            ctx.nz_stride = nz_stride
            return y

        @staticmethod
        def backward(ctx, dy):
            x, f, dcoefs, y, nz32, s32, noise, strength, b = ctx.saved_tensors
            has_d, has_nz, has_s, has_b = ctx.has
            need = ctx.needs_input_grad          # x, f, dcoefs, noise, strength, b
            if torch.is_grad_enabled():
                full = [x, None, dcoefs if has_d else None, noise if has_nz else None, strength if has_s else None, b if has_b else None]
                live = [t for t, nd in zip(full, need) if t is not None and nd and t.requires_grad]
                with torch.enable_grad():
                    yb = _up.upfirdn2d(x, f, padding=list(padding), flip_filter=flip_filter, gain=fgain)
                    out = modconv_epilogue(yb, dcoefs if has_d else None, noise if has_nz else None, b if has_b else None, act=act, alpha=alpha,
                                           gain=gain, clamp=None if clamp < 0 else clamp, noise_strength=strength if has_s else None)
                    grads = torch.autograd.grad([out], live, [dy], create_graph=True, allow_unused=True) if live else ()
                table = {id(t): g for t, g in zip(live, grads)}
                return tuple(table.get(id(t)) if (t is not None and nd) else None for t, nd in zip(full, need))
            assert not (has_nz and need[3]), 'a noise tensor that wants its own gradient takes the unfused path'
            n, c, oh, ow = y.shape
            dy = dy.contiguous(memory_format=torch.channels_last)
            want_gd, want_gb, want_gs = has_d and need[2], has_b and need[5], has_nz and has_s and need[4]
            dxb = torch.empty_like(y)
            acc = torch.zeros([2 * n * c + native.GS_SLOTS], dtype=torch.float32, device=y.device)
            gd_a, gb_a, gs_a = acc[:n * c], acc[n * c:2 * n * c], acc[2 * n * c:]
            d32 = dcoefs.to(torch.float32).contiguous() if has_d else None
            b16 = b.detach().to(y.dtype).contiguous() if (want_gd and has_b) else None
            with native.guard(y.device):
                native.check(native.lib().mg_bwd_fused_nhwc_y(
                    native.ptr(dy), native.ptr(y), None, native.ptr(d32), native.ptr(nz32 if (want_gs or (want_gd and has_nz)) else None),
                    native.ptr(s32 if (want_gd and has_nz and has_s) else None), native.ptr(b16), int(want_gd), native.ptr(dxb),
                    native.ptr(gd_a if want_gd else None), native.ptr(gb_a if want_gb else None), native.ptr(gs_a if want_gs else None),
                    native.dtype_code(y.dtype), idx, alpha, gain, clamp, ctx.nz_stride, n, c, oh * ow, native.stream_ptr(y.device)))
            gd = gd_a.reshape(n, c).to(dcoefs.dtype) if want_gd else None
            gb = gb_a.reshape(n, c).sum(0).to(b.dtype) if want_gb else None
            gs = gs_a.sum().to(strength.dtype) if want_gs else None
            gx = None
            if need[0]:     # the filter's own gradient (reference upfirdn2d.py:263-278): the same op, filter mirrored, padding recomputed
                fh, fw = int(f.shape[0]), int(f.shape[1])
                pb = [fw - px0 - 1, int(x.shape[3]) - ow + px0, fh - py0 - 1, int(x.shape[2]) - oh + py0]
                gx = _up.upfirdn2d(dxb, f, padding=pb, flip_filter=(not flip_filter), gain=fgain)
            return gx, None, gd, None, gs, gb

    _blur_cache[key] = BlurEpilogue
    return BlurEpilogue


def blur_epilogue(x, f, padding, flip_filter=False, fgain=1.0, dcoefs=None, noise=None, b=None, act='lrelu', alpha=None, gain=None,
                  clamp=None, noise_strength=None):
    """modconv_epilogue(upfirdn2d(x, f, padding=padding, flip_filter=flip_filter, gain=fgain), dcoefs, noise, b, ...) -- one native
    pass where ``blur_epilogue_applicable``, the two ops otherwise.  padding = [px0, px1, py0, py1]."""
    spec = _ba.activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    if blur_epilogue_applicable(x, f, padding, act):
        _ba._init()
        return _blur_epilogue_hip(act, alpha, gain, float(-1 if clamp is None else clamp), tuple(int(v) for v in padding), bool(flip_filter),
                                  float(fgain)).apply(x, f, dcoefs, noise, noise_strength, b)
    from . import upfirdn2d as _up
    yb = _up.upfirdn2d(x, f, padding=list(padding), flip_filter=flip_filter, gain=fgain)
    return modconv_epilogue(yb, dcoefs, noise, b, act=act, alpha=alpha, gain=gain, clamp=clamp, noise_strength=noise_strength)


class _ChannelScale(torch.autograd.Function):
    """y = x * s[N,C,1,1] on the native path; backward = one fused kernel (dx = dy*s, ds = sum_hw dy*x)."""

    @staticmethod
    def forward(ctx, x, s):
        n, c, h, w = x.shape
        nhwc = _is_nhwc(x)
        if not nhwc:
            x = x.contiguous()
        s32 = s.detach().to(torch.float32).contiguous()      # the kernel rounds it to x.dtype (styles.to(x.dtype), as the reference)
        y = torch.empty_like(x)
        ctx.nhwc = nhwc
        if x.numel():
            with native.guard(x.device):
                fwd = native.lib().mg_fma_bias_act_nhwc if nhwc else native.lib().mg_fma_bias_act
                native.check(fwd(native.ptr(x), native.ptr(s32), native.ptr(None), native.ptr(None),
                                                          native.ptr(y), native.dtype_code(x.dtype), 1, 0.0, 1.0, -1.0, 1.0, native.ptr(None), 0,
                                                          n, c, h * w, native.stream_ptr(x.device)))
        ctx.save_for_backward(x, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        n, c, h, w = x.shape
        if torch.is_grad_enabled():
            dx = dy * s.to(dy.dtype).reshape(n, c, 1, 1) if ctx.needs_input_grad[0] else None
            ds = (dy * x).sum(dim=[2, 3]).to(s.dtype) if ctx.needs_input_grad[1] else None
            return dx, ds
        dy = dy.contiguous(memory_format=torch.channels_last) if ctx.nhwc else dy.contiguous()
        dx = torch.empty_like(x)
        want_ds = ctx.needs_input_grad[1]
        gd = torch.zeros([n * c], dtype=torch.float32, device=x.device) if want_ds else None
        s32 = s.to(torch.float32).contiguous()
        if x.numel():
            with native.guard(x.device):
                bwd = native.lib().mg_bwd_fused_nhwc if ctx.nhwc else native.lib().mg_bwd_fused
                native.check(bwd(native.ptr(dy), native.ptr(None), native.ptr(x if want_ds else None),
                                                       native.ptr(s32), native.ptr(None), native.ptr(dx), native.ptr(gd),
                                                       native.ptr(None), native.ptr(None), native.dtype_code(x.dtype), 1, 0.0, 1.0,
                                                       -1.0, 0, n, c, h * w, native.stream_ptr(x.device)))
        return (dx if ctx.needs_input_grad[0] else None), (gd.reshape(n, c).to(s.dtype) if want_ds else None)


def channel_scale(x, s, impl='cuda'):
    """x [N,C,H,W] * s [N,C] broadcast over H,W -- the style modulation in front of a shared-weight convolution
    (reference training/networks.py:70: ``x * styles.to(x.dtype).reshape(batch_size, -1, 1, 1)``)."""
    if impl == 'cuda' and x.device.type == 'cuda' and x.dtype != torch.float64 and x.ndim == 4 and (x.is_contiguous() or _is_nhwc(x)):
        return _ChannelScale.apply(x, s)
    return x * s.to(x.dtype).reshape(x.shape[0], -1, 1, 1)
