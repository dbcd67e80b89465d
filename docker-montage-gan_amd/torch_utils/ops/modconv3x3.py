"""The 64-channel, full-resolution modulated 3x3 convolution as one matrix-core kernel.

``fused_modconv3x3(x, w, styles, dcoefs, noise, noise_strength, b, ...)`` computes what the training path of
``modulated_conv2d`` plus the layer's ``bias_act`` computes (reference training/networks.py:37-98, 311-313)

    y = clamp(act(conv3x3(x * styles[N,Ci,1,1], w) * dcoefs[N,Co,1,1] + noise * noise_strength + b) * gain)

-- as separate launches three of them and six trips of the activation through HBM (modulation pass, library convolution,
fused epilogue pass) -- in ONE launch of csrc/modconv.hip (``mg_modconv3x3_c64``: v_mfma_f32_32x32x16_bf16, the weights of a
wave's output-channel half resident in its registers with the modulation folded into them, the input halo double-buffered in
LDS by LDS-DMA, epilogue on the accumulators, two workgroups per CU).  With ``styles = dcoefs = noise = None`` it is the
discriminator's ``conv0`` + ``bias_act`` (reference :600-603); with everything ``None`` and a linear activation a plain
convolution, which is what the data gradient of these layers is (``conv3x3_c64(dy, flipped_transposed(w))``).
Rounding points: the convolution result to bf16, dcoefs to bf16, the fma sum to bf16 -- as the three-launch path; the
modulation is rounded on the WEIGHT (bf16(w * bf16(s)), the reference's own ``fused_modconv=True`` formulation, :91-93) where
the three launches round x * s: equal on integer data bit for bit, within bf16 rounding otherwise.

Scope: bf16, channels-last, 64 -> 64 channels, stride 1, padding 1, H % 4 == 0, W % 32 == 0 -- the 256x256 layers of the
benchmark configuration, the largest activations of both networks.  Anything else keeps the three-launch path
(``applicable()`` says which).

Gradients.  A plain ``backward()``: the fused epilogue gradient (``mg_bwd_fused_nhwc``: d(conv out), d(dcoefs), d(bias),
d(noise_strength) in one pass), the data gradient through THIS kernel on the flipped, transposed weights, the library's
weight gradient on the re-modulated input, and the fused modulation gradient.  Under ``create_graph=True`` (path-length
regularisation) the backward re-runs the differentiable three-launch composition on the saved inputs, so gradients of any order
exist.

Measured at [32,64,256,256] (profiles/r03_modconv_microbench.log): forward 260 us against 635 us for the three launches
(595 TFLOP/s), forward + backward 1411 against 1784 us; over the training step at batch 32: 616 against 633 ms
(profiles/r03_bench_b32_modconv_{on,off}.json).  On by default; ``MG_MODCONV=0`` switches it off.
"""
import os

import torch

from .. import native
from . import bias_act as _ba
from . import conv2d_gradfix
from . import modconv_epilogue as _me

enabled = os.environ.get('MG_MODCONV', '1') != '0'
wgrad_enabled = os.environ.get('MG_WGRAD', '1') != '0'          # weight / style gradients of these layers through csrc/wgrad.hip
reconstruct = os.environ.get('MG_MODCONV_RECON', '1') != '0'     # rebuild the raw convolution output from y in the backward pass


def applicable(x, weight, up=1, down=1):
    """True where the kernel's scope covers the call (GPU, bf16, channels-last, 64 -> 64, 3x3, no resampling, tile-aligned)."""
    return (enabled and x.is_cuda and x.dtype == torch.bfloat16 and x.ndim == 4 and up == 1 and down == 1
            and tuple(weight.shape) == (64, 64, 3, 3) and x.shape[1] == 64 and x.shape[2] % 4 == 0 and x.shape[3] % 32 == 0
            and x.shape[0] > 0 and _me._is_nhwc(x))


def _launch(x, w, s32, d32, nz32, nz_stride, strength32, b16, want_conv, want_xs, act_idx, alpha, gain, clamp):
    n, c, h, wd = x.shape
    y = torch.empty_like(x)
    yconv = torch.empty_like(x) if want_conv else None
    xs = torch.empty_like(x) if want_xs else None
    with native.guard(x.device):
        native.check(native.lib().mg_modconv3x3_c64(
            native.ptr(x), native.ptr(w), native.ptr(s32), native.ptr(d32), native.ptr(nz32), native.ptr(strength32), native.ptr(b16),
            native.ptr(yconv), native.ptr(y), native.ptr(xs), n, h, wd, nz_stride, act_idx, alpha, gain, clamp, native.stream_ptr(x.device)))
    return y, yconv, xs


def conv3x3_c64(x, w, scale=None):
    """Plain 3x3 convolution (stride 1, padding 1, 64 -> 64, bf16 channels-last) through the same kernel, no autograd: what the
    data gradient of these layers is -- ``conv3x3_c64(dy, flipped_transposed(w))`` (the layer's own forward problem, as
    conv2d_gradfix does for the wider layers through the library's forward solvers).  ``scale`` [N,64]: the result times
    scale[n, channel], rounded as ``conv * scale.to(bf16)`` -- the gradient of the modulation ``x * styles`` w.r.t. x, in the
    same launch."""
    n, c, h, wd = x.shape
    y = torch.empty_like(x)
    d32 = scale.detach().to(torch.float32).contiguous() if scale is not None else None
    with native.guard(x.device):
        native.check(native.lib().mg_modconv3x3_c64(
            native.ptr(x), native.ptr(w), None, native.ptr(d32), None, None, None, None, native.ptr(y), None, n, h, wd, 0, 1, 0.0, 1.0, -1.0,
            native.stream_ptr(x.device)))
    return y


def wgrad3x3_c64(dy, x, styles=None, w=None, raw=False):
    """Weight gradient of these layers in one matrix-core launch (csrc/wgrad.hip), the modulation folded in:
    dW[o,i,kh,kw] = sum_n styles.to(bf16)[n,i] * sum_p dy[n,o,p] * x[n,i,p+tap], and -- with ``styles`` and ``w`` given --
    ds[n,i] = sum_{o,tap} w[o,i,tap] * sum_p dy[n,o,p] * x[n,i,p+tap], the gradient of the modulation w.r.t. the styles.
    -> (dW [64,64,3,3] channels-last in dy.dtype (float32 with raw=True), ds [N,64] float32 or None)."""
    n, c, h, wd = x.shape
    want_ds = styles is not None and w is not None
    acc = torch.zeros([64 * 9 * 64 + (n * 64 if want_ds else 0)], dtype=torch.float32, device=x.device)
    dw32 = acc[:64 * 9 * 64]
    ds32 = acc[64 * 9 * 64:].view(n, 64) if want_ds else None
    s32 = styles.detach().to(torch.float32).contiguous() if styles is not None else None
    with native.guard(x.device):
        native.check(native.lib().mg_wgrad3x3_c64(native.ptr(dy), native.ptr(x), native.ptr(s32), native.ptr(w if want_ds else None),
                                                   native.ptr(dw32), native.ptr(ds32), n, h, wd, native.stream_ptr(x.device)))
    dw = dw32.view(64, 3, 3, 64).permute(0, 3, 1, 2)            # [O,I,3,3] with channels-last strides
    return (dw if raw else dw.to(dy.dtype)), ds32


def flipped_transposed(w):
    """[O,I,3,3] -> the weights of the transposed problem: w'[i,o,kh,kw] = w[o,i,2-kh,2-kw], channels-last."""
    return w.flip([2, 3]).transpose(0, 1).contiguous(memory_format=torch.channels_last)


def _composition(x, w, styles, dcoefs, noise, strength, b, act, alpha, gain, clamp):
    """The differentiable three-launch path (what the layers run where the kernel does not apply)."""
    xs = _me.channel_scale(x, styles) if styles is not None else x
    yc = conv2d_gradfix.conv2d(xs, w, padding=1)
    if dcoefs is None and noise is None:
        return _ba.bias_act(yc, b.to(yc.dtype) if b is not None else None, act=act, alpha=alpha, gain=gain, clamp=None if clamp < 0 else clamp)
    return _me.modconv_epilogue(yc, dcoefs, noise, b, act=act, alpha=alpha, gain=gain, clamp=None if clamp < 0 else clamp,
                                noise_strength=strength)


_cache = dict()


def _fused(act, alpha, gain, clamp):
    key = (act, alpha, gain, clamp)
    if key in _cache:
        return _cache[key]
    spec = _ba.activation_funcs[act]
    idx = spec.cuda_idx
    assert idx in (1, 3), 'linear or lrelu'
    null = _ba._null_tensor

    class FusedModConv3x3(torch.autograd.Function):
        """inputs: x, w (bf16, channels-last), styles, dcoefs, noise (raw), strength (0-dim), b"""

        @staticmethod
        def forward(ctx, x, w, styles, dcoefs, noise, strength, b):
            n, c, h, wd = x.shape
            s32 = styles.detach().to(torch.float32).contiguous() if styles is not None else None
            d32 = dcoefs.detach().to(torch.float32).contiguous() if dcoefs is not None else None
            nz32, nz_stride, st32 = None, 0, None
            if noise is not None:
                assert noise.numel() in (h * wd, n * h * wd), 'noise must be [H,W] or [N,1,H,W]'
                nz32 = noise.detach().to(torch.float32).contiguous()
                nz_stride = h * wd if noise.numel() == n * h * wd and n > 1 else 0
                if strength is not None:
                    st32 = strength.detach().to(torch.float32).reshape(1).contiguous()
            b16 = b.detach().to(x.dtype).contiguous() if b is not None else None
            wc = w.detach()
            assert wc.is_contiguous(memory_format=torch.channels_last) and wc.dtype == x.dtype
            need_bwd = any(ctx.needs_input_grad)
            # the weight gradient of the convolution reads x * s: written by the same launch (one more store of the tile
            # while it is in registers) rather than recomputed in the backward pass (a read and a write)
            want_xs = False     # (round 3: the kernel folds the modulation into its weights; x * s is recomputed where the library's weight gradient needs it)
            # the demodulation gradient sums dpre * (raw convolution output): rebuilt from y inside the backward kernel
            # (mg_bwd_fused_nhwc_y, recon) instead of written here and read there -- two passes over the activation less
            want_conv = need_bwd and dcoefs is not None and not reconstruct
            with conv2d_gradfix._timed_call(x, (64, 64, 3, 3), 1, 'fwd [mg_modconv3x3_c64]'):
                y, yconv, xs = _launch(x, wc, s32, d32, nz32, nz_stride, st32, b16, want_conv, want_xs, idx, alpha, gain, clamp)
            if need_bwd:
                ctx.save_for_backward(x, w, styles if styles is not None else null, dcoefs if dcoefs is not None else null,
                                      yconv if yconv is not None else null, y, xs if xs is not None else null,
                                      nz32 if nz32 is not None else null, st32 if st32 is not None else null,
                                      noise if noise is not None else null, strength if strength is not None else null,
                                      b if b is not None else null)
            ctx.has = (styles is not None, dcoefs is not None, noise is not None, strength is not None, b is not None)
            ctx.nz_stride = nz_stride
            ctx.b_dtype = None if b is None else b.dtype
            ctx.s_dtype = None if strength is None else strength.dtype
            return y

        @staticmethod
        def backward(ctx, dy):
            x, w, styles, dcoefs, yconv, y, xs_saved, nz32, st32, noise, strength, b = ctx.saved_tensors
            has_s, has_d, has_nz, has_st, has_b = ctx.has
            need = ctx.needs_input_grad          # x, w, styles, dcoefs, noise, strength, b
            if torch.is_grad_enabled():
                # create_graph=True: differentiate the three-launch composition of the saved inputs -- towards the inputs THIS pass
                # needs only (the regularisers ask for ws / the image: no weight, no bias).  Asking for more is not just wasted work:
                # the prepared weight is one output of the network's weight-preparation node, and a request for its gradient makes
                # the engine run every backward node between x and that node, i.e. the whole generator below this layer.
                full = [x, w, styles if has_s else None, dcoefs if has_d else None, noise if has_nz else None,
                        strength if has_st else None, b if has_b else None]
                live = [t for t, nd in zip(full, need) if t is not None and nd and t.requires_grad]
                with torch.enable_grad():
                    out = _composition(x, w, styles if has_s else None, dcoefs if has_d else None, noise if has_nz else None,
                                       strength if has_st else None, b if has_b else None, act, alpha, gain, clamp)
                    grads = torch.autograd.grad([out], live, [dy], create_graph=True, allow_unused=True) if live else ()
                table = {id(t): g for t, g in zip(live, grads)}
                return tuple(table.get(id(t)) if (t is not None and nd) else None for t, nd in zip(full, need))
            n, c, h, wd = x.shape
            assert not (has_nz and need[4]), 'a noise tensor that wants its own gradient takes the three-launch path'
            # (1) epilogue: d(conv out) and the float32 reductions, one pass
            dy = dy.contiguous(memory_format=torch.channels_last)
            ident = (act == 'linear' and gain == 1 and clamp < 0)
            want_gd, want_gb, want_gs = has_d and need[3], has_b and need[6], has_nz and has_st and need[5]
            dyc = torch.empty_like(x)
            acc = torch.zeros([2 * n * c + native.GS_SLOTS], dtype=torch.float32, device=x.device)
            gd_a, gb_a, gs_a = acc[:n * c], acc[n * c:2 * n * c], acc[2 * n * c:]
            d32 = dcoefs.to(torch.float32).contiguous() if has_d else None
            recon = bool(want_gd and not yconv.numel())
            b16 = b.detach().to(x.dtype).contiguous() if (recon and has_b) else None
            with native.guard(x.device):
                native.check(native.lib().mg_bwd_fused_nhwc_y(
                    native.ptr(dy), native.ptr(y if (recon or not ident) else None), native.ptr(yconv if (want_gd and not recon) else None), native.ptr(d32),
                    native.ptr(nz32 if (want_gs or (recon and has_nz)) else None), native.ptr(st32 if (recon and has_nz and has_st) else None),
                    native.ptr(b16), int(recon), native.ptr(dyc), native.ptr(gd_a if want_gd else None),
                    native.ptr(gb_a if want_gb else None), native.ptr(gs_a if want_gs else None), native.dtype_code(x.dtype), idx, alpha, gain,
                    clamp, ctx.nz_stride, n, c, h * wd, native.stream_ptr(x.device)))
            gd = gd_a.reshape(n, c).to(dcoefs.dtype) if want_gd else None
            gb = gb_a.reshape(n, c).sum(0).to(ctx.b_dtype) if want_gb else None
            gs = gs_a.sum().to(ctx.s_dtype) if want_gs else None
            # (2) convolution and modulation gradients
            need_w = bool(need[1] and not conv2d_gradfix.weight_gradients_disabled)
            need_xs = need[0] or (has_s and need[2])
            dx = dw = ds = None
            if wgrad_enabled:
                # two matrix-core launches: dx = conv(dyc, flipped w^T) * styles (this layer's forward kernel on the transposed
                # problem, the modulation's x-gradient as its per-channel output scale), and dW + d(styles) from csrc/wgrad.hip --
                # instead of: re-modulating x, the library's split-K weight gradient, the data gradient, a pass for dxs * s and sum dxs * x
                if need[0]:
                    with conv2d_gradfix._timed_call(x, (64, 64, 3, 3), 1, 'bwd1 [dx: mg_modconv3x3_c64]'):
                        dx = conv3x3_c64(dyc, flipped_transposed(w.detach()), scale=styles.detach() if has_s else None)
                if need_w or (has_s and need[2]):
                    with conv2d_gradfix._timed_call(x, (64, 64, 3, 3), 1, 'bwd1 [dw: mg_wgrad3x3_c64]'):
                        dw, ds32 = wgrad3x3_c64(dyc, x.detach(), styles.detach() if has_s else None, w.detach() if (has_s and need[2]) else None)
                    ds = ds32.to(styles.dtype) if (has_s and need[2]) else None
                    if not need_w:
                        dw = None
            elif need_xs or need_w:
                if has_s and need_w:
                    xs = xs_saved if xs_saved.numel() else _me.channel_scale(x.detach(), styles.detach())
                else:
                    xs = x.detach()
                with conv2d_gradfix._timed_call(xs, (64, 64, 3, 3), int(bool(need_xs)) + int(need_w), 'bwd' + str(int(bool(need_xs)) + int(need_w)) + ' [dx: mg_modconv3x3_c64]'):
                    # the data gradient is this layer's forward problem on dy and the flipped, transposed weights: the same kernel
                    dxs = conv3x3_c64(dyc, flipped_transposed(w.detach())) if need_xs else None
                    dw = torch.ops.aten.convolution_backward(dyc, xs, w.detach(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                             [False, True, False])[1] if need_w else None
                # (3) modulation: dx = dxs * s, ds = sum_hw dxs * x, one pass
                if need_xs and has_s:
                    dx = torch.empty_like(x)
                    gsd = torch.zeros([n * c], dtype=torch.float32, device=x.device) if need[2] else None
                    s32 = styles.to(torch.float32).contiguous()
                    with native.guard(x.device):
                        native.check(native.lib().mg_bwd_fused_nhwc(
                            native.ptr(dxs), native.ptr(None), native.ptr(x if need[2] else None), native.ptr(s32), native.ptr(None),
                            native.ptr(dx), native.ptr(gsd), native.ptr(None), native.ptr(None), native.dtype_code(x.dtype), 1, 0.0, 1.0,
                            -1.0, 0, n, c, h * wd, native.stream_ptr(x.device)))
                    ds = gsd.reshape(n, c).to(styles.dtype) if need[2] else None
                    if not need[0]:
                        dx = None
                elif need_xs:
                    dx = dxs
            return dx, dw, ds, gd, None, gs, gb

    _cache[key] = FusedModConv3x3
    return FusedModConv3x3


def fused_modconv3x3(x, w, styles=None, dcoefs=None, noise=None, noise_strength=None, b=None, act='lrelu', alpha=None, gain=None,
                     clamp=None):
    """See the module docstring.  The caller checks ``applicable(x, w)`` first."""
    spec = _ba.activation_funcs[act]
    alpha = float(spec.def_alpha if alpha is None else alpha)
    gain = float(spec.def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    _ba._init()
    return _fused(act, alpha, gain, clamp).apply(x, w, styles, dcoefs, noise, noise_strength, b)
